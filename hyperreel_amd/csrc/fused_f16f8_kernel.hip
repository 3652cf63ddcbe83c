// f16 + fp8 instance of the fused frame kernel (fused_impl.inc): the leading product as an f16 MFMA, the two correction products as one fp8
// K = 64 MFMA per pair of k-steps; see mlp_f16f8_kernel.hip.
#define HR_SPLIT_E _Float16
#define HR_SPLIT_RANGE_CHECK 1      // IEEE-half and e4m3 operands: keep the sticky overflow bit (mlp_split_core.inc)
#define HR_SPLIT_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define HR_FUSED_KERNEL hr_frame_f16f8_kernel
#define HR_FUSED_LAUNCH hr_launch_frame_f16f8
#define HR_SPLIT_F8LO 1
// the windowed positional encoding through v_sin_f32 / v_cos_f32 (|err| ~ 5e-7 on features of O(1)) instead of libm-grade sincosf: a fortieth of this
// arithmetic's own head error, and a third of the prologue's instructions (K1's phase trace: the prologue is 9 % of a tile).  The exact
// arithmetics (f16x3, bf16x3, fp32) keep sincosf
#define HR_FAST_SINCOS 1
#define HR_TUNING_SET hr_tuning_set_f16f8
#define HR_TUNING_PHASES hr_tuning_phases_f16f8
#include "fused_impl.inc"
