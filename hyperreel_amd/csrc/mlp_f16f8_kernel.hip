// f16 + fp8 instance of the split-precision MLP kernel (mlp_split_impl.inc): x = hi + lo and w = hi + lo as in mlp_f16x3_kernel.hip, but only the
// leading product x_hi*w_hi is a v_mfma_f32_32x32x16_f16; the two correction products x*w_lo + x_lo*w_hi (2^-11 of it) are ONE
// v_mfma_scale_f32_32x32x64_f8f6f4 per pair of k-steps with fp8 e4m3 images of their operands and E8M0 block scales (mlp_split_core.inc,
// hr_accumulate_f8): two thirds of the matrix-pipe time of f16x3, the same operand bytes, the result accurate to ~2^-16 per product instead of
// 2^-22 (f16x3) or 2^-12 (f16x2).  The input segment of the first / skip layers keeps the three f16 products.  gfx950 only.
#define HR_SPLIT_E _Float16
#define HR_SPLIT_RANGE_CHECK 1      // IEEE-half AND e4m3 operands: the sticky overflow bit covers both ranges (mlp_split_core.inc)
#define HR_SPLIT_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define HR_SPLIT_KERNEL hr_mlp_f16f8_kernel
#define HR_SPLIT_LAUNCH hr_launch_mlp_f16f8
#define HR_SPLIT_F8LO 1
// the windowed positional encoding through v_sin_f32 / v_cos_f32 (|err| ~ 5e-7 on features of O(1)) instead of libm-grade sincosf: a fortieth of this
// arithmetic's own head error, and a third of the prologue's instructions (K1's phase trace: the prologue is 9 % of a tile).  The exact
// arithmetics (f16x3, bf16x3, fp32) keep sincosf
#define HR_FAST_SINCOS 1
#define HR_W_LOAD_AUX 0            // weights through buffer loads (mlp_split_core.inc, hr_load_w)
#include "mlp_split_impl.inc"
