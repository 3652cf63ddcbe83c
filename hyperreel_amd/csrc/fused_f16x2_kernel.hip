// fp16 two-product instance of the fused frame kernel (fused_impl.inc): weights rounded once to half; see
// mlp_f16x2_kernel.hip.  The viewer mode (BASELINE config 5) pairs it with float16 texels.
#define HR_SPLIT_E _Float16
#define HR_SPLIT_RANGE_CHECK 1      // IEEE-half operands: keep the sticky overflow bit (mlp_split_core.inc)
#define HR_SPLIT_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define HR_FUSED_KERNEL hr_frame_f16x2_kernel
#define HR_FUSED_LAUNCH hr_launch_frame_f16x2
#define HR_SPLIT_PRODUCTS 2
#define HR_TUNING_SET hr_tuning_set_f16x2
#define HR_TUNING_PHASES hr_tuning_phases_f16x2
#include "fused_impl.inc"
