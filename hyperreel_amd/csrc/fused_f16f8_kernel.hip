// EXPERIMENTAL (-DHR_WITH_F16F8 -DHR_F16F8_V2 builds only, NOT yet run on the device): the frame kernel (fused_impl.inc) with the
// f16f8 arithmetic of mlp_f16f8v2_kernel.hip -- fp16 main product, both cross terms as e4m3 products, w_hi's bytes derived in registers.
#define HR_SPLIT_E _Float16
#define HR_SPLIT_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define HR_FUSED_KERNEL hr_frame_f16f8_kernel
#define HR_FUSED_LAUNCH hr_launch_frame_f16f8
#define HR_SPLIT_PRODUCTS 5
#define HR_TUNING_SET hr_tuning_set_f16f8
#define HR_TUNING_PHASES hr_tuning_phases_f16f8
#include "fused_impl.inc"
