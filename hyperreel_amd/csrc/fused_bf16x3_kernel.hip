// bf16x3 instance of the fused frame kernel (fused_impl.inc): the sample-prediction MLP as three
// v_mfma_f32_32x32x16_bf16 products per fp32 GEMM, handing its head to the sample wavefronts through LDS.
#define HR_SPLIT_E __bf16
#define HR_SPLIT_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define HR_FUSED_KERNEL hr_frame_bf16x3_kernel
#define HR_FUSED_LAUNCH hr_launch_frame_bf16x3
#define HR_TUNING_SET hr_tuning_set_bf16x3
#define HR_TUNING_PHASES hr_tuning_phases_bf16x3
#include "fused_impl.inc"
