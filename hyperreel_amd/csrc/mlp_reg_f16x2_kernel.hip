// f16x2 instance of the register-resident MLP kernel (mlp_reg_impl.inc); element type and products as in mlp_f16x2_kernel.hip.
#define HR_SPLIT_E _Float16
#define HR_SPLIT_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define HR_SPLIT_PRODUCTS 2
#define HR_REG_KERNEL hr_mlp_reg_f16x2_kernel
#define HR_REG_CHUNKS hr_reg_chunks_f16x2
#define HR_REG_LAUNCH hr_launch_mlp_reg_f16x2
#include "mlp_reg_impl.inc"
