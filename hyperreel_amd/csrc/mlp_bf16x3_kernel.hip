// bf16 instance of the split-precision MLP kernel (mlp_split_impl.inc): x = hi + lo with bf16 halves,
// three v_mfma_f32_32x32x16_bf16 products per fp32 GEMM.
#define HR_SPLIT_E __bf16
#define HR_SPLIT_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define HR_SPLIT_KERNEL hr_mlp_bf16x3_kernel
#define HR_SPLIT_LAUNCH hr_launch_mlp_bf16x3
#define HR_SPLIT_TRAIN_KERNEL hr_mlp_train_bf16x3_kernel      // + the training step's fused forward (bf16 halves: the fp32 exponent range, whatever the weights become)
#define HR_SPLIT_TRAIN_LAUNCH hr_launch_mlp_train_bf16x3
#define HR_W_LOAD_AUX 0            // weights through buffer loads (mlp_split_core.inc, hr_load_w)
#include "mlp_split_impl.inc"
