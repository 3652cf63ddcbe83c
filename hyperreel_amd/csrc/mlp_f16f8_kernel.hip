// EXPERIMENTAL (built only with -DHR_WITH_F16F8, tools/build_variant.py): fp16 main product + fp8 cross terms.
// x = hi + lo and w = hi + lo in IEEE half as in mlp_f16x3_kernel.hip; x_hi w_hi stays a v_mfma_f32_32x32x16_f16 product, the
// two cross terms x_hi w_lo + x_lo w_hi -- 2^-11 of the result -- become two v_mfma_f32_32x32x64_f8f6f4 products of e4m3 operands
// at twice the rate: the matrix work of f16x2 with ~2^-15 instead of ~2^-12 relative error per product (mlp_split_core.inc).
// The MLP input segment (layer 0, skip layers) keeps the three-product form.  Not validated on the device yet.
#define HR_SPLIT_E _Float16
#define HR_SPLIT_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define HR_SPLIT_KERNEL hr_mlp_f16f8_kernel
#define HR_SPLIT_LAUNCH hr_launch_mlp_f16f8
#define HR_SPLIT_PRODUCTS 4
#include "mlp_split_impl.inc"
