// Two-product fp16 instance of the split-precision MLP kernel (mlp_split_impl.inc): activations x = hi + lo in IEEE
// half (22 mantissa bits), weights rounded ONCE to half (11 bits): x_hi*w + x_lo*w, two v_mfma_f32_32x32x16_f16 per
// GEMM tile instead of three and half the weight stream.  The weight rounding is a fixed 2^-12 relative perturbation
// of the network: RGB moves by ~1e-5 on the benchmark scenes (measured next to bf16x3 in DESIGN.md).  Opt-in.
#define HR_SPLIT_E _Float16
#define HR_SPLIT_RANGE_CHECK 1      // IEEE-half operands: keep the sticky overflow bit (mlp_split_core.inc)
#define HR_SPLIT_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define HR_SPLIT_KERNEL hr_mlp_f16x2_kernel
#define HR_SPLIT_LAUNCH hr_launch_mlp_f16x2
#define HR_SPLIT_PRODUCTS 2
#define HR_W_LOAD_AUX 0            // weights through buffer loads (mlp_split_core.inc, hr_load_w)
#include "mlp_split_impl.inc"
