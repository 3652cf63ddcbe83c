// fp16 instance of the split-precision MLP kernel (mlp_split_impl.inc): x = hi + lo with IEEE half halves (22 mantissa
// bits in total, dropped terms O(2^-22)), three v_mfma_f32_32x32x16_f16 products per fp32 GEMM -- the accuracy of the
// fp32-MFMA kernel at the speed of the bf16 split.  fp16 saturates at 65504: opt-in (mlp_precision='f16x3').
#define HR_SPLIT_E _Float16
#define HR_SPLIT_RANGE_CHECK 1      // IEEE-half operands: keep the sticky overflow bit (mlp_split_core.inc)
#define HR_SPLIT_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define HR_SPLIT_KERNEL hr_mlp_f16x3_kernel
#define HR_SPLIT_LAUNCH hr_launch_mlp_f16x3
#define HR_SPLIT_LIST_NW8 1         // list-driven launches (the verified fast path's second pass) take the eight-wavefront form
#define HR_W_LOAD_AUX 0            // weights through buffer loads (mlp_split_core.inc, hr_load_w)
#include "mlp_split_impl.inc"
