// EXPERIMENTAL, second cut of mlp_f16f8_kernel.hip (built instead of it with -DHR_WITH_F16F8 -DHR_F16F8_V2): the fp8 bytes of a lane
// follow the k-order of its fp16 tiles, so e4m3(2^-12 w_hi) is derived in registers and only w_lo's bytes are streamed; the two fp8
// products of a 64-wide block are issued inside the ring loop.  Same launcher name: one of the two files is linked.  mlp_split_core.inc.
#define HR_SPLIT_E _Float16
#define HR_SPLIT_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define HR_SPLIT_KERNEL hr_mlp_f16f8_kernel
#define HR_SPLIT_LAUNCH hr_launch_mlp_f16f8
#define HR_SPLIT_PRODUCTS 5
#include "mlp_split_impl.inc"
