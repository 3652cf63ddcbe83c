// One-time layout transforms run by hr_model_finalize (not on the render path).
#include "hr_kernels.h"

// Reference planes are channel-first (1, C, H, W) (nlf/nets/tensorf_base.py:911-948,
// nlf/nets/tensorf_dynamic.py:126-173).  The sample kernel wants channel-last texels with
// the density and appearance channels of one plane side by side:
//   dst[(y*W + x)*tex + c_off + c] = src[(c*H + y)*W + x]
__global__ void hr_interleave_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W,
                                     int tex, int c_off)
{
    const int64_t n = (int64_t)C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        // i enumerates the destination order (y, x, c) so that writes are contiguous per texel
        const int c = (int)(i % C);
        const int64_t yx = i / C;
        dst[yx * tex + c_off + c] = src[(int64_t)c * H * W + yx];
    }
}

// out[r][k*P + c] = head[r][k*P_live + col_map[c]] (0 where the column was pruned)
__global__ void hr_head_export_kernel(const float* __restrict__ head, float* __restrict__ out, int64_t n_rays, int Z, int P,
                                      int P_live, int nq, HrColMap map)
{
    const int n_out = Z * P;
    const int64_t total = n_rays * n_out;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n_out;
        const int n = (int)(i - r * n_out);
        const int k = n / P, c = n - k * P;
        const int cn = map.col[c];
        out[i] = (cn >= 0) ? head[hr_head_index(r, k * P_live + cn, nq)] : 0.0f;
    }
}

void hr_launch_head_export(const float* head, float* out, int64_t n_rays, int Z, int P, int P_live, int nq, const HrColMap& map,
                           hipStream_t stream)
{
    const int64_t total = n_rays * Z * P;
    if (total <= 0) return;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(hr_head_export_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, head, out, n_rays, Z, P, P_live, nq, map);
}

void hr_launch_interleave(const float* src, float* dst, int C, int H, int W, int tex, int c_off, hipStream_t stream)
{
    const int64_t n = (int64_t)C * H * W;
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(hr_interleave_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, C, H, W, tex, c_off);
}
