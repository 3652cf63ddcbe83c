// Training path on the device: hr_ray_train (hr_train.h) for one ray per thread, plus the layout
// kernels around it -- MLP input features for the caller's autograd MLP, and the packed texel gradients back to
// the reference's (C, H, W) parameter layout.
//
// Mapping: a training batch is 16 384 rays (conf/experiment/training/*.yaml: batch_size), i.e. 0.5 M samples against
// the 20 M of a rendered frame, so the step is latency- and atomics-bound rather than bandwidth-bound: one 64-thread
// workgroup per 64 rays spreads the batch over all 256 CUs (one wavefront each), every lane walks its ray's samples
// with the intermediates in registers / scratch, and texel gradients go straight to HBM through hardware fp32 atomics
// (global_atomic_add_f32) on the channel-last packed layout, where the 4..16 channels of one tap share a cache line.
#include "hr_kernels.h"
#include "hr_train.h"

template <int ZP>
__global__ __launch_bounds__(64) void hr_train_kernel(const hr_config* __restrict__ cfgp, const HrTrainArgs a)
{
    const int64_t ray = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (ray >= a.n_rays) return;
    hr_ray_train<ZP>(*cfgp, a, ray);
}

void hr_launch_train(const hr_config& cfg, const HrTrainArgs& args, hipStream_t stream)
{
    if (args.n_rays <= 0) return;
    int ZP = 8;
    while (ZP < cfg.z_channels) ZP <<= 1;
    const unsigned blocks = (unsigned)((args.n_rays + 63) / 64);
    switch (ZP) {
        case 8: hipLaunchKernelGGL(hr_train_kernel<8>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 16: hipLaunchKernelGGL(hr_train_kernel<16>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 32: hipLaunchKernelGGL(hr_train_kernel<32>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 64: hipLaunchKernelGGL(hr_train_kernel<64>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 128: hipLaunchKernelGGL(hr_train_kernel<128>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 256: hipLaunchKernelGGL(hr_train_kernel<256>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        default: break;
    }
}

// rays (n, ray_dim) -> MLP input features (n, mlp_in): ray parameterisation + positional encoding
// (nlf/param.py:87-115,244-253; nlf/pe.py:53-66,210-221), what RayPredictionEmbedding feeds its net (embedding/ray.py:316-330)
__global__ __launch_bounds__(256) void hr_features_kernel(const hr_config* __restrict__ cfgp, const float* __restrict__ rays, int64_t n,
                                                         float* __restrict__ out)
{
    const int64_t ray = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (ray >= n) return;
    const hr_config& c = *cfgp;
    float f[HR_MAX_MLP_IN];
    const int m = hr_ray_features(c, rays + ray * c.ray_dim, f);
    for (int i = 0; i < m; ++i) out[ray * c.mlp_in + i] = f[i];
}

void hr_launch_features(const hr_config* cfg_dev, const float* rays, int64_t n, float* out, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(hr_features_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, cfg_dev, rays, n, out);
}

// dst[c][y][x] = src[y][x][c_off + c]: packed texel gradients -> the reference's parameter layout
__global__ __launch_bounds__(256) void hr_deinterleave_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W, int tex,
                                                             int c_off)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // over (c, y, x), x fastest: coalesced stores
    const int64_t hw = (int64_t)H * W;
    if (i >= hw * C) return;
    const int c = (int)(i / hw);
    const int64_t yx = i - (int64_t)c * hw;
    dst[i] = src[yx * tex + c_off + c];
}

void hr_launch_deinterleave(const float* src, float* dst, int C, int H, int W, int tex, int c_off, hipStream_t stream)
{
    const int64_t n = (int64_t)C * H * W;
    if (n <= 0) return;
    hipLaunchKernelGGL(hr_deinterleave_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, dst, C, H, W, tex, c_off);
}
