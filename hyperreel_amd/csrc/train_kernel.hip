// Training path on the device: the three phases of hr_train.h, plus the layout kernels around them --
// MLP input features for the caller's autograd MLP, and the packed texel gradients back to the reference's (C, H, W)
// parameter layout.
//
// Mapping: a training batch is 16 384 rays (conf/experiment/training/*.yaml: batch_size), i.e. 0.5 M samples against
// the 20 M of a rendered frame, so the step is latency- and atomics-bound rather than bandwidth-bound.
//   phase A (forward, compositing backward): one thread per ray, 64-thread workgroups -> one wavefront on each of the
//     256 CUs; the ray's samples are walked with the intermediates in registers / scratch;
//   phase B (gather backward): 16 lanes per (ray, sample), ONE TEXEL CHANNEL PER LANE: texel gradients go straight to
//     HBM through hardware fp32 atomics (global_atomic_add_f32) on the channel-last packed layout, and the channels of a
//     tap are one contiguous 32..64-byte run per instruction.  The memory system retires atomics per cache-line request,
//     not per lane: tools/atomic_ubench.hip measures 331 G atomics/s this way against 19.5 G/s for 64 lanes on 64
//     different lines, and the phase measured 12.2 ms (one thread per ray) and 10.4 ms (one thread per sample, channels
//     in a loop) per DoNeRF step before.  The ray's decode matrix and its gradient live in LDS (ds_add_f32);
//   phase C (intersection backward): one thread per (ray, sample), elementwise.
#include "hr_kernels.h"
#include "hr_mask.h"
#include "hr_train.h"

template <int ZP>
__global__ __launch_bounds__(64) void hr_train_kernel(const hr_config* __restrict__ cfgp, const HrTrainArgs a)
{
    const int64_t ray = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (ray >= a.n_rays) return;
    hr_ray_train<ZP>(*cfgp, a, ray);
}

// Phase B.  HR_TRAIN_LPS = 16 adjacent lanes per sample, one texel channel each (a plane pair has 8 or 16 channels per
// texel in every shipped model), so that the atomics of one tap are one contiguous run; a workgroup is 16 such groups
// and walks the samples of RPB whole rays (1 ray when it has 16 samples or more).
#define HR_TRAIN_LPS 16
template <int ZP>
__global__ __launch_bounds__(256) void hr_train_gather_bwd_kernel(const hr_config* __restrict__ cfgp, const HrTrainArgs a)
{
    const hr_config& c = *cfgp;
    constexpr int GROUPS = 256 / HR_TRAIN_LPS;
    constexpr int RPB = (ZP >= GROUPS) ? 1 : GROUPS / ZP;
    extern __shared__ float lds[];                 // [RPB][3 * CA] decode matrix, then [RPB][3 * CA] its gradient
    const int CA = a.ca_total, Z = c.z_channels;
    const int64_t ray0 = (int64_t)blockIdx.x * RPB;
    for (int e = threadIdx.x; e < RPB * 3 * CA; e += 256) {
        const int r = e / (3 * CA), i = e - r * 3 * CA;
        float v = 0.0f;
        if (ray0 + r < a.n_rays) {
            float sh[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const float* rr = a.rays + (ray0 + r) * c.ray_dim;
            if (c.shading == HR_SHADING_SH) hr_sh_deg2(rr[3], rr[4], rr[5], sh);
            v = hr_train_decode_coef(c, a, sh, i / CA, i % CA);
        }
        lds[e] = v;
        lds[RPB * 3 * CA + e] = 0.0f;
    }
    __syncthreads();
    const int grp = threadIdx.x / HR_TRAIN_LPS, lane = threadIdx.x % HR_TRAIN_LPS;
    for (int si = grp; si < RPB * Z; si += GROUPS) {
        const int r = si / Z, k = si - r * Z;
        if (ray0 + r >= a.n_rays) continue;
        hr_sample_train_bwd(c, a, ray0 + r, k, lds + r * 3 * CA, lds + (RPB + r) * 3 * CA, lane, HR_TRAIN_LPS);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < RPB * 3 * CA; e += 256) {
        const int r = e / (3 * CA), i = e - r * 3 * CA;
        if (ray0 + r >= a.n_rays) continue;
        float sh[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* rr = a.rays + (ray0 + r) * c.ray_dim;
        if (c.shading == HR_SHADING_SH) hr_sh_deg2(rr[3], rr[4], rr[5], sh);
        hr_train_fold_basis(c, a, sh, i / CA, i % CA, lds[RPB * 3 * CA + e]);
    }
}

// Phase C
__global__ __launch_bounds__(256) void hr_train_dist_bwd_kernel(const hr_config* __restrict__ cfgp, const HrTrainArgs a)
{
    const hr_config& c = *cfgp;
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_rays * c.z_channels) return;
    hr_sample_train_dist_bwd(c, a, s / c.z_channels, (int)(s % c.z_channels));
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
    if (!args.d_rgb) return;
    const int GROUPS = 256 / HR_TRAIN_LPS;
    const int RPB = (ZP >= GROUPS) ? 1 : GROUPS / ZP;
    const unsigned bblocks = (unsigned)((args.n_rays + RPB - 1) / RPB);
    const size_t lds = sizeof(float) * 2 * RPB * 3 * args.ca_total;
    switch (ZP) {
        case 8: hipLaunchKernelGGL(hr_train_gather_bwd_kernel<8>, dim3(bblocks), dim3(256), lds, stream, args.cfg_dev, args); break;
        case 16: hipLaunchKernelGGL(hr_train_gather_bwd_kernel<16>, dim3(bblocks), dim3(256), lds, stream, args.cfg_dev, args); break;
        case 32: hipLaunchKernelGGL(hr_train_gather_bwd_kernel<32>, dim3(bblocks), dim3(256), lds, stream, args.cfg_dev, args); break;
        case 64: hipLaunchKernelGGL(hr_train_gather_bwd_kernel<64>, dim3(bblocks), dim3(256), lds, stream, args.cfg_dev, args); break;
        case 128: hipLaunchKernelGGL(hr_train_gather_bwd_kernel<128>, dim3(bblocks), dim3(256), lds, stream, args.cfg_dev, args); break;
        case 256: hipLaunchKernelGGL(hr_train_gather_bwd_kernel<256>, dim3(bblocks), dim3(256), lds, stream, args.cfg_dev, args); break;
        default: break;
    }
    const int64_t ns = args.n_rays * cfg.z_channels;
    hipLaunchKernelGGL(hr_train_dist_bwd_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream, args.cfg_dev, args);
}

// Coarse level of a point_prediction cascade (hr_ray_rows ... in hr_train.h): rows forward per ray, then per sample the
// point backward and the intersection backward.  Elementwise work, no gather.
template <int ZP>
__global__ __launch_bounds__(64) void hr_rows_kernel(const hr_config* __restrict__ cfgp, const HrRowsArgs a)
{
    const int64_t ray = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (ray >= a.n_rays) return;
    hr_ray_rows<ZP>(*cfgp, a, ray);
}

__global__ __launch_bounds__(256) void hr_rows_bwd_kernel(const hr_config* __restrict__ cfgp, const HrRowsArgs a, int phase)
{
    const hr_config& c = *cfgp;
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_rays * c.z_channels) return;
    if (phase == 0) hr_sample_rows_bwd(c, a, s / c.z_channels, (int)(s % c.z_channels));
    else hr_sample_rows_dist_bwd(c, a, s / c.z_channels, (int)(s % c.z_channels));
}

void hr_launch_rows(const hr_config& cfg, const HrRowsArgs& args, hipStream_t stream)
{
    if (args.n_rays <= 0) return;
    int ZP = 8;
    while (ZP < cfg.z_channels) ZP <<= 1;
    const unsigned blocks = (unsigned)((args.n_rays + 63) / 64);
    switch (ZP) {
        case 8: hipLaunchKernelGGL(hr_rows_kernel<8>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 16: hipLaunchKernelGGL(hr_rows_kernel<16>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 32: hipLaunchKernelGGL(hr_rows_kernel<32>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 64: hipLaunchKernelGGL(hr_rows_kernel<64>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 128: hipLaunchKernelGGL(hr_rows_kernel<128>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 256: hipLaunchKernelGGL(hr_rows_kernel<256>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        default: break;
    }
    if (!args.d_rows) return;
    const int64_t ns = args.n_rays * cfg.z_channels;
    const unsigned sb = (unsigned)((ns + 255) / 256);
    hipLaunchKernelGGL(hr_rows_bwd_kernel, dim3(sb), dim3(256), 0, stream, args.cfg_dev, args, 0);
    hipLaunchKernelGGL(hr_rows_bwd_kernel, dim3(sb), dim3(256), 0, stream, args.cfg_dev, args, 1);
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

// Occupancy of the grids (hr_mask.h): one thread per lattice point, z fastest -- neighbouring threads read neighbouring
// line texels and the same plane texel rows.  8 M points (200^3) x 16 density channels x 6 taps: a few hundred microseconds,
// twice per training run (update_AlphaMask_list).
__global__ __launch_bounds__(256) void hr_dense_alpha_kernel(const hr_config* __restrict__ cfgp, const HrMaskArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n12 = (int64_t)a.n[1] * a.n[2];
    if (i >= n12 * a.n[0]) return;
    const int x = (int)(i / n12), y = (int)((i - (int64_t)x * n12) / a.n[2]), z = (int)(i % a.n[2]);
    a.alpha[i] = hr_point_alpha(*cfgp, a, x, y, z);
}

void hr_launch_dense_alpha(const HrMaskArgs& args, hipStream_t stream)
{
    const int64_t n = (int64_t)args.n[0] * args.n[1] * args.n[2];
    if (n <= 0) return;
    hipLaunchKernelGGL(hr_dense_alpha_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, args.cfg_dev, args);
}
