// Activation range of the sample-prediction MLP (BaseMLP.forward, nlf/nets/mlp.py:127-172) on a set of rays: the largest
// |input feature| and the largest |pre-activation| of every hidden Linear, evaluated in plain fp32 from the uploaded
// (reference-layout) weights.  hr_model_finalize / hr_model_calibrate use it to decide whether the fp16 split arithmetic
// (f16x3 / f16x2: hidden activations and input features are stored as IEEE halves, |x| < 65504) may be used for a model.
// Not a hot path: 4096 rays x 0.8 MFLOP, one workgroup per 8 rays, one output feature per thread.
#include "hr_kernels.h"
#include "hr_math.h"

constexpr int HR_RANGE_RAYS = 8;            // rays per workgroup
constexpr int HR_RANGE_MAXK = 512;          // widest hidden layer this kernel stages

__global__ __launch_bounds__(256) void hr_mlp_range_kernel(const hr_config cfg, const HrRangeArgs a)
{
    __shared__ float s_in[HR_RANGE_RAYS][256];                 // the MLP input (mlp_in <= 256)
    __shared__ float s_x[2][HR_RANGE_RAYS][HR_RANGE_MAXK];     // activations, ping-pong
    __shared__ float s_max[HR_MAX_LAYERS + 1];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * HR_RANGE_RAYS;
    if (tid <= HR_MAX_LAYERS) s_max[tid] = 0.0f;
    if (tid < HR_RANGE_RAYS) {
        for (int i = 0; i < cfg.mlp_in; ++i) s_in[tid][i] = 0.0f;
        if (r0 + tid < a.n_rays) hr_ray_features(cfg, a.rays + (r0 + tid) * cfg.ray_dim, s_in[tid]);
    }
    __syncthreads();
    float mx = 0.0f;
    for (int i = tid; i < HR_RANGE_RAYS * cfg.mlp_in; i += 256) mx = fmaxf(mx, fabsf(s_in[i / cfg.mlp_in][i % cfg.mlp_in]));
    atomicMax(reinterpret_cast<unsigned*>(&s_max[0]), __float_as_uint(mx));       // non-negative floats order like their bit patterns
    const int L = cfg.mlp_layers, W = cfg.mlp_hidden;
    int cur = 0;
    for (int l = 0; l + 1 < L; ++l) {                          // the last Linear's output (the head) stays fp32 in every mode
        const bool skip = (cfg.mlp_skip_mask >> l) & 1;
        const int K = (l == 0) ? cfg.mlp_in : (skip ? cfg.mlp_in + W : W);
        const float* wl = a.w[l];
        const float* bl = a.b[l];
        mx = 0.0f;
        for (int n = tid; n < W; n += 256) {
            float acc[HR_RANGE_RAYS];
            for (int r = 0; r < HR_RANGE_RAYS; ++r) acc[r] = bl[n];
            const float* wr = wl + (size_t)n * K;
            for (int k = 0; k < K; ++k) {
                const float w = wr[k];
                for (int r = 0; r < HR_RANGE_RAYS; ++r) {
                    float x;
                    if (l == 0) x = s_in[r][k];
                    else if (skip) x = (k < cfg.mlp_in) ? s_in[r][k] : s_x[cur][r][k - cfg.mlp_in];      // cat([input, x]), mlp.py:166-168
                    else x = s_x[cur][r][k];
                    acc[r] = fmaf(w, x, acc[r]);
                }
            }
            for (int r = 0; r < HR_RANGE_RAYS; ++r) {
                mx = fmaxf(mx, fabsf(acc[r]));
                s_x[cur ^ 1][r][n] = (acc[r] > 0.0f) ? acc[r] : acc[r] * cfg.leaky_slope;
            }
        }
        atomicMax(reinterpret_cast<unsigned*>(&s_max[l + 1]), __float_as_uint(mx));
        __syncthreads();
        cur ^= 1;
    }
    __syncthreads();
    if (tid < L && s_max[tid] > 0.0f) atomicMax(reinterpret_cast<unsigned*>(a.act_max + tid), __float_as_uint(s_max[tid]));
}

// rays of no particular camera: origins uniform in the box `lo..hi`, unit directions uniform on the sphere, the trailing
// columns (camera id, time) uniform in [0, 1) -- what hr_model_finalize calibrates on when it has no real rays
__global__ void hr_synthetic_rays_kernel(float* rays, int64_t n, int ray_dim, float3 lo, float3 hi, unsigned seed)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto rnd = [&](unsigned j) {                                // a hash, not a generator: reproducible per (ray, column)
        unsigned x = (unsigned)i * 0x9E3779B9u + j * 0x85EBCA6Bu + seed;
        x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
        return (float)(x >> 8) * (1.0f / 16777216.0f);
    };
    float* r = rays + i * ray_dim;
    r[0] = lo.x + (hi.x - lo.x) * rnd(0);
    r[1] = lo.y + (hi.y - lo.y) * rnd(1);
    r[2] = lo.z + (hi.z - lo.z) * rnd(2);
    const float z = 2.0f * rnd(3) - 1.0f, phi = 6.2831853f * rnd(4), s = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    r[3] = s * cosf(phi); r[4] = s * sinf(phi); r[5] = z;
    for (int c = 6; c < ray_dim; ++c) r[c] = rnd(5 + c);
}

void hr_launch_mlp_range(const hr_config& cfg, const HrRangeArgs& a, hipStream_t stream)
{
    if (a.n_rays <= 0 || cfg.mlp_layers == 0) return;
    const unsigned blocks = (unsigned)((a.n_rays + HR_RANGE_RAYS - 1) / HR_RANGE_RAYS);
    hipLaunchKernelGGL(hr_mlp_range_kernel, dim3(blocks), dim3(256), 0, stream, cfg, a);
}

bool hr_mlp_range_supported(const hr_config& cfg)
{
    return cfg.mlp_in <= 256 && cfg.mlp_hidden <= HR_RANGE_MAXK;
}

void hr_launch_synthetic_rays(float* rays, int64_t n, int ray_dim, const float lo[3], const float hi[3], unsigned seed, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(hr_synthetic_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rays, n, ray_dim,
                       make_float3(lo[0], lo[1], lo[2]), make_float3(hi[0], hi[1], hi[2]), seed);
}
