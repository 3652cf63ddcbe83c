// K1 (split-precision form) -- the sample-prediction MLP with every fp32 GEMM evaluated as
// three bf16 MFMA products (reference: BaseMLP.forward, nlf/nets/mlp.py:159-172).
//
// Why: the fp32-input MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the bf16 MFMA rate, and
// the MLP is 97 % of the FLOPs of a frame.  Writing x = x_hi + x_lo and w = w_hi + w_lo with
// bf16 halves (round-to-nearest-even, lo = bf16(x - hi)) gives
//        x*w = x_hi*w_hi + x_hi*w_lo + x_lo*w_hi + O(2^-18 |x w|)
// and every partial product is exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16, so
// three bf16 MFMAs reproduce the fp32 GEMM to ~2^-17 relative per product -- measured: raw
// head within 8e-6 (relative to max |head|) of the fp32 chain and RGB within 1e-5 of the
// reference on every model family (tests/test_gpu_parity.py), i.e. >= 10x inside the 1e-4 bar
// -- at 3/16 of the fp32-MFMA issue time.
//
// Structure (per workgroup: 64 rays, 4 waves, 2 workgroups per CU):
//   * "swapped" GEMM: D[n][m] = sum_k W[n][k] X[m][k], weights as the A operand and rays as
//     the B operand.  In the 32x32 accumulator layout a lane then holds 4 consecutive output
//     features of ONE ray per register quad, so the epilogue packs them into one 8-byte LDS
//     store (hi) + one (lo), and the last layer stores 16-byte float4s;
//   * activations live in LDS only, already split: Xh/Xl[64][W+8] bf16 (16-byte row pad ->
//     the 16 rows of a ds_read_b128 lane group fall on distinct bank slots);
//   * weights are split and tiled once by hr_model_finalize into the exact lane order of the
//     MFMA A operand: one coalesced 16-byte load per lane per tile, from L2;
//   * per 16-wide k-step a wave issues 4 global loads + 4 ds_read_b128 for 12 MFMAs
//     (2 n-tiles x 2 m-tiles x 3 products); the next k-step's operands are prefetched into a
//     second register set before the current MFMAs issue.
#include "hr_kernels.h"
#include "hr_math.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define HR_TILE_M 64

struct HrOperands {
    bf16x8 wh[2], wl[2];   // weights (A operand), 2 n-tiles
    bf16x8 xh[2], xl[2];   // activations (B operand), 2 m-tiles
};

__device__ __forceinline__ void hr_load_operands(HrOperands& o, const __bf16* __restrict__ xh, const __bf16* __restrict__ xl,
                                                 int stride, int kt, const bf16x8* __restrict__ wp, int wkt,
                                                 int tiles_total, const int (&tile)[2], int lane)
{
    const int xoff = (lane & 31) * stride + kt * 16 + 8 * (lane >> 5);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        o.xh[mt] = *reinterpret_cast<const bf16x8*>(xh + xoff + mt * 32 * stride);
        o.xl[mt] = *reinterpret_cast<const bf16x8*>(xl + xoff + mt * 32 * stride);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const size_t base = (((size_t)wkt * tiles_total + tile[nt]) * 2) * 64 + lane;
        o.wh[nt] = wp[base];
        o.wl[nt] = wp[base + 64];
    }
}

__device__ __forceinline__ void hr_mfma3(floatx16 (&acc)[2][2], const HrOperands& o)
{
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.wl[nt], o.xh[mt], acc[nt][mt], 0, 0, 0);
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.wh[nt], o.xl[mt], acc[nt][mt], 0, 0, 0);
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.wh[nt], o.xh[mt], acc[nt][mt], 0, 0, 0);
        }
}

// acc += W[:, segment] * X[segment]^T over nkt 16-wide k-steps, software-pipelined one step deep
__device__ __forceinline__ void hr_accumulate3(floatx16 (&acc)[2][2], const __bf16* xh, const __bf16* xl, int stride, int nkt,
                                               const bf16x8* wp, int kt0, int tiles_total, const int (&tile)[2], int lane)
{
    HrOperands cur, nxt;
    hr_load_operands(cur, xh, xl, stride, 0, wp, kt0, tiles_total, tile, lane);
    for (int kt = 0; kt + 1 < nkt; ++kt) {
        hr_load_operands(nxt, xh, xl, stride, kt + 1, wp, kt0 + kt + 1, tiles_total, tile, lane);
        hr_mfma3(acc, cur);
        cur = nxt;
    }
    hr_mfma3(acc, cur);
}

__device__ __forceinline__ void hr_split_store4(__bf16* xh, __bf16* xl, int idx, float v0, float v1, float v2, float v3)
{
    bf16x4 h, l;
    h[0] = (__bf16)v0; h[1] = (__bf16)v1; h[2] = (__bf16)v2; h[3] = (__bf16)v3;
    l[0] = (__bf16)(v0 - (float)h[0]);
    l[1] = (__bf16)(v1 - (float)h[1]);
    l[2] = (__bf16)(v2 - (float)h[2]);
    l[3] = (__bf16)(v3 - (float)h[3]);
    *reinterpret_cast<bf16x4*>(xh + idx) = h;
    *reinterpret_cast<bf16x4*>(xl + idx) = l;
}

template <int W>
__global__ __launch_bounds__(256, 2) void hr_mlp_bf16x3_kernel(const hr_config cfg, const HrMlpArgs a)
{
    constexpr int XS = W + 8;             // bf16 elements per activation row
    constexpr int NTW = W / 256;          // passes of 2 x 32 output features per wave in hidden layers
    static_assert(W % 256 == 0, "hidden width must be a multiple of 256");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int k0p = a.k0p;
    const int XSI = k0p + 8;
    __bf16* Xih = reinterpret_cast<__bf16*>(lds_raw);       // [64][k0p+8] MLP input, hi
    __bf16* Xil = Xih + HR_TILE_M * XSI;                     //              lo
    __bf16* Xh = Xil + HR_TILE_M * XSI;                      // [64][W+8] hidden activations, hi
    __bf16* Xl = Xh + HR_TILE_M * XS;                        //            lo
    float* stage = reinterpret_cast<float*>(Xh);             // fp32 features, only before layer 0

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t ray0 = (int64_t)blockIdx.x * HR_TILE_M;

    // ---- prologue: ray parameterisation + positional encoding (fp32), then split
    if (tid < HR_TILE_M) {
        const int64_t r = ray0 + tid;
        float* row = stage + tid * k0p;
        int n = 0;
        if (r < a.n_rays) n = hr_ray_features(cfg, a.rays + r * cfg.ray_dim, row);
        for (int i = n; i < k0p; ++i) row[i] = 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < HR_TILE_M * (k0p / 4); i += 256) {
        const int r = i / (k0p / 4), c4 = i - r * (k0p / 4);
        const float4 v = *reinterpret_cast<const float4*>(stage + r * k0p + 4 * c4);
        hr_split_store4(Xih, Xil, r * XSI + 4 * c4, v.x, v.y, v.z, v.w);
    }
    __syncthreads();

    const int L = cfg.mlp_layers;
    // ---- hidden layers: wave w owns output features [w*W/4, (w+1)*W/4) in NTW passes of 2 tiles
    for (int l = 0; l + 1 < L; ++l) {
        const bool skip = (cfg.mlp_skip_mask >> l) & 1;
        const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wsplit[l]);
        const int tiles_total = a.n_tiles[l];
        floatx16 acc[NTW][2][2];
#pragma unroll
        for (int p = 0; p < NTW; ++p) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[p][nt][mt][r] = 0.0f;
            const int tile[2] = {wave * (2 * NTW) + 2 * p, wave * (2 * NTW) + 2 * p + 1};
            int kt0 = 0;
            if (l == 0 || skip) {
                hr_accumulate3(acc[p], Xih, Xil, XSI, k0p / 16, wp, 0, tiles_total, tile, lane);
                kt0 = k0p / 16;
            }
            if (l > 0) hr_accumulate3(acc[p], Xh, Xl, XS, W / 16, wp, kt0, tiles_total, tile, lane);
        }
        __syncthreads();  // all waves have finished reading Xh/Xl
        const float* bias = a.bias[l];
#pragma unroll
        for (int p = 0; p < NTW; ++p)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int nbase = (wave * (2 * NTW) + 2 * p + nt) * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n0 = nbase + 8 * g;
                    const float4 b = *reinterpret_cast<const float4*>(bias + n0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        float v0 = acc[p][nt][mt][4 * g + 0] + b.x;
                        float v1 = acc[p][nt][mt][4 * g + 1] + b.y;
                        float v2 = acc[p][nt][mt][4 * g + 2] + b.z;
                        float v3 = acc[p][nt][mt][4 * g + 3] + b.w;
                        v0 = (v0 > 0.0f) ? v0 : v0 * cfg.leaky_slope;   // nn.LeakyReLU(0.01), mlp.py:149-154
                        v1 = (v1 > 0.0f) ? v1 : v1 * cfg.leaky_slope;
                        v2 = (v2 > 0.0f) ? v2 : v2 * cfg.leaky_slope;
                        v3 = (v3 > 0.0f) ? v3 : v3 * cfg.leaky_slope;
                        hr_split_store4(Xh, Xl, (mt * 32 + (lane & 31)) * XS + n0, v0, v1, v2, v3);
                    }
                }
            }
        __syncthreads();
    }

    // ---- last Linear: N = Z*P features in passes of 4 waves x 2 tiles of 32
    {
        const int l = L - 1;
        const bool skip = (cfg.mlp_skip_mask >> l) & 1;
        const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wsplit[l]);
        const int tiles_total = a.n_tiles[l];
        const float* bias = a.bias[l];
        for (int t0 = 0; t0 < tiles_total; t0 += 8) {
            const int tile[2] = {t0 + wave * 2, t0 + wave * 2 + 1};
            if (tile[0] >= tiles_total) continue;                       // wave-uniform
            const int tile_ld[2] = {tile[0], min(tile[1], tiles_total - 1)};
            floatx16 acc[2][2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.0f;
            int kt0 = 0;
            if (skip) {
                hr_accumulate3(acc, Xih, Xil, XSI, k0p / 16, wp, 0, tiles_total, tile_ld, lane);
                kt0 = k0p / 16;
            }
            hr_accumulate3(acc, Xh, Xl, XS, W / 16, wp, kt0, tiles_total, tile_ld, lane);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (tile[nt] >= tiles_total) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n0 = tile[nt] * 32 + 8 * g + 4 * (lane >> 5);
                    if (n0 >= a.n_out) continue;
                    const float4 b = *reinterpret_cast<const float4*>(bias + n0);   // bias is padded to the tile
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int64_t row = ray0 + mt * 32 + (lane & 31);
                        if (row >= a.n_rays) continue;
                        float4 v;
                        v.x = acc[nt][mt][4 * g + 0] + b.x;
                        v.y = acc[nt][mt][4 * g + 1] + b.y;
                        v.z = acc[nt][mt][4 * g + 2] + b.z;
                        v.w = acc[nt][mt][4 * g + 3] + b.w;
                        float* dst = a.head + row * a.n_out + n0;
                        if (n0 + 3 < a.n_out && (a.n_out & 3) == 0) {
                            *reinterpret_cast<float4*>(dst) = v;
                        } else {
                            dst[0] = v.x;
                            if (n0 + 1 < a.n_out) dst[1] = v.y;
                            if (n0 + 2 < a.n_out) dst[2] = v.z;
                            if (n0 + 3 < a.n_out) dst[3] = v.w;
                        }
                    }
                }
            }
        }
    }
}

template <int W>
static void hr_launch_mlp_bf16x3_w(const hr_config& cfg, const HrMlpArgs& args, unsigned blocks, size_t lds, hipStream_t stream)
{
    static size_t allowed = 0;
    if (lds > allowed) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hr_mlp_bf16x3_kernel<W>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        allowed = lds;
    }
    hipLaunchKernelGGL(hr_mlp_bf16x3_kernel<W>, dim3(blocks), dim3(256), lds, stream, cfg, args);
}

void hr_launch_mlp_bf16x3(const hr_config& cfg, const HrMlpArgs& args, hipStream_t stream)
{
    if (args.n_rays <= 0) return;
    const int W = cfg.mlp_hidden;
    const size_t lds = (size_t)HR_TILE_M * 2 * ((args.k0p + 8) + (W + 8)) * sizeof(__bf16);
    const unsigned blocks = (unsigned)((args.n_rays + HR_TILE_M - 1) / HR_TILE_M);
    if (W == 256) hr_launch_mlp_bf16x3_w<256>(cfg, args, blocks, lds, stream);   // other widths: rejected by hr_model_create
}
