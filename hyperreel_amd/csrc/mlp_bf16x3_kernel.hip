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
// Structure (per workgroup: 32*MT rays, 4 waves):
//   * "swapped" GEMM: D[n][m] = sum_k W[n][k] X[m][k], weights as the A operand and rays as
//     the B operand.  In the 32x32 accumulator layout a lane then holds 4 consecutive output
//     features of ONE ray per register quad, so the epilogue packs them into one 8-byte LDS
//     store (hi) + one (lo), and the last layer stores 16-byte float4s;
//   * activations live in LDS only, already split: Xh/Xl[32*MT][W+8] bf16 (16-byte row pad ->
//     the 16 rows of a ds_read_b128 lane group fall on distinct bank slots);
//   * weights are split and tiled once by hr_model_finalize into the exact lane order of the
//     MFMA A operand: one coalesced 16-byte load per lane per tile, from L2.  A wave owns
//     64 output features (2 n-tiles) and ALL rays of the workgroup (MT m-tiles), so per 16-wide
//     k-step it issues 4 global loads + 2*MT ds_read_b128 for 6*MT MFMAs.  MT = 4 (128 rays,
//     one workgroup per CU, 128 accumulator VGPRs) halves the weight bytes that cross the
//     CU's vector L1 per MFMA relative to MT = 2 (64 rays, two workgroups per CU);
//   * weights run 3 k-steps ahead of their use in a 4-slot register ring, activations one.
#include <cstdlib>

#include "hr_kernels.h"
#include "hr_math.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int NT>
struct HrWOps {
    bf16x8 h[NT], l[NT];   // weights (A operand), hi/lo halves of NT n-tiles
};
template <int MT>
struct HrXOps {
    bf16x8 h[MT], l[MT];   // activations (B operand), hi/lo halves of MT m-tiles
};

template <int NT>
__device__ __forceinline__ void hr_load_w(HrWOps<NT>& o, const bf16x8* __restrict__ wp, int wkt, int tiles_total,
                                          const int (&tile)[NT], int lane)
{
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const size_t base = (((size_t)wkt * tiles_total + tile[nt]) * 2) * 64 + lane;
        o.h[nt] = wp[base];
        o.l[nt] = wp[base + 64];
    }
}

template <int MT>
__device__ __forceinline__ void hr_load_x(HrXOps<MT>& o, const __bf16* __restrict__ xh, const __bf16* __restrict__ xl,
                                          int stride, int kt, int lane)
{
    const int xoff = (lane & 31) * stride + kt * 16 + 8 * (lane >> 5);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        o.h[mt] = *reinterpret_cast<const bf16x8*>(xh + xoff + mt * 32 * stride);
        o.l[mt] = *reinterpret_cast<const bf16x8*>(xl + xoff + mt * 32 * stride);
    }
}

// 6*MT MFMAs of one k-step; products are the outer loop so that consecutive MFMAs write
// different accumulators (no back-to-back dependent issue)
template <int NT, int MT>
__device__ __forceinline__ void hr_mfma3(floatx16 (&acc)[NT][MT], const HrWOps<NT>& w, const HrXOps<MT>& x)
{
    // raised issue priority for the MFMA burst: the co-resident workgroup's epilogue/prologue
    // VALU work then fills the gaps instead of delaying the matrix pipe (measured: 1.483 -> 1.395 ms)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.l[nt], x.h[mt], acc[nt][mt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h[nt], x.l[mt], acc[nt][mt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h[nt], x.h[mt], acc[nt][mt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}

// acc += W[:, segment] * X[segment]^T over NKT (multiple of 4, compile-time) 16-wide k-steps.
// Weights come from L2 (several hundred cycles): a 4-slot register ring keeps them 3 k-steps
// ahead of their use; activations come from LDS and run one step ahead.
template <int NKT, int NT, int MT>
__device__ __forceinline__ void hr_accumulate3_pipe(floatx16 (&acc)[NT][MT], const __bf16* xh, const __bf16* xl, int stride,
                                                    const bf16x8* wp, int kt0, int tiles_total, const int (&tile)[NT], int lane)
{
    static_assert(NKT % 4 == 0 && NKT >= 4, "k-steps must come in fours");
    HrWOps<NT> w0, w1, w2, w3;
    HrXOps<MT> x0, x1;
    hr_load_w<NT>(w0, wp, kt0, tiles_total, tile, lane);
    hr_load_w<NT>(w1, wp, kt0 + 1, tiles_total, tile, lane);
    hr_load_w<NT>(w2, wp, kt0 + 2, tiles_total, tile, lane);
    hr_load_x<MT>(x0, xh, xl, stride, 0, lane);
#pragma unroll 1
    for (int kt = 0; kt < NKT - 4; kt += 4) {
        hr_load_w<NT>(w3, wp, kt0 + kt + 3, tiles_total, tile, lane);
        hr_load_x<MT>(x1, xh, xl, stride, kt + 1, lane);
        hr_mfma3<NT, MT>(acc, w0, x0);
        hr_load_w<NT>(w0, wp, kt0 + kt + 4, tiles_total, tile, lane);
        hr_load_x<MT>(x0, xh, xl, stride, kt + 2, lane);
        hr_mfma3<NT, MT>(acc, w1, x1);
        hr_load_w<NT>(w1, wp, kt0 + kt + 5, tiles_total, tile, lane);
        hr_load_x<MT>(x1, xh, xl, stride, kt + 3, lane);
        hr_mfma3<NT, MT>(acc, w2, x0);
        hr_load_w<NT>(w2, wp, kt0 + kt + 6, tiles_total, tile, lane);
        hr_load_x<MT>(x0, xh, xl, stride, kt + 4, lane);
        hr_mfma3<NT, MT>(acc, w3, x1);
    }
    // last four k-steps: nothing left to prefetch beyond NKT-1
    hr_load_w<NT>(w3, wp, kt0 + NKT - 1, tiles_total, tile, lane);
    hr_load_x<MT>(x1, xh, xl, stride, NKT - 3, lane);
    hr_mfma3<NT, MT>(acc, w0, x0);
    hr_load_x<MT>(x0, xh, xl, stride, NKT - 2, lane);
    hr_mfma3<NT, MT>(acc, w1, x1);
    hr_load_x<MT>(x1, xh, xl, stride, NKT - 1, lane);
    hr_mfma3<NT, MT>(acc, w2, x0);
    hr_mfma3<NT, MT>(acc, w3, x1);
}

// Same contraction with a ring of R weight slots (R-1 k-steps ahead), fully unrolled so that every slot
// index is a compile-time constant.  HR_MLP_RING selects it (A/B: tools/ab.sh "-DHR_MLP_RING=8" ...).
template <int NKT, int NT, int MT, int R>
__device__ __forceinline__ void hr_accumulate3_ring(floatx16 (&acc)[NT][MT], const __bf16* xh, const __bf16* xl, int stride,
                                                    const bf16x8* wp, int kt0, int tiles_total, const int (&tile)[NT], int lane)
{
    HrWOps<NT> w[R];
    HrXOps<MT> x[2];
#pragma unroll
    for (int i = 0; i < R - 1; ++i)
        if (i < NKT) hr_load_w<NT>(w[i], wp, kt0 + i, tiles_total, tile, lane);
    hr_load_x<MT>(x[0], xh, xl, stride, 0, lane);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        if (kt + R - 1 < NKT) hr_load_w<NT>(w[(kt + R - 1) % R], wp, kt0 + kt + R - 1, tiles_total, tile, lane);
        if (kt + 1 < NKT) hr_load_x<MT>(x[(kt + 1) & 1], xh, xl, stride, kt + 1, lane);
        hr_mfma3<NT, MT>(acc, w[kt % R], x[kt & 1]);
    }
}

#ifdef HR_MLP_RING
#define HR_ACCUMULATE_HIDDEN(NKT_, NT_, MT_, ...) hr_accumulate3_ring<NKT_, NT_, MT_, HR_MLP_RING>(__VA_ARGS__)
#else
#define HR_ACCUMULATE_HIDDEN(NKT_, NT_, MT_, ...) hr_accumulate3_pipe<NKT_, NT_, MT_>(__VA_ARGS__)
#endif

// Input segment (k0p/16 = 1..4 k-steps): short, no ring.
template <int NT, int MT>
__device__ __forceinline__ void hr_accumulate3(floatx16 (&acc)[NT][MT], const __bf16* xh, const __bf16* xl, int stride, int nkt,
                                               const bf16x8* wp, int kt0, int tiles_total, const int (&tile)[NT], int lane)
{
    for (int kt = 0; kt < nkt; ++kt) {
        HrWOps<NT> w;
        HrXOps<MT> x;
        hr_load_w<NT>(w, wp, kt0 + kt, tiles_total, tile, lane);
        hr_load_x<MT>(x, xh, xl, stride, kt, lane);
        hr_mfma3<NT, MT>(acc, w, x);
    }
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

// W: hidden width (256).  MT: 32-ray m-tiles per workgroup (2 -> 64 rays, 4 -> 128).  NW: waves
// per workgroup (4 or 8); a wave owns NT = 8 / NW tiles of 32 hidden features.
template <int W, int MT, int NW>
__global__ __launch_bounds__(64 * NW, (MT == 2) ? (NW / 2) : (NW / 4)) void hr_mlp_bf16x3_kernel(const hr_config cfg, const HrMlpArgs a)
{
    constexpr int TM = 32 * MT;           // rays per workgroup
    constexpr int XS = W + 8;             // bf16 elements per activation row
    constexpr int NT = (W / 32) / NW;     // hidden-layer tiles per wave
    constexpr int NTHREADS = 64 * NW;
    static_assert(W == 256 && (NT == 1 || NT == 2), "hidden width 256 with 4 or 8 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int k0p = a.k0p;
    const int XSI = k0p + 8;
    __bf16* Xih = reinterpret_cast<__bf16*>(lds_raw);       // [TM][k0p+8] MLP input, hi
    __bf16* Xil = Xih + TM * XSI;                            //              lo
    __bf16* Xh = Xil + TM * XSI;                             // [TM][W+8] hidden activations, hi
    __bf16* Xl = Xh + TM * XS;                               //            lo
    float* stage = reinterpret_cast<float*>(Xh);             // fp32 features, only before layer 0

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t ray0 = (int64_t)blockIdx.x * TM;

    // optional per-wave phase timeline (hr_debug_trace_mlp): s_memtime stamps, 64 slots per wave
    unsigned long long* tr = a.trace ? a.trace + ((size_t)blockIdx.x * NW + wave) * 64 : nullptr;
    int tri = 0;
#define HR_STAMP() do { if (tr && lane == 0 && tri < 64) tr[tri] = __builtin_readcyclecounter(); ++tri; } while (0)
    HR_STAMP();                                              // 0: start
#ifdef HR_MLP_DESYNC
    // experiment: the two workgroups that share a CU in the first dispatch round are blocks b and b + 256 (XCD = b % 8,
    // CU = (b / 8) % 32); delay the second one by HR_MLP_DESYNC x 3.4 us so that its GEMM phases meet the other's epilogues
    if ((blockIdx.x >> 8) & 1)
        for (int i = 0; i < HR_MLP_DESYNC; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    // ---- prologue: ray parameterisation + positional encoding (fp32), then split
    if (tid < TM) {
        const int64_t r = ray0 + tid;
        float* row = stage + tid * k0p;
        int n = 0;
        if (r < a.n_rays) n = hr_ray_features(cfg, a.rays + r * cfg.ray_dim, row);
        for (int i = n; i < k0p; ++i) row[i] = 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < TM * (k0p / 4); i += NTHREADS) {
        const int r = i / (k0p / 4), c4 = i - r * (k0p / 4);
        const float4 v = *reinterpret_cast<const float4*>(stage + r * k0p + 4 * c4);
        hr_split_store4(Xih, Xil, r * XSI + 4 * c4, v.x, v.y, v.z, v.w);
    }
    __syncthreads();
    HR_STAMP();                                              // 1: prologue done

    const int L = cfg.mlp_layers;
    // ---- hidden layers: wave w owns output features [w*W/NW, (w+1)*W/NW)
    for (int l = 0; l + 1 < L; ++l) {
        const bool skip = (cfg.mlp_skip_mask >> l) & 1;
        const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wsplit[l]);
        const int tiles_total = a.n_tiles[l];
        const float* bias = a.bias[l];
        floatx16 acc[NT][MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.0f;
        int tile[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) tile[nt] = wave * NT + nt;
#ifdef HR_MLP_BIAS_EARLY
        float4 bq[NT][4];                    // this lane's bias values, fetched under the GEMM instead of after it
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[nt][g] = *reinterpret_cast<const float4*>(bias + tile[nt] * 32 + 4 * (lane >> 5) + 8 * g);
#endif
        int kt0 = 0;
        if (l == 0 || skip) {
            hr_accumulate3<NT, MT>(acc, Xih, Xil, XSI, k0p / 16, wp, 0, tiles_total, tile, lane);
            kt0 = k0p / 16;
        }
        if (l > 0) HR_ACCUMULATE_HIDDEN(W / 16, NT, MT, acc, Xh, Xl, XS, wp, kt0, tiles_total, tile, lane);
        HR_STAMP();                          // 2+3l: GEMM of layer l issued
        __syncthreads();                     // all waves have finished reading Xh/Xl
        HR_STAMP();                          // 3+3l: barrier passed
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int nbase = tile[nt] * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = nbase + 8 * g;
#ifdef HR_MLP_BIAS_EARLY
                const float4 b = bq[nt][g];
#else
                const float4 b = *reinterpret_cast<const float4*>(bias + n0);
#endif
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float v0 = acc[nt][mt][4 * g + 0] + b.x;
                    float v1 = acc[nt][mt][4 * g + 1] + b.y;
                    float v2 = acc[nt][mt][4 * g + 2] + b.z;
                    float v3 = acc[nt][mt][4 * g + 3] + b.w;
                    v0 = (v0 > 0.0f) ? v0 : v0 * cfg.leaky_slope;   // nn.LeakyReLU(0.01), mlp.py:149-154
                    v1 = (v1 > 0.0f) ? v1 : v1 * cfg.leaky_slope;
                    v2 = (v2 > 0.0f) ? v2 : v2 * cfg.leaky_slope;
                    v3 = (v3 > 0.0f) ? v3 : v3 * cfg.leaky_slope;
                    hr_split_store4(Xh, Xl, (mt * 32 + (lane & 31)) * XS + n0, v0, v1, v2, v3);
                }
            }
        }
        __syncthreads();
        HR_STAMP();                              // 4+3l: epilogue + barrier done
    }

    // ---- last Linear: N = Z*P_live features in tiles of 32.  With 4 waves: full passes of 2 tiles
    //      per wave, and a remainder of up to 4 tiles as one tile per wave so that the waves stay
    //      balanced (DoNeRF after dead-column pruning: 11 tiles = 8 + 3).  With 8 waves: one tile per
    //      wave per pass.
    {
        const int l = L - 1;
        const bool skip = (cfg.mlp_skip_mask >> l) & 1;
        const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wsplit[l]);
        const int tiles_total = a.n_tiles[l];
        const float* bias = a.bias[l];
        auto store_tile = [&](int tile_n, const floatx16 (&acc)[MT]) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = tile_n * 32 + 8 * g + 4 * (lane >> 5);
                if (n0 >= a.n_out) continue;
                const float4 b = *reinterpret_cast<const float4*>(bias + n0);   // bias is padded to the tile
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int64_t row = ray0 + mt * 32 + (lane & 31);
                    if (row >= a.n_rays) continue;
                    float4 v;
                    v.x = acc[mt][4 * g + 0] + b.x;
                    v.y = acc[mt][4 * g + 1] + b.y;
                    v.z = acc[mt][4 * g + 2] + b.z;
                    v.w = acc[mt][4 * g + 3] + b.w;
                    // HQ layout: the 32 lanes of a half-wave write 512 contiguous bytes.  (Plain stores on
                    // purpose: non-temporal ones shave 4 % off this kernel but evict the head from the caches
                    // the sample kernel then reads it through: 2.90 vs 2.73 ms per frame end to end.)
                    *reinterpret_cast<float4*>(a.head + hr_head_index(row, n0, a.nq)) = v;
                }
            }
        };
        int t0 = 0;
        if constexpr (NT == 2) {
#pragma unroll 1
            for (; tiles_total - t0 > NW; t0 += 2 * NW) {              // two tiles per wave
                const int tile[2] = {t0 + wave * 2, t0 + wave * 2 + 1};
                if (tile[0] >= tiles_total) continue;                   // wave-uniform
                const int tile_ld[2] = {tile[0], min(tile[1], tiles_total - 1)};
                floatx16 acc[2][MT];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.0f;
                int kt0 = 0;
                if (skip) {
                    hr_accumulate3<2, MT>(acc, Xih, Xil, XSI, k0p / 16, wp, 0, tiles_total, tile_ld, lane);
                    kt0 = k0p / 16;
                }
                HR_ACCUMULATE_HIDDEN(W / 16, 2, MT, acc, Xh, Xl, XS, wp, kt0, tiles_total, tile_ld, lane);
                HR_STAMP();                      // last layer: GEMM of this pass issued
                store_tile(tile[0], acc[0]);
                if (tile[1] < tiles_total) store_tile(tile[1], acc[1]);
                HR_STAMP();                      // last layer: stores of this pass issued
            }
        }
#pragma unroll 1
        for (; t0 < tiles_total; t0 += NW) {                            // one tile per wave
            if (t0 + wave >= tiles_total) continue;
            const int tile[1] = {t0 + wave};
            floatx16 acc[1][MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][mt][r] = 0.0f;
            int kt0 = 0;
            if (skip) {
                hr_accumulate3<1, MT>(acc, Xih, Xil, XSI, k0p / 16, wp, 0, tiles_total, tile, lane);
                kt0 = k0p / 16;
            }
            HR_ACCUMULATE_HIDDEN(W / 16, 1, MT, acc, Xh, Xl, XS, wp, kt0, tiles_total, tile, lane);
            HR_STAMP();
            store_tile(tile[0], acc[0]);
            HR_STAMP();
        }
    }
#undef HR_STAMP
}

template <int W, int MT, int NW>
static void hr_launch_mlp_bf16x3_t(const hr_config& cfg, const HrMlpArgs& args, hipStream_t stream)
{
    constexpr int TM = 32 * MT;
    size_t lds = (size_t)TM * 2 * ((args.k0p + 8) + (W + 8)) * sizeof(__bf16);
    // experiment knob: HR_MLP_LDS_PAD=<KB> pads the allocation (e.g. 20 forces one workgroup per CU at 64 rays)
    static const size_t pad = [] { const char* e = getenv("HR_MLP_LDS_PAD"); return e ? (size_t)atoi(e) * 1024 : (size_t)0; }();
    lds += pad;
    const unsigned blocks = (unsigned)((args.n_rays + TM - 1) / TM);
    static size_t allowed = 0;
    if (lds > allowed) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hr_mlp_bf16x3_kernel<W, MT, NW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        allowed = lds;
    }
    hipLaunchKernelGGL((hr_mlp_bf16x3_kernel<W, MT, NW>), dim3(blocks), dim3(64 * NW), lds, stream, cfg, args);
}

void hr_launch_mlp_bf16x3(const hr_config& cfg, const HrMlpArgs& args, hipStream_t stream)
{
    if (args.n_rays <= 0) return;
    // rays per workgroup: 64 (default: two workgroups per CU hide each other's epilogues;
    // measured 1.50 ms per 640k rays) or 128 (one workgroup per CU, half the L1 weight traffic
    // but nothing to overlap with; 1.83 ms).  HR_MLP_TILE=128 selects the latter for A/B runs.
    static const int tile_m = [] {
        const char* e = getenv("HR_MLP_TILE");
        return (e && atoi(e) == 128) ? 128 : 64;
    }();
    // the 128-ray tile needs 2*128*((k0p+8)+(W+8))*2 bytes of LDS <= 160 KiB
    const bool fits128 = (size_t)128 * 2 * ((args.k0p + 8) + (cfg.mlp_hidden + 8)) * 2 <= 160 * 1024;
    if (cfg.mlp_hidden == 256) {   // other widths: rejected by hr_model_create
        static const int nwaves = [] { const char* e = getenv("HR_MLP_WAVES"); return (e && atoi(e) == 8) ? 8 : 4; }();
        if (tile_m == 128 && fits128) hr_launch_mlp_bf16x3_t<256, 4, 4>(cfg, args, stream);
        else if (nwaves == 8) hr_launch_mlp_bf16x3_t<256, 2, 8>(cfg, args, stream);
        else hr_launch_mlp_bf16x3_t<256, 2, 4>(cfg, args, stream);
    }
}
