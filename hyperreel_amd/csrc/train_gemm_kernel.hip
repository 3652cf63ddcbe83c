// GEMMs of the sample-prediction MLP in TRAINING: what torch.autograd + addmm do for BaseMLP.forward
// (nlf/nets/mlp.py:159-172) under INRSystem.training_step (nlf/__init__.py:634-709) -- forward  y = leaky(x W^T + b),
// and for the backward  dx = dy' W,  dW = dy'^T x,  db = sum_rows dy'  with dy' = dy * leaky'(y).
//
// One kernel, three operand layouts.  C[M,N] = sum_k A(m,k) B(k,n) with fp32 operands in global memory; a 64x64 output tile
// per workgroup (4 wavefronts, a 32x32 MFMA tile each), K in steps of 16.  Each step the workgroup stages a 64x16 slice of
// A and a 16x64 slice of B through registers into LDS, SPLIT on the way into bf16 hi / lo halves (hi = bf16(v), lo =
// bf16(v - hi)) in the lane order of v_mfma_f32_32x32x16_bf16's operands, and every wavefront issues the three products
// hi*hi + hi*lo + lo*hi with fp32 accumulation -- the same arithmetic as the inference kernels' bf16x3 mode (relative
// error ~2^-17 per product; bf16 halves on purpose: gradients span the fp32 exponent range, fp16 halves would flush them).
//   forward  (NT): A = x (k contiguous),            B(k,n) = W[n][k] (k contiguous), epilogue + bias, LeakyReLU
//   dgrad    (NN): A = dy' (k contiguous),          B(k,n) = W[k][n] (n contiguous)
//   wgrad    (TN): A(m,k) = dy'[k][m] (m contiguous), B(k,n) = x[k][n] (n contiguous), K = the batch: split over
//                  blockIdx.z into partial tiles that a second kernel adds in a fixed order (deterministic, no atomics);
//                  the workgroups of the first n-tile also accumulate the row sums of A = db.
// dy' is formed while A is staged: dy * (y > 0 ? 1 : slope) -- the LeakyReLU mask costs no pass over memory.
// The FORWARD uses a three-way split (hi + mid + lo = 24 mantissa bits) and six products (hh, hm, mh, mm, hl, lh): its head
// feeds the threshold decisions of the sample stage (`dist <= near`), and a sample that flips against the fp32 reference
// changes a ray's whole gradient; the backward GEMMs are linear in continuous quantities and keep three products.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hr_kernels.h"

typedef float hr_acc16 __attribute__((ext_vector_type(16)));
typedef __bf16 hr_bf8 __attribute__((ext_vector_type(8)));

struct HrGemmArgs {
    const float* A; int64_t sam, sak;        // element (m, k) at A[m * sam + k * sak]
    const float* B; int64_t sbk, sbn;        // element (k, n) at B[k * sbk + n * sbn]
    const float* mask; int64_t smm, smk;     // optional: A(m,k) *= (mask(m,k) > 0 ? 1 : slope)   (same index order as A)
    float slope;
    float* C; int64_t ldc;                   // row-major output (M, N); wgrad: partial z at C + z * M * ldc
    const float* bias;                       // forward epilogue (N) or NULL
    int act;                                 // forward epilogue: LeakyReLU with `slope` when 1
    float* rowsum;                           // wgrad: partial row sums of A, (splits, M) or NULL
    int M, N, K;
    int k_per_split;                         // multiple of HR_GK; gridDim.z splits
};

constexpr int HR_GT = 64;                    // tile edge
constexpr int HR_GK = 32;                    // contraction elements staged per step (two MFMA k-steps)
constexpr int HR_GS = HR_GK + 8;             // bf16 elements per LDS row (+ 8: the rows of a ds_read_b128 group fall on distinct banks)

// eight consecutive elements along the contiguous dimension of an operand: two 16-byte loads when the run is whole and
// aligned, element loads (zero beyond the edge) otherwise
__device__ __forceinline__ void hr_gemm_load8(const float* p, int valid, bool vec, float (&v)[8])
{
    if (vec && valid >= 8) {
        const float4 lo = *reinterpret_cast<const float4*>(p), hi = *reinterpret_cast<const float4*>(p + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    } else if (valid >= 8) {                             // whole but not 16-byte aligned (a skip layer's input row starts at column 2): no predicate per element
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p[e];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (e < valid) ? p[e] : 0.0f;
    }
}

// eight elements `stride` apart (an operand that is contiguous along its ROW index is read along k this way: the 64 lanes of a wavefront
// take 64 consecutive rows, so every one of the eight loads is a coalesced 256-byte row of the matrix)
__device__ __forceinline__ void hr_gemm_load8s(const float* p, int64_t stride, int valid, float (&v)[8])
{
    if (valid >= 8) {                                     // the common case without a predicate per element: eight loads in flight
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p[(int64_t)e * stride];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (e < valid) ? p[(int64_t)e * stride] : 0.0f;
    }
}

// A_MC: A is contiguous along m (else along k).  B_NC: B is contiguous along n (else along k).
// SIX: the six-product form (operands split three ways).
// Per step a workgroup stages a 64 x 32 slice of A and of B.  EVERY thread owns 8 consecutive k of one row of each operand -- read as two
// 16-byte loads where the operand is contiguous along k (row = tid / 4, k = 8 (tid % 4) ..) and as eight coalesced element loads where it is
// contiguous along the row index (row = tid % 64, k = 8 (tid / 64) ..) -- so its split halves go to LDS as ONE 16-byte store per part
// whatever the layout (round 3's form stored the row-contiguous operands of dgrad / wgrad as 32 / 64 two-byte writes per thread and step:
// the LDS pipe, not the matrix pipe, set those kernels' time).  One LDS buffer, two barriers per step, 20-31 KB per workgroup (four to
// six workgroups per CU); the global loads run two steps ahead in two register sets.
template <bool A_MC, bool B_NC, bool SIX>
__global__ __launch_bounds__(256) void hr_gemm_bf16x3_kernel(const HrGemmArgs a)
{
    constexpr int PARTS = SIX ? 3 : 2;
    __shared__ __attribute__((aligned(16))) __bf16 As[PARTS][HR_GT * HR_GS], Bs[PARTS][HR_GT * HR_GS];
    __shared__ float rs[A_MC ? 4 * HR_GT : 1];           // wgrad: per-thread partial row sums of A, added in a fixed order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * HR_GT, n0 = blockIdx.y * HR_GT;
    const int kb = blockIdx.z * a.k_per_split;
    const int ke = min(a.K, kb + a.k_per_split);
    const bool do_rowsum = (a.rowsum != nullptr) && (blockIdx.y == 0);
    float racc = 0.0f;                                   // this thread's share of its row's sum (A_MC: row tid % 64, k-octet tid / 64)
    hr_acc16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const int a_row = A_MC ? (tid & 63) : (tid >> 2), a_k = A_MC ? 8 * (tid >> 6) : 8 * (tid & 3);
    const int b_row = B_NC ? (tid & 63) : (tid >> 2), b_k = B_NC ? 8 * (tid >> 6) : 8 * (tid & 3);
    const bool vecA = !A_MC && ((a.sam & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0);
    const bool vecM = !A_MC && a.mask && ((a.smm & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.mask) & 15) == 0);
    const bool vecB = !B_NC && ((a.sbn & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.B) & 15) == 0);
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;

    float av0[8], bv0[8], av1[8], bv1[8];               // two slices in flight (steps s + 1 and s + 2 while step s is in LDS)
    auto fetch = [&](float (&av)[8], float (&bv)[8], int k0) {
        {   // A (and its LeakyReLU mask)
            const int gm = m0 + a_row, gk = k0 + a_k;
            const int valid = (gm < a.M) ? max(0, min(8, ke - gk)) : 0;
            if constexpr (A_MC) {
                const int64_t off = (int64_t)gk * a.sak + gm;
                hr_gemm_load8s(a.A + (valid > 0 ? off : 0), a.sak, valid, av);
                if (a.mask) {
                    float mv[8];
                    hr_gemm_load8s(a.mask + (valid > 0 ? (int64_t)gk * a.smk + gm : 0), a.smk, valid, mv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) av[e] = (mv[e] > 0.0f) ? av[e] : av[e] * a.slope;
                }
            } else {
                const int64_t off = (int64_t)gm * a.sam + gk;
                hr_gemm_load8(a.A + (valid > 0 ? off : 0), valid, vecA && ((off & 3) == 0), av);
                if (a.mask) {
                    float mv[8];
                    const int64_t moff = (int64_t)gm * a.smm + gk;
                    hr_gemm_load8(a.mask + (valid > 0 ? moff : 0), valid, vecM && ((moff & 3) == 0), mv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) av[e] = (mv[e] > 0.0f) ? av[e] : av[e] * a.slope;
                }
            }
        }
        {
            const int gn = n0 + b_row, gk = k0 + b_k;
            const int valid = (gn < a.N) ? max(0, min(8, ke - gk)) : 0;
            if constexpr (B_NC) {
                hr_gemm_load8s(a.B + (valid > 0 ? (int64_t)gk * a.sbk + gn : 0), a.sbk, valid, bv);
            } else {
                const int64_t off = (int64_t)gn * a.sbn + gk;
                hr_gemm_load8(a.B + (valid > 0 ? off : 0), valid, vecB && ((off & 3) == 0), bv);
            }
        }
    };
    auto split_store = [&](const float (&av)[8], const float (&bv)[8]) {
        hr_bf8 ap[PARTS], bp[PARTS];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const __bf16 ah = (__bf16)av[e], bh = (__bf16)bv[e];
            ap[0][e] = ah;
            bp[0][e] = bh;
            const float ar = av[e] - (float)ah, br = bv[e] - (float)bh;      // exact remainders
            if constexpr (SIX) {
                const __bf16 amid = (__bf16)ar, bmid = (__bf16)br;
                ap[1][e] = amid;
                bp[1][e] = bmid;
                ap[2][e] = (__bf16)(ar - (float)amid);
                bp[2][e] = (__bf16)(br - (float)bmid);
            } else {
                ap[1][e] = (__bf16)ar;
                bp[1][e] = (__bf16)br;
            }
        }
#pragma unroll
        for (int q = 0; q < PARTS; ++q) {
            *reinterpret_cast<hr_bf8*>(&As[q][a_row * HR_GS + a_k]) = ap[q];
            *reinterpret_cast<hr_bf8*>(&Bs[q][b_row * HR_GS + b_k]) = bp[q];
        }
        if (A_MC && do_rowsum) {                           // db: row sums of the (masked) A slice, in fp32, k ascending
#pragma unroll
            for (int e = 0; e < 8; ++e) racc += av[e];
        }
    };

    auto mfma_step = [&]() {
        // operands: lane l holds 8 consecutive k (k = 8 * (l >> 5) ..) of row / column (l & 31)
#pragma unroll
        for (int kk = 0; kk < HR_GK; kk += 16) {
            const int ro = (lane & 31) * HR_GS + kk + 8 * (lane >> 5);
            const hr_bf8 a_h = *reinterpret_cast<const hr_bf8*>(&As[0][wm * HR_GS + ro]);
            const hr_bf8 b_h = *reinterpret_cast<const hr_bf8*>(&Bs[0][wn * HR_GS + ro]);
            if constexpr (SIX) {                           // smallest terms first
                const hr_bf8 a_m = *reinterpret_cast<const hr_bf8*>(&As[1][wm * HR_GS + ro]);
                const hr_bf8 b_m = *reinterpret_cast<const hr_bf8*>(&Bs[1][wn * HR_GS + ro]);
                const hr_bf8 a_l = *reinterpret_cast<const hr_bf8*>(&As[2][wm * HR_GS + ro]);
                const hr_bf8 b_l = *reinterpret_cast<const hr_bf8*>(&Bs[2][wn * HR_GS + ro]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, b_h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_l, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_m, b_m, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_m, b_h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_m, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_h, acc, 0, 0, 0);
            } else {
                const hr_bf8 a_l = *reinterpret_cast<const hr_bf8*>(&As[1][wm * HR_GS + ro]);
                const hr_bf8 b_l = *reinterpret_cast<const hr_bf8*>(&Bs[1][wn * HR_GS + ro]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, b_h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_l, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_h, acc, 0, 0, 0);
            }
        }
    };

    // Software pipeline, two slices deep: while step s is multiplied out of LDS, the elements of steps s + 1 and s + 2 are on their way in
    // the two register sets -- a slice is requested two whole steps before it is split into LDS (one step ahead left every step waiting
    // out most of a memory latency: 3 us per step, 25 us for a 16 384 x 256 x 256 GEMM)
    if (kb < ke) {
        fetch(av0, bv0, kb);
        split_store(av0, bv0);                                         // step 0 -> LDS
        if (kb + HR_GK < ke) fetch(av1, bv1, kb + HR_GK);              // step 1 -> set 1
        if (kb + 2 * HR_GK < ke) fetch(av0, bv0, kb + 2 * HR_GK);      // step 2 -> set 0
    }
    for (int k0 = kb; k0 < ke; k0 += 2 * HR_GK) {
        __syncthreads();                                   // the slice of step s is complete
        mfma_step();
        if (k0 + HR_GK >= ke) break;
        __syncthreads();                                   // every wavefront has read it
        split_store(av1, bv1);                             // step s + 1 -> LDS
        if (k0 + 3 * HR_GK < ke) fetch(av1, bv1, k0 + 3 * HR_GK);
        __syncthreads();
        mfma_step();
        if (k0 + 2 * HR_GK >= ke) break;
        __syncthreads();
        split_store(av0, bv0);                             // step s + 2 -> LDS
        if (k0 + 4 * HR_GK < ke) fetch(av0, bv0, k0 + 4 * HR_GK);
    }
    // accumulator layout of the 32x32 tile: register r of lane l is D[8 * (r >> 2) + 4 * (l >> 5) + (r & 3)][l & 31]
    float* Cz = a.C + (int64_t)blockIdx.z * a.M * a.ldc;
    const int gn = n0 + wn + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        if (gm < a.M && gn < a.N) {
            float v = acc[r];
            if (a.bias) v += a.bias[gn];
            if (a.act) v = (v > 0.0f) ? v : v * a.slope;
            Cz[(int64_t)gm * a.ldc + gn] = v;
        }
    }
    if (A_MC && do_rowsum) {                               // four k-octet threads per row
        rs[(tid >> 6) * HR_GT + (tid & 63)] = racc;
        __syncthreads();
        if (tid < HR_GT && m0 + tid < a.M)
            a.rowsum[(int64_t)blockIdx.z * a.M + m0 + tid] = (rs[tid] + rs[HR_GT + tid]) + (rs[2 * HR_GT + tid] + rs[3 * HR_GT + tid]);
    }
}

// out[i] = sum_z part[z * n + i], z ascending -- for two arrays at once (dW's partial tiles and db's partial row sums)
__global__ void hr_sum_partials_kernel(const float* part0, int64_t n0, float* out0, const float* part1, int64_t n1, float* out1, int splits)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float* part = part0;
    float* out = out0;
    int64_t n = n0;
    if (i >= n0) { i -= n0; part = part1; out = out1; n = n1; }
    if (i >= n) return;
    float s = 0.0f;
    int z = 0;
    for (; z + 8 <= splits; z += 8) {                      // eight loads in flight, added in ascending z
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = part[(int64_t)(z + j) * n + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; z < splits; ++z) s += part[(int64_t)z * n + i];
    out[i] = s;
}

// wgrad: K = the batch.  256 rows per workgroup (8 steps): 16 384 rows x a 256 x 256 layer = 1024 workgroups, four per CU
static int hr_wgrad_splits(int64_t rows) { const int64_t s = (rows + 255) / 256; return (int)(s < 1 ? 1 : (s > 64 ? 64 : s)); }

size_t hr_linear_workspace_bytes(int64_t rows, int in, int out)
{
    return sizeof(float) * (size_t)hr_wgrad_splits(rows) * ((size_t)out * in + out);
}

void hr_launch_linear_forward(const float* x, int64_t ldx, int64_t rows, int in, const float* w, const float* b, int out, float slope,
                              float* y, int64_t ldy, hipStream_t stream)
{
    if (rows <= 0 || out <= 0) return;
    HrGemmArgs a = {};
    a.A = x; a.sam = ldx; a.sak = 1;
    a.B = w; a.sbk = 1; a.sbn = in;                       // B(k, n) = W[n][k]
    a.slope = slope < 0.0f ? 0.0f : slope;
    a.C = y; a.ldc = ldy;
    a.bias = b; a.act = slope >= 0.0f ? 1 : 0;
    a.M = (int)rows; a.N = out; a.K = in;
    a.k_per_split = (in + HR_GK - 1) & ~(HR_GK - 1);
    dim3 grid((unsigned)((rows + HR_GT - 1) / HR_GT), (unsigned)((out + HR_GT - 1) / HR_GT), 1);
    hipLaunchKernelGGL((hr_gemm_bf16x3_kernel<false, false, true>), grid, dim3(256), 0, stream, a);
}

void hr_launch_linear_backward(const float* x, int64_t ldx, const float* w, const float* y, int64_t ldy, const float* dy, int64_t ld_dy,
                               int64_t rows, int in, int out, float slope, float* dx, int64_t ld_dx, float* dw, float* db, float* workspace,
                               hipStream_t stream)
{
    if (rows <= 0) return;
    const bool masked = (y != nullptr) && slope >= 0.0f;
    if (dx) {                                             // dx (rows, in) = dy' (rows, out) W (out, in)
        HrGemmArgs a = {};
        a.A = dy; a.sam = ld_dy; a.sak = 1;
        a.mask = masked ? y : nullptr; a.smm = ldy; a.smk = 1;
        a.slope = slope;
        a.B = w; a.sbk = in; a.sbn = 1;
        a.C = dx; a.ldc = ld_dx;
        a.M = (int)rows; a.N = in; a.K = out;
        a.k_per_split = (out + HR_GK - 1) & ~(HR_GK - 1);
        dim3 grid((unsigned)((rows + HR_GT - 1) / HR_GT), (unsigned)((in + HR_GT - 1) / HR_GT), 1);
        hipLaunchKernelGGL((hr_gemm_bf16x3_kernel<false, true, false>), grid, dim3(256), 0, stream, a);
    }
    {                                                     // dW (out, in) = dy'^T (out, rows) x (rows, in); db = row sums of dy'^T
        const int splits = hr_wgrad_splits(rows);
        HrGemmArgs a = {};
        a.A = dy; a.sam = 1; a.sak = ld_dy;               // A(m, k) = dy[k][m]
        a.mask = masked ? y : nullptr; a.smm = 1; a.smk = ldy;
        a.slope = slope;
        a.B = x; a.sbk = ldx; a.sbn = 1;
        a.C = workspace; a.ldc = in;
        a.rowsum = workspace + (size_t)splits * out * in;
        a.M = out; a.N = in; a.K = (int)rows;
        a.k_per_split = (int)((((rows + splits - 1) / splits) + HR_GK - 1) & ~(int64_t)(HR_GK - 1));
        dim3 grid((unsigned)((out + HR_GT - 1) / HR_GT), (unsigned)((in + HR_GT - 1) / HR_GT), (unsigned)splits);
        hipLaunchKernelGGL((hr_gemm_bf16x3_kernel<true, true, false>), grid, dim3(256), 0, stream, a);
        const int64_t nw = (int64_t)out * in;
        hipLaunchKernelGGL(hr_sum_partials_kernel, dim3((unsigned)((nw + out + 255) / 256)), dim3(256), 0, stream, workspace, nw, dw, a.rowsum,
                           (int64_t)out, db, splits);
    }
}
