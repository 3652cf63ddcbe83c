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
    int k_per_split;                         // multiple of 16; gridDim.z splits
};

constexpr int HR_GT = 64;                    // tile edge
constexpr int HR_GS = 24;                    // bf16 elements per LDS row (16 + 8: rows of a ds_read_b128 group on distinct banks)

// A_MC: A is contiguous along m (else along k).  B_NC: B is contiguous along n (else along k).
// SIX: the six-product form (operands split three ways).
template <bool A_MC, bool B_NC, bool SIX>
__global__ __launch_bounds__(256) void hr_gemm_bf16x3_kernel(const HrGemmArgs a)
{
    __shared__ __attribute__((aligned(16))) __bf16 Ah[HR_GT * HR_GS], Al[HR_GT * HR_GS], Bh[HR_GT * HR_GS], Bl[HR_GT * HR_GS];
    __shared__ __attribute__((aligned(16))) __bf16 Am[SIX ? HR_GT * HR_GS : 8], Bm[SIX ? HR_GT * HR_GS : 8];     // middle parts
    __shared__ float rs[16 * HR_GT];                      // wgrad: per-k-thread partial row sums of A, added in a fixed order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * HR_GT, n0 = blockIdx.y * HR_GT;
    const int kb = blockIdx.z * a.k_per_split;
    const int ke = min(a.K, kb + a.k_per_split);
    const bool do_rowsum = (a.rowsum != nullptr) && (blockIdx.y == 0);
    float racc[4] = {0.0f, 0.0f, 0.0f, 0.0f};            // this thread's share of the row sums (its 4 rows are fixed: A_MC)
    hr_acc16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    // staging: 256 threads x 4 elements = a 64x16 slice.  The 4 elements of a thread run along the contiguous dimension.
    //   A_MC: thread -> k = tid / 16, m = 4 * (tid % 16) + e       else: m = tid / 4, k = 4 * (tid % 4) + e
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    for (int k0 = kb; k0 < ke; k0 += 16) {
        float av[4], bv[4];
        int am[4], ak[4], bn[4], bk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (A_MC) { ak[e] = tid >> 4; am[e] = 4 * (tid & 15) + e; } else { am[e] = tid >> 2; ak[e] = 4 * (tid & 3) + e; }
            if (B_NC) { bk[e] = tid >> 4; bn[e] = 4 * (tid & 15) + e; } else { bn[e] = tid >> 2; bk[e] = 4 * (tid & 3) + e; }
            const int gm = m0 + am[e], gka = k0 + ak[e];
            float v = 0.0f;
            if (gm < a.M && gka < ke) {
                v = a.A[(int64_t)gm * a.sam + (int64_t)gka * a.sak];
                if (a.mask) v = (a.mask[(int64_t)gm * a.smm + (int64_t)gka * a.smk] > 0.0f) ? v : v * a.slope;
            }
            av[e] = v;
            const int gn = n0 + bn[e], gkb = k0 + bk[e];
            bv[e] = (gn < a.N && gkb < ke) ? a.B[(int64_t)gkb * a.sbk + (int64_t)gn * a.sbn] : 0.0f;
        }
        __syncthreads();                                   // the previous step's MFMAs have read the LDS tiles
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 ah = (__bf16)av[e], bh = (__bf16)bv[e];
            Ah[am[e] * HR_GS + ak[e]] = ah;
            Bh[bn[e] * HR_GS + bk[e]] = bh;
            const float ar = av[e] - (float)ah, br = bv[e] - (float)bh;      // exact remainders
            if constexpr (SIX) {
                const __bf16 amid = (__bf16)ar, bmid = (__bf16)br;
                Am[am[e] * HR_GS + ak[e]] = amid;
                Bm[bn[e] * HR_GS + bk[e]] = bmid;
                Al[am[e] * HR_GS + ak[e]] = (__bf16)(ar - (float)amid);
                Bl[bn[e] * HR_GS + bk[e]] = (__bf16)(br - (float)bmid);
            } else {
                Al[am[e] * HR_GS + ak[e]] = (__bf16)ar;
                Bl[bn[e] * HR_GS + bk[e]] = (__bf16)br;
            }
        }
        if (A_MC && do_rowsum) {                           // db: row sums of the (masked) A slice, in fp32
#pragma unroll
            for (int e = 0; e < 4; ++e) racc[e] += av[e];
        }
        __syncthreads();
        // operands: lane l holds 8 consecutive k (k = 8 * (l >> 5) ..) of row / column (l & 31)
        const int ro = (lane & 31) * HR_GS + 8 * (lane >> 5);
        const hr_bf8 a_h = *reinterpret_cast<const hr_bf8*>(Ah + wm * HR_GS + ro);
        const hr_bf8 a_l = *reinterpret_cast<const hr_bf8*>(Al + wm * HR_GS + ro);
        const hr_bf8 b_h = *reinterpret_cast<const hr_bf8*>(Bh + wn * HR_GS + ro);
        const hr_bf8 b_l = *reinterpret_cast<const hr_bf8*>(Bl + wn * HR_GS + ro);
        if constexpr (SIX) {                               // smallest terms first
            const hr_bf8 a_m = *reinterpret_cast<const hr_bf8*>(Am + wm * HR_GS + ro);
            const hr_bf8 b_m = *reinterpret_cast<const hr_bf8*>(Bm + wn * HR_GS + ro);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, b_h, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_l, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_m, b_m, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_m, b_h, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_m, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_h, acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, b_h, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_l, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_h, acc, 0, 0, 0);
        }
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
    if (A_MC && do_rowsum) {
#pragma unroll
        for (int e = 0; e < 4; ++e) rs[(tid >> 4) * HR_GT + 4 * (tid & 15) + e] = racc[e];
        __syncthreads();
        if (tid < HR_GT && m0 + tid < a.M) {
            float t = 0.0f;
            for (int i = 0; i < 16; ++i) t += rs[i * HR_GT + tid];
            a.rowsum[(int64_t)blockIdx.z * a.M + m0 + tid] = t;
        }
    }
}

// out[i] = sum_z part[z * n + i], z ascending
__global__ void hr_sum_partials_kernel(const float* part, int64_t n, int splits, float* out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (int z = 0; z < splits; ++z) s += part[(int64_t)z * n + i];
    out[i] = s;
}

static int hr_wgrad_splits(int64_t rows) { const int64_t s = (rows + 511) / 512; return (int)(s < 1 ? 1 : (s > 64 ? 64 : s)); }

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
    a.k_per_split = (in + 15) & ~15;
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
        a.k_per_split = (out + 15) & ~15;
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
        a.k_per_split = (int)((((rows + splits - 1) / splits) + 15) & ~(int64_t)15);
        dim3 grid((unsigned)((out + HR_GT - 1) / HR_GT), (unsigned)((in + HR_GT - 1) / HR_GT), (unsigned)splits);
        hipLaunchKernelGGL((hr_gemm_bf16x3_kernel<true, true, false>), grid, dim3(256), 0, stream, a);
        const int64_t nw = (int64_t)out * in;
        hipLaunchKernelGGL(hr_sum_partials_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, stream, workspace, nw, splits, dw);
        hipLaunchKernelGGL(hr_sum_partials_kernel, dim3((unsigned)((out + 255) / 256)), dim3(256), 0, stream, a.rowsum, (int64_t)out, splits, db);
    }
}
