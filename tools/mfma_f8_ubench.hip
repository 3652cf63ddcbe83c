// What the f16+fp8 split of the MLP (mlp_split_core.inc, HR_SPLIT_F8LO) relies on, measured on the device:
//   1. v_cvt_pk_fp8_f32 on gfx950: OCP e4m3 bytes, what happens above 448 and below the subnormal range;
//   2. v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 operands: lane l holds row l % 32 and the 32 K values of block l / 32 (byte j <-> k = 32 (l / 32) + j),
//      and the E8M0 scale a lane passes applies to ITS row and ITS block;
//   3. issue cost: cycles per instruction of the fp8 K=64 form against v_mfma_f32_32x32x16_f16 (expected 64 vs 32).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f8_ubench.hip -o tools/_bin/mfma_f8_ubench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ void cvt_kernel(const float* x, int n, unsigned char* o)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.0f, 0, false);
        o[i] = (unsigned char)(r & 0xff);
    }
}

__global__ void mfma_kernel(const v8i* a, const v8i* b, const int* sa, const int* sb, v16f* c)
{
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    c[threadIdx.x] = acc;
}

template <int MODE>
__global__ void rate_kernel(float* out, int iters, long long* cyc)
{
    v16f acc[4] = {};
    v8i a8, b8;
    h8 ah, bh;
    for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838 + threadIdx.x; b8[i] = 0x30303030 + i; ah[i] = (_Float16)(0.5f + i); bh[i] = (_Float16)(0.25f * threadIdx.x); }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MODE == 0 || MODE == 2) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[t], 0, 0, 0);
            }
            if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ah, acc[t], 0, 0, 0);
            if (MODE == 1 || MODE == 2) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 0, 0, 0, 127, 0, 125);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static float e4m3_to_float(unsigned char b)
{
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) v = ldexpf((float)m, -9);
    else v = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main()
{
    // ---- 1. conversions
    const float xs[] = {0.0f, 1.0f, -1.0f, 448.0f, 449.0f, 464.0f, 480.0f, 1000.0f, -1000.0f, 1e9f, INFINITY, NAN, 0.015625f, 0.001953125f, 0.0009765625f,
                        0.00097f, 0.0029f, 1.0625f, 1.1875f, 17.0f, 0.3f};
    const int nx = sizeof(xs) / sizeof(float);
    float* dx; unsigned char* db;
    CK(hipMalloc((void**)&dx, sizeof(xs))); CK(hipMalloc((void**)&db, nx));
    CK(hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, dx, nx, db);
    unsigned char hb[64];
    CK(hipMemcpy(hb, db, nx, hipMemcpyDeviceToHost));
    for (int i = 0; i < nx; ++i) printf("cvt_pk_fp8_f32(%g) = 0x%02x = %g\n", xs[i], hb[i], e4m3_to_float(hb[i]));

    // ---- 2. operand layout and block scales
    srand(7);
    std::vector<unsigned char> A(32 * 64), B(32 * 64);         // [row][k]
    std::vector<int> SA(32 * 2), SB(32 * 2);                  // [row][block]
    auto rnd8 = []() { unsigned char b; do { b = (unsigned char)(rand() & 0xff); } while (((b >> 3) & 15) == 15 || ((b >> 3) & 15) > 10); return b; };
    for (auto& v : A) v = rnd8();
    for (auto& v : B) v = rnd8();
    for (auto& v : SA) v = 120 + rand() % 12;
    for (auto& v : SB) v = 122 + rand() % 12;
    std::vector<v8i> la(64), lb(64);
    std::vector<int> lsa(64), lsb(64);
    for (int l = 0; l < 64; ++l) {
        const int row = l & 31, blk = l >> 5;
        unsigned char ta[32], tb[32];
        for (int j = 0; j < 32; ++j) { ta[j] = A[row * 64 + 32 * blk + j]; tb[j] = B[row * 64 + 32 * blk + j]; }
        memcpy(&la[l], ta, 32); memcpy(&lb[l], tb, 32);
        lsa[l] = SA[row * 2 + blk]; lsb[l] = SB[row * 2 + blk];
    }
    v8i *da, *dbb; int *dsa, *dsb; v16f* dc;
    CK(hipMalloc((void**)&da, 64 * 32)); CK(hipMalloc((void**)&dbb, 64 * 32)); CK(hipMalloc((void**)&dsa, 256)); CK(hipMalloc((void**)&dsb, 256)); CK(hipMalloc((void**)&dc, 64 * 64));
    CK(hipMemcpy(da, la.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(dbb, lb.data(), 64 * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, lsa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, lsb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, da, dbb, dsa, dsb, dc);
    std::vector<float> C(64 * 16);
    CK(hipMemcpy(C.data(), dc, 64 * 64, hipMemcpyDeviceToHost));
    double worst = 0.0, worst_noscale = 0.0, ref_max = 0.0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int n = l & 31, m = (r / 4) * 8 + (l >> 5) * 4 + (r & 3);     // D[m][n]: m indexes the first operand's rows
            double want = 0.0, plain = 0.0;
            for (int k = 0; k < 64; ++k) {
                const double p = (double)e4m3_to_float(A[m * 64 + k]) * (double)e4m3_to_float(B[n * 64 + k]);
                want += p * ldexp(1.0, SA[m * 2 + k / 32] - 127 + SB[n * 2 + k / 32] - 127);
                plain += p;
            }
            worst = fmax(worst, fabs(want - C[l * 16 + r]));
            worst_noscale = fmax(worst_noscale, fabs(plain - C[l * 16 + r]));
            ref_max = fmax(ref_max, fabs(want));
        }
    printf("layout + per-(row, block) scales: max |D - expected| = %.3g (max |expected| %.3g; against the unscaled sum %.3g) -> %s\n", worst, ref_max, worst_noscale,
           worst <= 1e-5 * ref_max ? "AS ASSUMED" : "DIFFERENT");

    // ---- 3. issue cost
    float* dout; long long* dcyc;
    CK(hipMalloc((void**)&dout, 4 * 256 * 1024)); CK(hipMalloc((void**)&dcyc, 8));
    const int iters = 4096;
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) CK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(256), dim3(256), 0, 0, dout, iters, dcyc);
            if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(256), dim3(256), 0, 0, dout, iters, dcyc);
            if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(256), dim3(256), 0, 0, dout, iters, dcyc);
            if (rep == 1) CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
        }
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        long long cyc;
        CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
        const char* what[] = {"3 x f16 32x32x16 per tile (f16x3)", "1 x fp8 32x32x64 (scaled) per tile", "2 x f16 32x32x16 + 1 x fp8 32x32x64 per tile (f16+fp8)"};
        printf("%-58s %.1f counter ticks per tile-step (4 tiles per step, one wavefront per SIMD)\n", what[mode], (double)cyc / iters / 4);
        printf("%-58s %.3f ms for %d steps -> %.1f ns per tile-step\n", "", ms, iters, ms * 1e6 / iters / 4);
    }
    return 0;
}
