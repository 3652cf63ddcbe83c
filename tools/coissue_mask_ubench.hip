// Does a lane mask written by a vector compare reach the vector select that reads it, beside MFMA wavefronts on the same SIMD?  (the
// frame kernel's run-to-run single-ray difference sits in lanes 32-63 of a sample wavefront, in arithmetic whose inputs are identical, and it
// reacts to wait states after v_cmp / before v_cndmask -- profiles/r05_frame_kernel_difference_bisect.txt.  tools/coissue_math_ubench.hip cannot
// see a stale mask: its compares come out the same way in every lane and every step.)
// One kernel, 12 wavefronts per workgroup, one workgroup per CU (one matrix + two vector wavefronts per SIMD).  Wavefronts 0-3: back-to-back
// v_mfma_f32_32x32x16_f16 fed from LDS (ds_read_b128 per pair of MFMAs, the MLP role's shape), or idle.  Wavefronts 4-11: per lane and step two
// 24-bit pseudo-random numbers u, v and two words a, b;  r = (u < v) ? a : b  folded into a checksum.  All integer-exact, so the host computes
// the expected checksum of every lane.  The select is written three ways:
//   C      whatever the compiler emits (v_cmp into VCC or an SGPR pair, its own wait states, v_cndmask);
//   asm1   v_cmp_lt_f32 s[n:n+1] ; s_nop 1 ; v_cndmask  -- the two wait states LLVM's hazard recognizer gives this pair on gfx940/950;
//   asm0   the same with s_nop 0 -- one wait state LESS than the rule: shows whether this test can see the hazard at all.
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_mask_ubench.hip -o tools/_bin/coissue_mask_ubench && tools/_bin/coissue_mask_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__host__ __device__ inline unsigned lcg(unsigned s) { return s * 1664525u + 1013904223u; }
__host__ __device__ inline unsigned seed_of(int cu, int t) { return (unsigned)(cu * 512 + t) * 2654435761u + 12345u; }

template <int HOW>
__global__ __launch_bounds__(768) void k(unsigned* out, int iters, int mfma_on)
{
    __shared__ half8 s_act[512];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < 512) for (int i = 0; i < 8; ++i) s_act[tid][i] = (_Float16)(0.001f * ((tid * 8 + i) % 977));
    __syncthreads();
    if (wave < 4) {
        if (!mfma_on) return;
        half8 b;
        for (int i = 0; i < 8; ++i) b[i] = (_Float16)(0.02f * (lane - i));
        floatx16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
        for (int it = 0; it < iters * 2; ++it) {
            const half8 a0 = s_act[(it * 64 + lane) & 511], a1 = s_act[(it * 64 + 256 + lane) & 511];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b, acc3, 0, 0, 0);
        }
        if (acc0[0] + acc1[1] + acc2[2] + acc3[3] == 12345.678f) out[0] = 1u;       // keep the loop
        return;
    }
    const int t = tid - 256;
    unsigned s = seed_of(blockIdx.x, t), h = 0u;
    for (int it = 0; it < iters; ++it) {
        s = lcg(s); const float u = (float)(s >> 8);
        s = lcg(s); const float v = (float)(s >> 8);
        s = lcg(s); const unsigned a = s;
        s = lcg(s); const unsigned b = s;
        unsigned r;
        if (HOW == 0) {
            r = (u < v) ? a : b;
        } else {
            unsigned long long m;
            if (HOW == 1) asm volatile("v_cmp_lt_f32 %1, %2, %3\n\ts_nop 1\n\tv_cndmask_b32 %0, %5, %4, %1" : "=v"(r), "=&s"(m) : "v"(u), "v"(v), "v"(a), "v"(b));
            else          asm volatile("v_cmp_lt_f32 %1, %2, %3\n\ts_nop 0\n\tv_cndmask_b32 %0, %5, %4, %1" : "=v"(r), "=&s"(m) : "v"(u), "v"(v), "v"(a), "v"(b));
        }
        h = h * 31u + r;
        if ((it & 7) == 0) h ^= __builtin_bit_cast(unsigned, ((const volatile float*)s_act)[(t * 5 + it) & 2047]) & 0u;      // LDS traffic in the vector role as well (the value is masked out)
    }
    out[1 + (size_t)blockIdx.x * 512 + t] = h;
}

template <int HOW>
static int run(const char* name, int cus, int iters, int rounds)
{
    const size_t n = 1 + (size_t)cus * 512;
    unsigned* d = nullptr;
    hipMalloc((void**)&d, n * sizeof(unsigned));
    static std::vector<unsigned> want;
    std::vector<unsigned> got(n);
    const bool have = want.size() == n;
    want.resize(n);
    for (int cu = 0; cu < (have ? 0 : cus); ++cu)
        for (int t = 0; t < 512; ++t) {
            unsigned s = seed_of(cu, t), h = 0u;
            for (int it = 0; it < iters; ++it) {
                s = lcg(s); const unsigned u = s >> 8;
                s = lcg(s); const unsigned v = s >> 8;
                s = lcg(s); const unsigned a = s;
                s = lcg(s); const unsigned b = s;
                h = h * 31u + ((u < v) ? a : b);
            }
            want[1 + (size_t)cu * 512 + t] = h;
        }
    long bad[2] = {0, 0}, top[2] = {0, 0}, bad_rounds[2] = {0, 0};
    for (int on = 0; on < 2; ++on)
        for (int r = 0; r < rounds; ++r) {
            hipMemset(d, 0, n * sizeof(unsigned));
            hipLaunchKernelGGL(k<HOW>, dim3(cus), dim3(768), 0, 0, d, iters, on);
            hipMemcpy(got.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
            long b = 0;
            for (size_t i = 1; i < n; ++i)
                if (got[i] != want[i]) { ++b; if (((i - 1) & 63) >= 32) ++top[on]; }
            bad[on] += b; bad_rounds[on] += b ? 1 : 0;
        }
    printf("%-6s %d rounds x %zu lanes x %d selects: beside MFMA wavefronts %ld lanes wrong in %ld rounds (%ld of them lanes 32-63); matrix wavefronts idle: %ld lanes wrong in %ld rounds (%ld in lanes 32-63)\n",
           name, rounds, n - 1, iters, bad[1], bad_rounds[1], top[1], bad[0], bad_rounds[0], top[0]);
    hipFree(d);
    return bad[1] != 0;
}

int main(int argc, char** argv)
{
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, rounds = argc > 2 ? atoi(argv[2]) : 20;
    printf("%d CUs, one 12-wavefront workgroup per CU (4 matrix wavefronts fed from LDS + 8 vector wavefronts)\n", cus);
    run<0>("C", cus, iters, rounds);
    run<1>("asm1", cus, iters, rounds);
    run<2>("asm0", cus, iters, rounds);
    return 0;
}
