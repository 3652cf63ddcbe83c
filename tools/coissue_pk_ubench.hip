// Do the packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32, which the compiler's SLP vectoriser forms out of the sample stage's scalar
// arithmetic) compute the same bits beside MFMA wavefronts on the same SIMD?  (the first intermediate of the frame kernel's sample
// role that differs between two renders of the same rays is o.o -- products and sums the compiler emitted as v_pk_mul_f32 with op_sel_hi
// modifiers and v_pk_add_f32; the activated head value and the radius in front of it are identical.  profiles/r05_frame_kernel_difference_bisect.txt)
// Same shape as tools/coissue_mask_ubench.hip: 4 matrix wavefronts fed from LDS + 8 vector wavefronts per workgroup, one workgroup per CU.
// Vector wavefronts: per lane and step four floats in [0.5, 1.5) from an integer generator;
//   C     two products and two sums as the compiler emits them
//   pk    v_pk_mul_f32 then v_pk_add_f32
//   sel   v_pk_mul_f32 with op_sel_hi:[1,0] (high result = a.hi * b.lo), the form in the frame kernel, then v_pk_add_f32
//   mov   the same, its source pair written by two v_mov_b32 directly in front of it (as in the frame kernel)
//   exec  'mov' in a data-dependent half of the lanes, the scalar form in the others (partial EXEC)
// every product and sum of two floats is exactly defined, so the host computes every lane's checksum.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/coissue_pk_ubench.hip -o tools/_bin/coissue_pk_ubench && tools/_bin/coissue_pk_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));

__host__ __device__ inline unsigned lcg(unsigned s) { return s * 1664525u + 1013904223u; }
__host__ __device__ inline unsigned seed_of(int cu, int t) { return (unsigned)(cu * 512 + t) * 2654435761u + 12345u; }
__host__ __device__ inline float unit(unsigned s) { return 0.5f + (float)(s >> 8) * (1.0f / 16777216.0f); }
__host__ __device__ inline unsigned bits(float x) { unsigned u; memcpy(&u, &x, 4); return u; }

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
        s = lcg(s); const float a0 = unit(s);
        s = lcg(s); const float a1 = unit(s);
        s = lcg(s); const float b0 = unit(s);
        s = lcg(s); const float b1 = unit(s);
        float q0, q1;
        if (HOW == 0) {                                   // as the compiler emits it
            const float p0 = a0 * b0, p1 = a1 * b1;
            q0 = p0 + a0; q1 = p1 + a1;
        } else {                                          // (64-bit integers carry the register pairs: an inline-asm "+v" / "=v" operand of a 2-float vector type comes back with element 0 in both halves on this hipcc)
            const unsigned long long a = ((unsigned long long)__builtin_bit_cast(unsigned, a1) << 32) | __builtin_bit_cast(unsigned, a0);
            const unsigned long long b = ((unsigned long long)__builtin_bit_cast(unsigned, b1) << 32) | __builtin_bit_cast(unsigned, b0);
            unsigned long long p, q;
            const bool scalar_lane = (HOW == 4) && (s & 0x10000u);              // HOW 4: a data-dependent half of the lanes takes the scalar form (partial EXEC around the packed one)
            if (scalar_lane) {
                const float p0 = a0 * b0, p1 = a1 * b0;
                q0 = p0 + a0; q1 = p1 + a1;
            } else {
                if (HOW == 1) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(a), "v"(b));                          // {a0 b0, a1 b1}
                else if (HOW == 2) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(b));     // {a0 b0, a1 b0}: the form in the frame kernel
                else          // ... and its surroundings there: the register pair is written by two v_mov_b32 DIRECTLY in front of the packed instruction
                    asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_pk_mul_f32 %0, v[20:21], %3 op_sel_hi:[1,0]" : "=v"(p) : "v"(a0), "v"(a1), "v"(b) : "v20", "v21");
                asm volatile("s_nop 0\n\tv_pk_add_f32 %0, %1, %2" : "=v"(q) : "v"(p), "v"(a));
                q0 = __builtin_bit_cast(float, (unsigned)q); q1 = __builtin_bit_cast(float, (unsigned)(q >> 32));
            }
        }
        unsigned w0 = __builtin_bit_cast(unsigned, q0), w1 = __builtin_bit_cast(unsigned, q1);
        h = (h ^ w0) * 31u;               // (not h * 31 + w: this hipcc selects v_mad_u64_u32 for it and hands the SAME 64-bit register pair to both steps -- q[1] is never added)
        h = (h ^ w1) * 2654435761u;
        if ((it & 7) == 0) h ^= __builtin_bit_cast(unsigned, ((const volatile float*)s_act)[(t * 5 + it) & 2047]) & 0u;
    }
    out[1 + (size_t)blockIdx.x * 512 + t] = h;
}

template <int HOW>
static int run(const char* name, int cus, int iters, int rounds)
{
    const size_t n = 1 + (size_t)cus * 512;
    unsigned* d = nullptr;
    (void)hipMalloc((void**)&d, n * sizeof(unsigned));
    std::vector<unsigned> want(n), got(n);
    for (int cu = 0; cu < cus; ++cu)
        for (int t = 0; t < 512; ++t) {
            unsigned s = seed_of(cu, t), h = 0u;
            for (int it = 0; it < iters; ++it) {
                s = lcg(s); const float a0 = unit(s);
                s = lcg(s); const float a1 = unit(s);
                s = lcg(s); const float b0 = unit(s);
                s = lcg(s); const float b1 = unit(s);
                volatile float p0 = a0 * b0, p1 = a1 * (HOW >= 2 ? b0 : b1);
                const float q0 = p0 + a0, q1 = p1 + a1;
                h = (h ^ bits(q0)) * 31u; h = (h ^ bits(q1)) * 2654435761u;
            }
            want[1 + (size_t)cu * 512 + t] = h;
        }
    long bad[2] = {0, 0}, top[2] = {0, 0}, bad_rounds[2] = {0, 0}, host_bad = 0;
    for (int on = 0; on < 2; ++on)
        for (int r = 0; r < rounds; ++r) {
            (void)hipMemset(d, 0, n * sizeof(unsigned));
            hipLaunchKernelGGL(k<HOW>, dim3(cus), dim3(768), 0, 0, d, iters, on);
            (void)hipMemcpy(got.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
            long b = 0;
            if (on == 0 && r == 0) {
                for (size_t i = 1; i < n; ++i) host_bad += got[i] != want[i];
                want = got;                       // the reference from here on: the first run with the matrix wavefronts idle
            }
            for (size_t i = 1; i < n; ++i)
                if (got[i] != want[i]) { ++b; if (((i - 1) & 63) >= 32) ++top[on]; }
            bad[on] += b; bad_rounds[on] += b ? 1 : 0;
        }
    printf("%-4s %d rounds x %zu lanes x %d steps: beside MFMA wavefronts %ld lanes wrong in %ld rounds (%ld of them lanes 32-63); matrix wavefronts idle: %ld lanes wrong in %ld rounds (%ld in lanes 32-63); first idle run against the host's arithmetic: %ld lanes differ\n",
           name, rounds, n - 1, iters, bad[1], bad_rounds[1], top[1], bad[0], bad_rounds[0], top[0], host_bad);
    (void)hipFree(d);
    return bad[1] != 0;
}

int main(int argc, char** argv)
{
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int iters = argc > 1 ? atoi(argv[1]) : 10000, rounds = argc > 2 ? atoi(argv[2]) : 20;
    printf("%d CUs, one 12-wavefront workgroup per CU (4 matrix wavefronts fed from LDS + 8 vector wavefronts)\n", cus);
    run<0>("C", cus, iters, rounds);
    run<1>("pk", cus, iters, rounds);
    run<2>("sel", cus, iters, rounds);
    run<3>("mov", cus, iters, rounds);
    run<4>("exec", cus, iters, rounds);
    return 0;
}
