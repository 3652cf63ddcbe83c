// The frame kernel's failing neighbourhood, rebuilt piece by piece (profiles/r05_frame_kernel_difference_bisect.txt): does the
// packed-fp32 sequence the compiler had formed for o.o differ beside MFMA wavefronts when everything else that surrounds it in the sample role
// is there too?  tools/coissue_pk_ubench.hip (the instructions alone, 2.6e10 lane-steps) says no.  Here:
//   matrix wavefronts (0-3): weights through global loads (1 MB, L2 resident) + activations through ds_read_b128 + v_mfma_f32_32x32x16_f16,
//                            s_setprio toggled around the MFMA group like the MLP role;
//   vector wavefronts (4-11), per step: an s_load_dword in flight, two v_mov_b32 writing the source pair, v_pk_mul_f32 op_sel_hi:[1,0] x2,
//                            v_pk_mul_f32 op_sel_hi:[0,1] squares, v_pk_add_f32 -- then the sphere quadratic (IEEE sqrt and divisions, selects)
//                            on the sums, a DPP row shift of the result and an LDS read; MASK bits switch the ingredients:
//       1 scalar load in flight   2 the quadratic behind it   4 DPP + LDS read   8 half of the lanes disabled at a data-dependent time
// Reference = the same kernel's first run with the matrix wavefronts idle (and the host's arithmetic for the products).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/coissue_mix_ubench.hip -o tools/_bin/coissue_mix_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__host__ __device__ inline unsigned lcg(unsigned s) { return s * 1664525u + 1013904223u; }
__host__ __device__ inline unsigned seed_of(int cu, int t) { return (unsigned)(cu * 512 + t) * 2654435761u + 12345u; }
__host__ __device__ inline float unit(unsigned s) { return 0.5f + (float)(s >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ float quad_root(float oo, float dd, float od, float radius)
{
    float a = dd, b = 2.0f * od, cc = oo - radius * radius;
    float disc = b * b - 4.0f * a * cc;
    disc = (disc < 0.0f) ? 0.0f : disc;
    float sq = sqrtf(disc + 1e-8f);
    float t1 = (-b + sq) / (2.0f * a);
    float t2 = (-b - sq) / (2.0f * a);
    t1 = (disc <= 0.0f) ? 0.0f : t1;
    t2 = (disc <= 0.0f) ? 0.0f : t2;
    return ((t2 < 0.0f) || (radius < 0.0f)) ? t1 : t2;
}

template <int MASK>
__global__ __launch_bounds__(768) void k(unsigned* out, const half8* __restrict__ weights, const unsigned* __restrict__ konst, int iters, int mfma_on)
{
    __shared__ half8 s_act[512];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < 512) for (int i = 0; i < 8; ++i) s_act[tid][i] = (_Float16)(0.001f * ((tid * 8 + i) % 977));
    __syncthreads();
    if (wave < 4) {
        if (!mfma_on) return;
        floatx16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
        for (int it = 0; it < iters * 2; ++it) {
            const half8 w0 = weights[((it * 4 + wave) * 64 + lane) & 65535], w1 = weights[((it * 4 + wave) * 64 + 32768 + lane) & 65535];
            const half8 a0 = s_act[(it * 64 + lane) & 511], a1 = s_act[(it * 64 + 256 + lane) & 511];
            __builtin_amdgcn_s_setprio(3);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a0, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, a1, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a1, acc3, 0, 0, 0);
            __builtin_amdgcn_s_setprio(2);
        }
        if (acc0[0] + acc1[1] + acc2[2] + acc3[3] == 12345.678f) out[0] = 1u;       // keep the loop
        return;
    }
    const int t = tid - 256;
    unsigned s = seed_of(blockIdx.x, t), h = 0u, hp = 0u;
    for (int it = 0; it < iters; ++it) {
        s = lcg(s); const float r0 = unit(s);
        s = lcg(s); const float r1 = unit(s);
        s = lcg(s); const float r2 = unit(s);
        s = lcg(s); const float sx = unit(s);
        const float sy = sx * 1.25f, sz = sx * 0.75f;
        if ((MASK & 8) && ((s >> 9) & 1u)) { h = h * 3u + 1u; continue; }          // a data-dependent half of the lanes sits this step out
        // ox = r0 sx, oy = r1 sx' ... exactly the compiler's sequence in the frame kernel: pair (r0, r1) written by two v_mov_b32, multiplied by
        // (s, ?) with op_sel_hi:[1,0]; then squares with op_sel_hi:[0,1] forms and sums.  64-bit integers carry the register pairs.
        const unsigned long long ss = ((unsigned long long)__builtin_bit_cast(unsigned, sy) << 32) | __builtin_bit_cast(unsigned, sx);
        const unsigned long long zz = ((unsigned long long)__builtin_bit_cast(unsigned, r2) << 32) | __builtin_bit_cast(unsigned, sz);
        unsigned long long p01, p2, q01, q2, sum;
        unsigned kk = 1u;
        if (MASK & 1)
            asm volatile("s_load_dword %3, %7, 0x0\n\tv_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\tv_pk_mul_f32 %0, v[20:21], %6 op_sel_hi:[1,0]\n\t"
                         "v_pk_mul_f32 %1, %8, %8 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_mul_f32 %2, %0, %0"
                         : "=&v"(p01), "=&v"(p2), "=&v"(q01), "=&s"(kk) : "v"(r0), "v"(r1), "v"(ss), "s"(konst), "v"(zz) : "v20", "v21", "memory");
        else
            asm volatile("v_mov_b32 v20, %3\n\tv_mov_b32 v21, %4\n\tv_pk_mul_f32 %0, v[20:21], %5 op_sel_hi:[1,0]\n\t"
                         "v_pk_mul_f32 %1, %6, %6 op_sel:[0,1] op_sel_hi:[1,0]\n\tv_pk_mul_f32 %2, %0, %0"
                         : "=&v"(p01), "=&v"(p2), "=&v"(q01) : "v"(r0), "v"(r1), "v"(ss), "v"(zz) : "v20", "v21");
        // p01 = {r0 sx, r1 sx}; p2 = {sz r2, r2 sz} (both halves the same product); q01 = {ox ox, oy oy}
        asm volatile("v_pk_mul_f32 %0, %2, %2\n\ts_nop 0\n\tv_pk_add_f32 %1, %3, %0" : "=&v"(q2), "=&v"(sum) : "v"(p2), "v"(q01));      // {ox ox + oz oz, oy oy + oz oz}
        const float e0 = __builtin_bit_cast(float, (unsigned)sum), e1 = __builtin_bit_cast(float, (unsigned)(sum >> 32));
        const float oy2 = __builtin_bit_cast(float, (unsigned)(q01 >> 32));
        const float oo = e0 + oy2;                                                 // ox ox + oz oz + oy oy
        hp = (hp ^ __builtin_bit_cast(unsigned, oo)) * 31u;
        hp = (hp ^ __builtin_bit_cast(unsigned, e1)) * 2654435761u;
        float r = oo;
        if (MASK & 2) r = quad_root(oo * 0.2f * (float)kk, 1.0f, -0.45f - 0.01f * e1, 0.8f + 0.05f * r0);
        if (MASK & 4) {
            r += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r), 0x111, 0xf, 0xf, true));      // row_shr:1
            r += ((const volatile float*)s_act)[(t * 5 + it) & 2047];
        }
        h = (h ^ __builtin_bit_cast(unsigned, r)) * 31u;
    }
    out[1 + ((size_t)blockIdx.x * 512 + t) * 2] = h;
    out[2 + ((size_t)blockIdx.x * 512 + t) * 2] = hp;
}

template <int MASK>
static int run(const char* name, int cus, int iters, int rounds, const half8* w, const unsigned* konst)
{
    const size_t lanes = (size_t)cus * 512, n = 1 + 2 * lanes;
    unsigned* d = nullptr;
    (void)hipMalloc((void**)&d, n * sizeof(unsigned));
    std::vector<unsigned> want(n), got(n);
    long bad[2] = {0, 0}, badp[2] = {0, 0}, top[2] = {0, 0}, bad_rounds[2] = {0, 0};
    for (int on = 0; on < 2; ++on)
        for (int r = 0; r < rounds; ++r) {
            (void)hipMemset(d, 0, n * sizeof(unsigned));
            hipLaunchKernelGGL(k<MASK>, dim3(cus), dim3(768), 0, 0, d, w, konst, iters, on);
            (void)hipMemcpy(got.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
            if (on == 0 && r == 0) want = got;
            long b = 0;
            for (size_t i = 0; i < lanes; ++i) {
                const bool x = got[1 + 2 * i] != want[1 + 2 * i], y = got[2 + 2 * i] != want[2 + 2 * i];
                if (x || y) { ++b; if ((i & 63) >= 32) ++top[on]; }
                badp[on] += y;
            }
            bad[on] += b; bad_rounds[on] += b ? 1 : 0;
        }
    printf("%-30s %d rounds x %zu lanes x %d steps: beside matrix wavefronts %ld lanes differ in %ld rounds (%ld in lanes 32-63; %ld already in the products' checksum); "
           "matrix wavefronts idle: %ld in %ld rounds\n", name, rounds, lanes, iters, bad[1], bad_rounds[1], top[1], badp[1], bad[0], bad_rounds[0]);
    fflush(stdout);
    (void)hipFree(d);
    return bad[1] != 0;
}

int main(int argc, char** argv)
{
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int iters = argc > 1 ? atoi(argv[1]) : 10000, rounds = argc > 2 ? atoi(argv[2]) : 20;
    half8* w = nullptr;
    unsigned* konst = nullptr;
    std::vector<_Float16> hw(65536 * 8);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (_Float16)(0.002f * (float)((i * 37) % 911) - 0.9f);
    (void)hipMalloc((void**)&w, hw.size() * 2);
    (void)hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    const unsigned one = 1u;
    (void)hipMalloc((void**)&konst, 256);
    (void)hipMemcpy(konst, &one, 4, hipMemcpyHostToDevice);
    printf("%d CUs, one 12-wavefront workgroup per CU (4 matrix wavefronts: global loads + LDS + MFMA + s_setprio; 8 vector wavefronts)\n", cus);
    run<0>("packed sequence alone", cus, iters, rounds, w, konst);
    run<1>("+ scalar load in flight", cus, iters, rounds, w, konst);
    run<3>("+ quadratic", cus, iters, rounds, w, konst);
    run<7>("+ DPP, LDS read", cus, iters, rounds, w, konst);
    run<15>("+ data-dependent EXEC", cus, iters, rounds, w, konst);
    return 0;
}
