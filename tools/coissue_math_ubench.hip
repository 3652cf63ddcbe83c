// Does vector arithmetic compute the same bits beside MFMA wavefronts on the same SIMD?  (the run-to-run single-ray difference of the
// frame kernel -- the head the sample stage reads is identical, the DISTANCE it computes from it is not, and only when matrix wavefronts share
// the SIMD.)  One kernel, 12 wavefronts per workgroup, one workgroup per CU (the frame kernel's mix: one matrix + two vector wavefronts per SIMD): wavefronts 0-3 issue back-to-back v_mfma_f32_32x32x16_f16 (or idle,
// mode 0), wavefronts 4-11 evaluate a chain of the sample stage's own operations -- IEEE division, IEEE square root, expf, reciprocal -- on
// per-lane inputs and fold the bits of every result into a per-lane checksum.  The checksums of a run WITH the matrix wavefronts are compared
// with those of a run without.  OPS selects the chain: 1 division, 2 square root, 4 expf + rcp (sigmoid), 8 the quadratic of the sphere intersect.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/coissue_math_ubench.hip -o tools/_bin/coissue_math_ubench && tools/_bin/coissue_math_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float hr_quad_root(float oo, float dd, float od, float radius)
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

template <int OPS>
__global__ __launch_bounds__(768) void k(unsigned* out, int iters, int mfma_on, float seed)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (wave < 4) {
        if (!mfma_on) return;
        half8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (lane + i)); b[i] = (_Float16)(0.02f * (lane - i)); }
        floatx16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
        for (int it = 0; it < iters * 4; ++it) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc3, 0, 0, 0);
        }
        if (acc0[0] + acc1[1] + acc2[2] + acc3[3] == 12345.678f) out[0] = 1u;       // keep the loop
        return;
    }
    unsigned h = 0u;
    float x = seed + 0.001f * (float)(blockIdx.x * 512 + (tid - 256));
    for (int it = 0; it < iters; ++it) {
        x = x * 1.0001f + 0.37f;
        if (x > 50.0f) x -= 49.0f;
        float r = 0.0f;
        if (OPS & 1) r += (x + 1.3f) / (0.7f + x * 0.11f);
        if (OPS & 2) r += sqrtf(x * 3.7f + 0.2f);
        if (OPS & 4) r += 1.0f / (1.0f + expf(-0.1f * x));
        if (OPS & 8) r += hr_quad_root(0.3f + 0.01f * x, 1.0f, -0.4f - 0.003f * x, 0.9f + 0.002f * x);
        h = h * 31u + __builtin_bit_cast(unsigned, r);
    }
    out[1 + (size_t)blockIdx.x * 512 + (tid - 256)] = h;
}

template <int OPS>
static int run(const char* name, int cus, int iters, int rounds)
{
    const size_t n = 1 + (size_t)cus * 512;
    unsigned* d = nullptr;
    hipMalloc((void**)&d, n * sizeof(unsigned));
    std::vector<unsigned> ref(n), got(n);
    hipMemset(d, 0, n * sizeof(unsigned));
    hipLaunchKernelGGL(k<OPS>, dim3(cus), dim3(768), 0, 0, d, iters, 0, 0.5f);
    hipMemcpy(ref.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
    long bad_lanes = 0, bad_rounds = 0, top = 0;
    for (int r = 0; r < rounds; ++r) {
        hipMemset(d, 0, n * sizeof(unsigned));
        hipLaunchKernelGGL(k<OPS>, dim3(cus), dim3(768), 0, 0, d, iters, 1, 0.5f);
        hipMemcpy(got.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
        long b = 0;
        for (size_t i = 1; i < n; ++i)
            if (got[i] != ref[i]) { ++b; if (((i - 1) & 63) >= 32) ++top; }
        bad_lanes += b; bad_rounds += b ? 1 : 0;
    }
    // and the control: two runs WITHOUT the matrix wavefronts
    hipMemset(d, 0, n * sizeof(unsigned));
    hipLaunchKernelGGL(k<OPS>, dim3(cus), dim3(768), 0, 0, d, iters, 0, 0.5f);
    hipMemcpy(got.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
    long ctl = 0;
    for (size_t i = 1; i < n; ++i) ctl += got[i] != ref[i];
    printf("%-28s %d rounds beside MFMA wavefronts: %ld rounds with a differing lane, %ld lanes in all (%ld of them in lanes 32-63); control without MFMA: %ld lanes differ\n",
           name, rounds, bad_rounds, bad_lanes, top, ctl);
    hipFree(d);
    return bad_lanes != 0;
}

int main(int argc, char** argv)
{
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, rounds = argc > 2 ? atoi(argv[2]) : 20;
    printf("%d CUs, %d chain steps per lane, one 12-wavefront workgroup per CU (4 matrix + 8 vector wavefronts: one + two per SIMD, the frame kernel's mix)\n", cus, iters);
    run<1>("IEEE division", cus, iters, rounds);
    run<2>("IEEE square root", cus, iters, rounds);
    run<4>("expf + reciprocal (sigmoid)", cus, iters, rounds);
    run<8>("sphere quadratic", cus, iters, rounds);
    run<15>("all four", cus, iters, rounds);
    return 0;
}
