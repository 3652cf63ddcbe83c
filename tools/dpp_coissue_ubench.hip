// Does a cross-lane (DPP) instruction that reads a register a vector instruction has just written still see all 64 lanes of it when a
// matrix wavefront of ANOTHER kernel shares the SIMD?  (frames of the sample kernel differ in ONE ray -- always lanes 48-63 of
// a wavefront, the fourth pass of a wave64 instruction -- only while an f16 MLP kernel of another launch is co-resident.)
// Kernel V: every wavefront runs a long deterministic chain  x = fma(x, a, b);  y = dpp(x);  x = x + c * y  with the compiler's own
// hazard padding, for several DPP controls and a packed-fp32 producer; its final values are compared with those of a run ALONE.
// Kernel M: back-to-back v_mfma_f32_32x32x16_{f16|bf16}, one workgroup per CU, ~200 VGPRs, on another stream.
//   hipcc --offload-arch=gfx950 -O3 tools/dpp_coissue_ubench.hip -o /tmp/dcu && /tmp/dcu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float dpp(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)); }

__global__ __launch_bounds__(256, 6) void kv(const float* in, float* out, int iters)
{
    float x0 = in[threadIdx.x & 255], x1 = in[(threadIdx.x + 3) & 255], x2 = in[(threadIdx.x + 7) & 255], x3 = in[(threadIdx.x + 11) & 255];
    const float a = 0.999f, b = 0.001f, c = 0.0625f;
    for (int it = 0; it < iters; ++it) {
        x0 = __builtin_fmaf(x0, a, b); x0 = x0 + c * (dpp<0x111>(x0) - x0);          // row_shr:1
        x1 = __builtin_fmaf(x1, a, b); x1 = x1 + c * (dpp<0x138>(x1) - x1);          // wave_shr:1
        x2 = __builtin_fmaf(x2, a, b); x2 = x2 + c * (dpp<0xB1>(x2) - x2);           // quad_perm [1,0,3,2]
        f2 p = {x3, x2};
        p = __builtin_elementwise_fma(p, f2{a, a}, f2{b, b});                        // v_pk_fma_f32 producer
        x3 = p.x + c * (dpp<0x128>(p.x) - p.x);                                      // row_ror:8
        x2 = p.y;
    }
    float* o = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    o[0] = x0; o[1] = x1; o[2] = x2; o[3] = x3;
}

template <bool F16>
__global__ __launch_bounds__(256, 2) void km(const float* in, float* out, int iters)
{
    floatx16 acc[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    const int lane = threadIdx.x & 63;
    float keep[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) { keep[i] = in[(lane + i) & 255]; asm volatile("" : "+v"(keep[i])); }
    __builtin_amdgcn_s_setprio(2);
    for (int it = 0; it < iters; ++it) {
        if constexpr (F16) {
            halfx8 av, bv;
            for (int i = 0; i < 8; ++i) { av[i] = (_Float16)in[(lane * 8 + i + it) & 255]; bv[i] = (_Float16)in[(lane * 8 + i + 64 + it) & 255]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[j], 0, 0, 0);
        } else {
            bf16x8 av, bv;
            for (int i = 0; i < 8; ++i) { av[i] = (__bf16)in[(lane * 8 + i + it) & 255]; bv[i] = (__bf16)in[(lane * 8 + i + 64 + it) & 255]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][9];
#pragma unroll
    for (int i = 0; i < 96; ++i) { asm volatile("" : "+v"(keep[i])); s += keep[i]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    const int NB = 256 * 24, ITERS = 1500;
    float *in, *outv, *outm;
    hipMalloc(&in, 4096); hipMalloc(&outv, (size_t)NB * 256 * 4 * 4); hipMalloc(&outm, 4 * 256 * 1024);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0.5f + 0.001f * i;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    std::vector<float> ref((size_t)NB * 256 * 4), got(ref.size());
    hipLaunchKernelGGL(kv, dim3(NB), dim3(256), 0, sb, in, outv, ITERS);
    hipDeviceSynchronize();
    hipMemcpy(ref.data(), outv, ref.size() * 4, hipMemcpyDeviceToHost);
    for (int mode = 0; mode < 3; ++mode) {
        long bad_total = 0; int bad_launches = 0; long lane_hist[4] = {0, 0, 0, 0}; long chain_hist[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 30; ++rep) {
            hipMemset(outv, 0, ref.size() * 4);
            hipDeviceSynchronize();
            if (mode == 1) hipLaunchKernelGGL(km<true>, dim3(256), dim3(256), 0, sa, in, outm, 9000);
            if (mode == 2) hipLaunchKernelGGL(km<false>, dim3(256), dim3(256), 0, sa, in, outm, 9000);
            hipLaunchKernelGGL(kv, dim3(NB), dim3(256), 0, sb, in, outv, ITERS);
            hipDeviceSynchronize();
            hipMemcpy(got.data(), outv, got.size() * 4, hipMemcpyDeviceToHost);
            long bad = 0;
            for (size_t i = 0; i < got.size(); ++i)
                if (memcmp(&got[i], &ref[i], 4) != 0) { ++bad; ++lane_hist[((i / 4) % 64) / 16]; ++chain_hist[i % 4]; }
            bad_total += bad; bad_launches += bad ? 1 : 0;
        }
        printf("mode %d (%s): %ld differing values in %d of 30 launches; by lane quarter [0-15, 16-31, 32-47, 48-63]: %ld %ld %ld %ld; by chain [row_shr, wave_shr, quad_perm, pk_fma+row_ror]: %ld %ld %ld %ld\n",
               mode, mode == 0 ? "vector kernel alone" : mode == 1 ? "beside the f16 MFMA kernel" : "beside the bf16 MFMA kernel", bad_total, bad_launches,
               lane_hist[0], lane_hist[1], lane_hist[2], lane_hist[3], chain_hist[0], chain_hist[1], chain_hist[2], chain_hist[3]);
        fflush(stdout);
    }
    return 0;
}
