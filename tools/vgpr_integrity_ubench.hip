// Do the registers of a vector wavefront survive next to matrix wavefronts of ANOTHER kernel on the same SIMD?
// Kernel V (256 threads, ~80 VGPRs, several blocks per CU): every lane fills NR registers with a pattern of (block, thread, register),
// keeps them live through a few thousand dependent-free VALU instructions on OTHER registers, then checks every one of them.
// Kernel M (256 threads, ~200 VGPRs, one block per CU, persistent for ~2 ms): v_mfma_f32_32x32x16_{f16|bf16} back to back with
// operands streamed from memory.  V runs alone, then beside M (two streams); mismatches are counted with their lane and register.
//   hipcc --offload-arch=gfx950 -O3 tools/vgpr_integrity_ubench.hip -o /tmp/vgi && /tmp/vgi
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NR = 48;
__global__ __launch_bounds__(256, 5) void kv(unsigned* bad, unsigned* detail, int iters, const float* in)
{
    unsigned r[NR];
    const unsigned tag = blockIdx.x * 256u + threadIdx.x;
#pragma unroll
    for (int i = 0; i < NR; ++i) { r[i] = tag * 64u + i; asm volatile("" : "+v"(r[i])); }
    float x0 = in[threadIdx.x & 255], x1 = in[(threadIdx.x + 1) & 255], x2 = in[(threadIdx.x + 2) & 255], x3 = in[(threadIdx.x + 3) & 255];
    const float a = in[300], b = in[301];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
            asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x1));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
            asm volatile("v_rcp_f32 %0, %0" : "+v"(x3));
        }
    }
    unsigned nbad = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        asm volatile("" : "+v"(r[i]));
        if (r[i] != tag * 64u + i) {
            ++nbad;
            const unsigned slot = atomicAdd(bad + 1, 1u);
            if (slot < 64) { detail[4 * slot] = tag; detail[4 * slot + 1] = i; detail[4 * slot + 2] = r[i]; detail[4 * slot + 3] = tag * 64u + i; }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
    if (x0 + x1 + x2 + x3 == 12345.678f) bad[2] = 1;
}

template <bool F16>
__global__ __launch_bounds__(256, 2) void km(const float* in, float* out, int iters)
{
    floatx16 acc[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    const int lane = threadIdx.x & 63;
    float keep[96];                                 // pad the allocation to ~200 registers like the MLP kernel's
#pragma unroll
    for (int i = 0; i < 96; ++i) { keep[i] = in[(lane + i) & 255]; asm volatile("" : "+v"(keep[i])); }
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
    float *in, *out; unsigned *bad, *detail;
    hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&bad, 64); hipMalloc(&detail, 64 * 16);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0.5f + 0.001f * i;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    for (int mode = 0; mode < 3; ++mode) {          // 0: V alone, 1: beside the f16 matrix kernel, 2: beside the bf16 one
        unsigned total = 0, events = 0;
        for (int rep = 0; rep < 30; ++rep) {
            hipMemset(bad, 0, 64);
            hipDeviceSynchronize();
            if (mode == 1) hipLaunchKernelGGL(km<true>, dim3(256), dim3(256), 0, sa, in, out, 12000);
            if (mode == 2) hipLaunchKernelGGL(km<false>, dim3(256), dim3(256), 0, sa, in, out, 12000);
            hipLaunchKernelGGL(kv, dim3(256 * 40), dim3(256), 0, sb, bad, detail, 600, in);
            hipDeviceSynchronize();
            unsigned hb[4]; hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
            total += hb[0]; events += hb[1] ? 1 : 0;
            if (hb[1]) {
                unsigned d[16]; hipMemcpy(d, detail, 64, hipMemcpyDeviceToHost);
                for (int i = 0; i < 4 && i < (int)hb[1]; ++i)
                    printf("  mode %d rep %d: thread tag %u (block %u, lane %u) register %u holds %08x, expected %08x\n", mode, rep, d[4 * i], d[4 * i] / 256, d[4 * i] % 64, d[4 * i + 1], d[4 * i + 2], d[4 * i + 3]);
            }
        }
        printf("mode %d (%s): corrupted register values %u in %u of 30 launches\n", mode, mode == 0 ? "vector kernel alone" : mode == 1 ? "beside f16 MFMA kernel" : "beside bf16 MFMA kernel", total, events);
        fflush(stdout);
    }
    return 0;
}
