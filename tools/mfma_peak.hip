// Sustained rate of v_mfma_f32_32x32x16_bf16 on this chip: register-resident operands, no memory traffic,
// WAVES waves per SIMD, ACC independent accumulators per wave.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int ACC>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)in[(threadIdx.x * 8 + i) & 1023]; b[i] = (__bf16)in[(threadIdx.x * 8 + i + 512) & 1023]; }
    floatx16 acc[ACC];
    for (int j = 0; j < ACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < ACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < ACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACC>
void run(int blocks_per_cu, int iters, const float* din, float* dout)
{
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<ACC>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<ACC>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 /*waves*/ * iters * ACC * 2.0 * 32 * 32 * 16;
    printf("acc=%d waves/SIMD=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", ACC, blocks_per_cu, iters, ms, flops / (ms * 1e-3) / 1e12);
}

int main()
{
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    float *din, *dout;
    hipMalloc(&din, 4096); hipMalloc(&dout, 256 * 8 * 256 * 4);
    hipMemcpy(din, h.data(), 4096, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<4>(1, 20000, din, dout);
        run<4>(2, 20000, din, dout);
        run<8>(2, 20000, din, dout);
        run<4>(2, 200000, din, dout);   // ~long enough to reach the sustained clock
    }
    return 0;
}
