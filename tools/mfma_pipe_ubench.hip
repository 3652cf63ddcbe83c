// What one wavefront per SIMD can sustain on the matrix pipe with the operand traffic of the register-resident MLP kernel
// (csrc/mlp_reg_impl.inc): ACC accumulators in flight, three products per k-step, optionally the A operands read from LDS
// one k-step ahead (inline-asm ds_read_b128 + manual lgkmcnt), optionally 16 distinct B operands.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_pipe_ubench.hip -o /tmp/mpu && /tmp/mpu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16

struct W { h8 th[2], tl[2]; };
__device__ __forceinline__ void issue(W& w, unsigned a0)
{
    const unsigned a1 = a0 + 2048;
    asm volatile("ds_read_b128 %0, %1" : "=v"(w.th[0]) : "v"(a0) : "memory");
    asm volatile("ds_read_b128 %0, %1" : "=v"(w.th[1]) : "v"(a1) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(w.tl[0]) : "v"(a0) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(w.tl[1]) : "v"(a1) : "memory");
}
__device__ __forceinline__ void wait4(W& w) { asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(w.th[0]), "+v"(w.th[1]), "+v"(w.tl[0]), "+v"(w.tl[1]) :: "memory"); }

// MODE 0: operands in registers.  1: A operands from LDS one k-step ahead.  2: + s_barrier every 8 k-steps.  3: + 32 KB global -> LDS DMA per 8 k-steps.  NB: distinct B operands (1 or 16).  ACC: 1, 2, 4 accumulators
// ZERO: all operands 0 (same instruction stream, no toggling) -- the difference to random data is the power limit at work
// ZERO = 2: every operand element 1.0 (non-zero, but identical from one MFMA to the next)
template <int MODE, int ACC, int NB, int ZERO = 0>
__global__ __launch_bounds__(256, 1) void k(const float* in, float* out, int iters, const unsigned char* wsrc, unsigned long long* clk)
{
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8 * 4096 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = ZERO == 1 ? 0.0f : ZERO == 2 ? __builtin_bit_cast(float, 0x3C003C00u) : in[i & 1023] * 1e-3f;     // (LDS operands: modes >= 1 only)
    __syncthreads();
    h8 bh[NB], bl[NB];
    for (int s = 0; s < NB; ++s)
        for (int i = 0; i < 8; ++i) { bh[s][i] = ZERO == 1 ? (_Float16)0 : ZERO == 2 ? (_Float16)1 : (_Float16)in[(threadIdx.x * 8 + i + s) & 1023]; bl[s][i] = ZERO == 1 ? (_Float16)0 : ZERO == 2 ? (_Float16)1 : (_Float16)(in[(threadIdx.x * 8 + i + 512 + s) & 1023] * 1e-3f); }
    floatx16 acc[ACC];
    for (int j = 0; j < ACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + lane * 16;
    W cur, nxt;
    issue(cur, base);
    const int wave = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 3) {
            for (int i = wave; i < 32; i += 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + ((it & 31) * 32 + i) * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(lds + 40960 + i * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (MODE >= 1) { issue(nxt, base + ((s + 1) & 7) * 4096); wait4(cur); }
#pragma unroll
            for (int j = 0; j < ACC; ++j) acc[j] = MFMA(cur.tl[j & 1], bh[s % NB], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < ACC; ++j) acc[j] = MFMA(cur.th[j & 1], bl[s % NB], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < ACC; ++j) acc[j] = MFMA(cur.th[j & 1], bh[s % NB], acc[j], 0, 0, 0);
            if (MODE >= 1) cur = nxt;
        }
        if (MODE >= 2) {
            if (MODE >= 3) __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float sum = 0.f;
    for (int j = 0; j < ACC; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int MODE, int ACC, int NB, int ZERO = 0>
void run(int iters, const float* din, float* dout, const unsigned char* wsrc = nullptr)
{
    static unsigned long long* clk = nullptr;
    if (!clk) hipMalloc(&clk, 16);
    const int blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, ACC, NB, ZERO>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, ACC, NB, ZERO>), dim3(blocks), dim3(256), 100 * 1024, 0, din, dout, iters, wsrc, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, ACC, NB, ZERO>), dim3(blocks), dim3(256), 100 * 1024, 0, din, dout, iters, wsrc, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 8 * 3 * ACC;
    unsigned long long hc[2];
    hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("mode=%d acc=%d nb=%d%s: %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per wave   s_memtime/s_memrealtime = %.3f (x100 MHz)\n", MODE, ACC, NB, ZERO == 1 ? " zeros" : ZERO == 2 ? " ones" : "", ms, (double)blocks * 4 * n * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12, ms * 1e6 / n, (double)hc[0] / (double)hc[1]);
}

int main()
{
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    float *din, *dout;
    hipMalloc(&din, 4096); hipMalloc(&dout, 256 * 256 * 4);
    hipMemcpy(din, h.data(), 4096, hipMemcpyHostToDevice);
    unsigned char* wsrc;
    hipMalloc(&wsrc, 32 * 32 * 1024);
    hipMemset(wsrc, 0, 32 * 32 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 1, 1>(4000, din, dout);
        run<0, 2, 1>(4000, din, dout);
        run<0, 4, 1>(4000, din, dout);
        run<0, 2, 16>(4000, din, dout);
        run<1, 2, 1>(4000, din, dout);
        run<1, 2, 16>(4000, din, dout);
        run<1, 4, 16>(4000, din, dout);
        run<0, 2, 16, 1>(4000, din, dout);
        run<0, 2, 16, 2>(4000, din, dout);
        run<2, 2, 16>(4000, din, dout);
        run<3, 2, 16>(4000, din, dout, wsrc);
    }
    return 0;
}
