// How fast can ONE wavefront issue vector-ALU instructions on gfx950, as a function of the independent chains (ILP) in its
// instruction stream and of the wavefronts sharing its SIMD?  Decides whether the frame kernel's sample wavefronts (2 per
// SIMD next to an MLP wavefront, mostly dependent scalar-style chains) are bound by dependent-issue latency.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_ilp_ubench.hip -o /tmp/vub && /tmp/vub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// OP 0: v_fma_f32   1: v_pk_fma_f32   2: v_mov_b32 dpp (quad_perm) + v_add   3: v_rcp_f32 (transcendental)   4: v_mul_lo_u32
template <int ILP, int OP>
__global__ void k(const float* in, float* out, int iters, unsigned long long* clk)
{
    float x[ILP];
    float x2[ILP];
    for (int j = 0; j < ILP; ++j) { x[j] = in[(threadIdx.x + j) & 255]; x2[j] = in[(threadIdx.x + j + 7) & 255]; }
    const float a = in[300], b = in[301];
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int j = 0; j < ILP; ++j) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b));
                else if (OP == 1) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 v = {x[j], x2[j]};
                    const f2 aa = {a, a}, bb = {b, b};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(aa), "v"(bb));
                    x[j] = v.x; x2[j] = v.y;
                } else if (OP == 2) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[j]));
                else if (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[j]));
                else {
                    int v = __builtin_bit_cast(int, x[j]);
                    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v) : "v"(__builtin_bit_cast(int, a)));
                    x[j] = __builtin_bit_cast(float, v);
                }
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < ILP; ++j) s += x[j] + x2[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 16 + (threadIdx.x >> 6)] = c1 - c0;     // every wavefront's own time: the oldest one is favoured by the arbiter
}

template <int ILP, int OP>
static void run(const char* name, int waves_per_simd, const float* in, float* out, unsigned long long* clk)
{
    const int iters = 2000;
    const int threads = 256 * waves_per_simd;        // waves_per_simd wavefronts on each of the CU's 4 SIMDs
    hipLaunchKernelGGL((k<ILP, OP>), dim3(256), dim3(threads), 0, 0, in, out, iters, clk);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<ILP, OP>), dim3(256), dim3(threads), 0, 0, in, out, iters, clk);
    hipDeviceSynchronize();
    const int nw = 4 * waves_per_simd;
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), clk, 256 * 16 * 8, hipMemcpyDeviceToHost);
    double c_first = 0, c_last = 0;          // the wavefront that finishes first / last on its CU, averaged over the CUs
    for (int b = 0; b < 256; ++b) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < nw; ++w) { lo = h[b * 16 + w] < lo ? h[b * 16 + w] : lo; hi = h[b * 16 + w] > hi ? h[b * 16 + w] : hi; }
        c_first += (double)lo; c_last += (double)hi;
    }
    c_first /= 256; c_last /= 256;
    const double n = (double)iters * 16 * ILP;
    printf("%-12s ILP %d, %d wave(s)/SIMD: %6.2f cycles per instruction for the fastest wavefront, %6.2f for the slowest; %5.2f per instruction per SIMD\n", name, ILP,
           waves_per_simd, c_first / n, c_last / n, c_last / n / waves_per_simd);
}

int main()
{
    float *in, *out;
    unsigned long long* clk;
    hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&clk, 8 * 256 * 16);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0.5f + 0.001f * i;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
#define ALL(OP, NAME) \
    run<1, OP>(NAME, 1, in, out, clk); run<2, OP>(NAME, 1, in, out, clk); run<4, OP>(NAME, 1, in, out, clk); run<8, OP>(NAME, 1, in, out, clk); \
    run<1, OP>(NAME, 2, in, out, clk); run<2, OP>(NAME, 2, in, out, clk); run<1, OP>(NAME, 3, in, out, clk); run<1, OP>(NAME, 4, in, out, clk); run<4, OP>(NAME, 4, in, out, clk);
    ALL(0, "v_fma_f32")
    ALL(1, "v_pk_fma_f32")
    ALL(2, "v_add dpp")
    ALL(3, "v_rcp_f32")
    ALL(4, "v_mul_lo_u32")
    return 0;
}
