// Does the vector-memory address unit (TA) reward quad-coalesced 16-byte gathers?  Same instruction count and bytes:
//   mode 0: every lane reads 16 B of ITS OWN random 64-byte texel (4 consecutive loads cover the texel)     [the sample kernel today]
//   mode 1: the 4 lanes of a quad read the 4 x 16 B of ONE random texel (one load covers it; 4 loads = 4 texels)
//   mode 2: the 2 lanes of a pair read 2 x 16 B (a 32-byte run) of one random texel
// hipcc --offload-arch=gfx950 -O3 tools/gather_ubench.hip -o /tmp/gub && /tmp/gub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned rnd(unsigned x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ grid, unsigned n_texels, float* out, int iters)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned s = (MODE == 0 ? tid : MODE == 1 ? (tid >> 2) : (tid >> 1)) * 2654435761u + 12345u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        // 6 "taps" of 4 loads each, like one plane pair with a 64-byte texel
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            if (MODE == 0) {
                s = rnd(s);
                const float4* p = grid + (size_t)(s % n_texels) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float4 v = p[q]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            } else if (MODE == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s = rnd(s);
                    const float4 v = grid[(size_t)(s % n_texels) * 4 + (tid & 3)];
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            } else {          // MODE 2: the 2 lanes of a pair read 2 x 16 B of one random 32-byte half texel
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s = rnd(s);
                    const float4 v = grid[(size_t)(s % n_texels) * 4 + (q & 2) + (tid & 1)];
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
        }
    }
    out[tid] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE>
void run(const float4* grid, unsigned n_texels, float* out, int iters)
{
    const int blocks = 256 * 5 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, grid, n_texels, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, grid, n_texels, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double lane_loads = (double)blocks * 256 * iters * 24;
    printf("mode %d: %.3f ms  %.2f G lane-loads/s  = %.2f lane-loads/clk/CU @2.4GHz, %.1f TB/s\n", MODE, ms, lane_loads / ms / 1e6,
           lane_loads / (ms * 1e-3) / 256 / 2.4e9, lane_loads * 16 / (ms * 1e-3) / 1e12);
}

int main()
{
    const unsigned max_texels = 46u * 1024 * 1024 / 64;   // 46 MB of 64-byte texels, like the DoNeRF grids
    float4* grid; float* out;
    hipMalloc(&grid, (size_t)max_texels * 64); hipMalloc(&out, 256 * 5 * 8 * 256 * 4);
    hipMemset(grid, 0, (size_t)max_texels * 64);
    // working set: L1-resident (16 KB), L2-resident (1 MB), the whole grid (46 MB: MALL / HBM)
    for (unsigned kb : {16u, 1024u, 46u * 1024u}) {
        const unsigned n_texels = kb * 1024 / 64;
        printf("-- working set %u KB\n", kb);
        run<0>(grid, n_texels, out, 16); run<1>(grid, n_texels, out, 16); run<2>(grid, n_texels, out, 16);
        run<0>(grid, n_texels, out, 16); run<1>(grid, n_texels, out, 16); run<2>(grid, n_texels, out, 16);
    }
    return 0;
}
