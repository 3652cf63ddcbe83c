// How should the training path's texel scatter-adds be laid out over the lanes?  Same number of fp32 atomics
// (global_atomic_add_f32, no return) in every mode, onto texels of 16 floats (64 bytes):
//   mode 0: every lane owns one random texel and adds its 16 channels one after the other         [one thread per sample]
//           -> an instruction touches 64 different cache lines
//   mode 1: 16 consecutive lanes own one random texel, one channel each                            [one thread per channel]
//           -> an instruction touches 4 cache lines, 64 contiguous bytes per 16 lanes
//   mode 2: like 1, but each lane first sums the contributions of 4 samples that hit the same texel (a proxy for
//           pre-reducing duplicates in LDS)
// on a large table (600 x 600 texels: the planes) and a small one (600 texels: the lines, heavy contention).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_ubench.hip -o /tmp/aub && /tmp/aub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned rnd(unsigned x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void k(float* __restrict__ grid, unsigned n_texels, int iters)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned s = (MODE == 0 ? tid : (tid >> 4)) * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = rnd(s);
        float* p = grid + (size_t)(s % n_texels) * 16;
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) unsafeAtomicAdd(p + c, 1.0f);
        } else {
            unsafeAtomicAdd(p + (tid & 15), 1.0f);
        }
    }
}

template <int MODE>
static void run(const char* what, float* grid, unsigned n_texels)
{
    // 2^26 atomics in every mode
    const int threads = 1 << 20;
    const int iters = (MODE == 0) ? 4 : 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(threads / 256), dim3(256), 0, 0, grid, n_texels, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(threads / 256), dim3(256), 0, 0, grid, n_texels, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = 5.0 * (double)threads * iters * (MODE == 0 ? 16 : 1);
    printf("%-44s mode %d: %7.2f G atomics/s  (%.3f ms per 2^26)\n", what, MODE, n / (ms * 1e-3) / 1e9, ms / 5.0);
}

int main()
{
    const unsigned big = 600 * 600, small = 600;
    float* grid = nullptr;
    hipMalloc((void**)&grid, sizeof(float) * 16 * big);
    hipMemset(grid, 0, sizeof(float) * 16 * big);
    run<0>("planes (360k texels), lane = sample", grid, big);
    run<1>("planes (360k texels), lane = channel", grid, big);
    run<0>("lines (600 texels), lane = sample", grid, small);
    run<1>("lines (600 texels), lane = channel", grid, small);
    run<0>("time plane (6036 texels), lane = sample", grid, 12 * 503);
    run<1>("time plane (6036 texels), lane = channel", grid, 12 * 503);
    hipFree(grid);
    return 0;
}
