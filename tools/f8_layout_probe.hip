#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
// position p = (lane half h, byte j): p = 32 h + j.  A one-hot in row m at pa; B one-hot in row n at pb.
__global__ void probe(float* out, int m, int n)
{
    const int lane = threadIdx.x;
    for (int pa = 0; pa < 64; ++pa)
        for (int pb = 0; pb < 64; ++pb) {
            v8i a = {}, b = {};
            if (lane == 32 * (pa >> 5) + m) { unsigned char* q = (unsigned char*)&a; q[pa & 31] = 0x38; }
            if (lane == 32 * (pb >> 5) + n) { unsigned char* q = (unsigned char*)&b; q[pb & 31] = 0x38; }
            v16f acc = {};
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 100 + lane, 0, 127);
            // D[m][n]: lane = n + 32*((m/4)&1), r = (m/8)*4 + m%4
            if (lane == n + 32 * ((m >> 2) & 1)) out[pa * 64 + pb] = acc[(m >> 3) * 4 + (m & 3)];
        }
}
int main()
{
    float* d; hipMalloc((void**)&d, 64 * 64 * 4);
    static float h[64 * 64];
    for (int t = 0; t < 2; ++t) {
        const int m = t ? 5 : 0, n = t ? 9 : 0;
        hipMemset(d, 0, sizeof(h));
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, m, n);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("m=%d n=%d: for each A position pa: matching B position pb and log2(D)+127 (= the scale_a lane used + ...)\n", m, n);
        for (int pa = 0; pa < 64; ++pa) {
            printf("pa=%2d:", pa);
            for (int pb = 0; pb < 64; ++pb) if (h[pa * 64 + pb] != 0.0f) printf(" pb=%2d scale_lane=%g", pb, log2f(h[pa * 64 + pb]) + 127 - 100);
            printf("\n");
        }
    }
    return 0;
}
