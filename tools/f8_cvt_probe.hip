#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef short v2s __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, const float* sc, int n, unsigned* o)
{
    int i = threadIdx.x;
    if (i < n) {
        v2s old = {0, 0};
        v2s r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, x[i], -x[i], sc[i], false);
        o[i] = __builtin_bit_cast(unsigned, r);
    }
}
static float e4m3_to_float(unsigned char b)
{
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) v = ldexpf((float)m, -9);
    else v = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}
int main()
{
    const float xs[] = {1.0f, 1000.0f, 1000.0f, 1000.0f, 3.0f, 3.0f, 1e6f, 500.0f, 0.3f, 0.3f, 0.3f, 100.f};
    const float ss[] = {1.0f, 1.0f, 4.0f, 0.25f, 0.125f, 3.0f, 1.0f, 1.0f, 0.0009765625f, 1.5f, 1024.0f, 0.5f};
    const int n = 12;
    float *dx, *ds; unsigned* dout; unsigned h[16];
    (void)hipMalloc((void**)&dx, 64); (void)hipMalloc((void**)&ds, 64); (void)hipMalloc((void**)&dout, 64);
    (void)hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(ds, ss, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, ds, n, dout);
    (void)hipMemcpy(h, dout, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("cvt_scalef32_pk_fp8_f32(x=%g, -x, scale=%g) = 0x%08x -> %g, %g\n", xs[i], ss[i], h[i], e4m3_to_float(h[i] & 0xff), e4m3_to_float((h[i] >> 8) & 0xff));
    return 0;
}
