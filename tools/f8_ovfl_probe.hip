#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v2s __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, int n, unsigned* o, int set)
{
    if (set) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);     // MODE.FP16_OVFL
    int i = threadIdx.x;
    if (i < n) {
        v2s old = {0, 0};
        v2s r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, x[i], -x[i], 1.0f, false);
        int r2 = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], -x[i], 0, false);
        _Float16 h = (_Float16)x[i];
        o[3 * i] = __builtin_bit_cast(unsigned, r) & 0xffff;
        o[3 * i + 1] = r2 & 0xffff;
        o[3 * i + 2] = __builtin_bit_cast(unsigned short, h);
    }
}
int main()
{
    const float xs[] = {1.0f, 448.0f, 470.0f, 500.0f, 1000.0f, 1e6f, 70000.0f, __builtin_inff(), __builtin_nanf("")};
    const int n = 9;
    float* dx; unsigned* dout; unsigned h[64];
    (void)hipMalloc((void**)&dx, 64); (void)hipMalloc((void**)&dout, 256);
    (void)hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice);
    for (int set = 0; set < 2; ++set) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, n, dout, set);
        (void)hipMemcpy(h, dout, n * 12, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i) printf("FP16_OVFL=%d x=%g: cvt_scalef32_pk_fp8 -> 0x%04x, cvt_pk_fp8 -> 0x%04x, cvt_f16 -> 0x%04x\n", set, xs[i], h[3 * i], h[3 * i + 1], h[3 * i + 2]);
    }
    return 0;
}
