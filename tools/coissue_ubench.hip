// Can a SIMD of gfx950 run its matrix pipe and its vector ALU at the same time from DIFFERENT wavefronts -- and at what price?
// One workgroup per CU (256 of them: realistic power), NM "matrix" wavefronts + NV "vector" wavefronts per SIMD:
//   matrix role: back-to-back v_mfma_f32_32x32x16_f16 on register operands, 4 independent accumulators (the MLP's inner loop without its loads);
//   vector role: a stream of one kind of instruction with ILP independent chains: v_fma_f32 | DPP add | v_rcp_f32 | ds_read_b128 (LDS) |
//                global_load_dwordx4 (L2-resident, scattered 16 B per lane: the gather).
// Each role is timed alone (the other role's wavefronts exit at once) and together; s_memtime per wavefront, s_memrealtime for the clock.
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_ubench.hip -o /tmp/cub && /tmp/cub
// This is what decides whether overlapping the sample-prediction MLP with the per-sample stage on one CU (frame kernel; the co-resident
// pair) can approach max(MLP, samples) or is bound by their sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// VOP 0: v_fma_f32   1: DPP add   2: v_rcp_f32   3: ds_read_b128   4: global_load_dwordx4 (scattered)
template <int NM, int NV, int ILP, int VOP>
__global__ __launch_bounds__(256 * (NM + NV)) void k(const float* in, const f4* table, float* out, int iters_m, int iters_v, int run_m, int run_v,
                                                     unsigned long long* clk)
{
    __shared__ f4 lds[1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = f4{in[i & 255], 1.f, 2.f, 3.f};
    __syncthreads();
    const bool is_m = wave < 4 * NM;              // wavefront w sits on SIMD w % 4: the first 4 * NM wavefronts are NM per SIMD
    unsigned long long c0 = 0, c1 = 0, r0 = 0, r1 = 0;
    float sink = 0.f;
    if (is_m) {
        if (!run_m) return;
        halfx8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[(lane * 8 + i) & 255]; b[i] = (_Float16)in[(lane * 8 + i + 64) & 255]; }
        floatx16 acc[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        __builtin_amdgcn_s_setprio(2);
        r0 = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters_m; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
            a[it & 7] += (_Float16)0.001f;        // (operands change: the power manager sees toggling inputs)
        }
        c1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int j = 0; j < 4; ++j) sink += acc[j][0] + acc[j][7];
    } else {
        if (!run_v) return;
        float x[ILP];
        for (int j = 0; j < ILP; ++j) x[j] = in[(threadIdx.x + j) & 255];
        const float ca = in[300], cb = in[301];
        unsigned off = (unsigned)((lane * 37 + wave * 11) & 1023);
        r0 = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters_v; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int j = 0; j < ILP; ++j) {
                    if (VOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(ca), "v"(cb));
                    else if (VOP == 1) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[j]));
                    else if (VOP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[j]));
                    else if (VOP == 3) {
                        const f4 v = lds[(off + j * 64 + r * 8) & 1023];
                        x[j] += v.x;
                    } else {
                        const f4 v = table[(size_t)((off * 2654435761u + (unsigned)(it * 8 + r) * 40503u + j * 977u) & 0xFFFFF)];      // 16 MB table: L2 / Infinity-Cache resident
                        x[j] += v.x;
                    }
                }
            }
        }
        c1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int j = 0; j < ILP; ++j) sink += x[j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if (lane == 0) {
        clk[(blockIdx.x * 32 + wave) * 2 + 0] = c1 - c0;
        clk[(blockIdx.x * 32 + wave) * 2 + 1] = r1 - r0;
    }
}

struct Res { double cyc_m, cyc_v, ghz; };

template <int NM, int NV, int ILP, int VOP>
static Res run(const float* in, const f4* table, float* out, unsigned long long* clk, int run_m, int run_v, int iters_m, int iters_v)
{
    const int threads = 256 * (NM + NV);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(clk, 0, 256 * 32 * 2 * 8);
        hipLaunchKernelGGL((k<NM, NV, ILP, VOP>), dim3(256), dim3(threads), 0, 0, in, table, out, iters_m, iters_v, run_m, run_v, clk);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(256 * 32 * 2);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cm = 0, cv = 0, ghz = 0; int nm = 0, nv = 0, ng = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 4 * (NM + NV); ++w) {
            const double c = (double)h[(b * 32 + w) * 2], r = (double)h[(b * 32 + w) * 2 + 1];
            if (c == 0) continue;
            if (w < 4 * NM) { cm += c; ++nm; } else { cv += c; ++nv; }
            if (r > 0) { ghz += c / r / 10.0; ++ng; }
        }
    return Res{nm ? cm / nm : 0, nv ? cv / nv : 0, ng ? ghz / ng : 0};
}

template <int NM, int NV, int ILP, int VOP>
static void all(const char* name, const float* in, const f4* table, float* out, unsigned long long* clk)
{
    const int im = 4000, iv = (VOP >= 3) ? 1500 : 6000;
    const double n_m = (double)im * 16, n_v = (double)iv * 8 * ILP;
    const Res a = run<NM, NV, ILP, VOP>(in, table, out, clk, 1, 0, im, iv);
    const Res b = run<NM, NV, ILP, VOP>(in, table, out, clk, 0, 1, im, iv);
    // together: the vector role sized to run about as long as the matrix role
    const int iv2 = (int)(iv * (a.cyc_m / (b.cyc_v > 0 ? b.cyc_v : 1)));
    const Res c = run<NM, NV, ILP, VOP>(in, table, out, clk, 1, 1, im, iv2 > 0 ? iv2 : 1);
    const double n_v2 = (double)(iv2 > 0 ? iv2 : 1) * 8 * ILP;
    printf("%-22s %dM+%dV/SIMD ILP %d | alone: %6.2f cyc/MFMA @%.2f GHz, %6.2f cyc/vec-instr/wave @%.2f GHz | together: %6.2f cyc/MFMA (x%.2f), %6.2f cyc/vec-instr/wave (x%.2f) @%.2f GHz\n",
           name, NM, NV, ILP, a.cyc_m / n_m, a.ghz, b.cyc_v / n_v, b.ghz, c.cyc_m / n_m, (c.cyc_m / n_m) / (a.cyc_m / n_m), c.cyc_v / n_v2,
           (c.cyc_v / n_v2) / (b.cyc_v / n_v), c.ghz);
    fflush(stdout);
}

int main()
{
    float *in, *out;
    f4* table;
    unsigned long long* clk;
    hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 2048); hipMalloc(&clk, 256 * 32 * 2 * 8); hipMalloc(&table, (size_t)16 << 20);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0.5f + 0.001f * i;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    hipMemset(table, 0, (size_t)16 << 20);
#define ROW(NV, ILP) \
    all<1, NV, ILP, 0>("v_fma_f32", in, table, out, clk); \
    all<1, NV, ILP, 1>("v_add_f32 dpp", in, table, out, clk); \
    all<1, NV, ILP, 2>("v_rcp_f32", in, table, out, clk); \
    all<1, NV, ILP, 3>("ds_read_b128", in, table, out, clk); \
    all<1, NV, ILP, 4>("global_load x4 scattered", in, table, out, clk);
    ROW(1, 1)
    ROW(2, 1)
    ROW(3, 1)
    ROW(3, 4)

    return 0;
}
