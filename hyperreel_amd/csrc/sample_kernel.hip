// K2+K3 -- everything after the MLP, one lane per (ray, sample):
//   head activations -> ray/primitive intersection -> near/far mask -> per-ray sort ->
//   points -> contraction -> advect -> offset   (reference: Intersect.forward,
//   nlf/intersect/base.py:142-259; nlf/embedding/point.py:371-396,780-831)
//   -> VM feature gather -> density -> alpha/transmittance -> colour decode -> composite
//   (TensorVMNoSample.forward, nlf/nets/tensorf_no_sample.py:128-280;
//    TensorVMKeyframeTime.forward, nlf/nets/tensorf_dynamic.py:645-839;
//    raw2alpha, utils/tensorf_utils.py:242-253).
//
// CDNA4 mapping
//   * the Z samples of a ray sit in adjacent lanes of ONE wavefront (ZP = 8/16/32/64 lanes; z_channels
//     above 64 spread a ray over 2 or 4 wavefronts of the block and add LDS hand-overs),
//     so the per-ray sort is an in-register bitonic network over DPP/bpermute shuffles,
//     the transmittance is a wave-level segmented prefix product and the final colour a
//     segmented butterfly sum -- no LDS round trips, no atomics, no global intermediates;
//   * the block's slice of the MLP head is staged once through LDS with coalesced
//     dwordx4 loads; each lane then reads its P values at an odd dword stride (conflict free);
//   * feature grids are stored channel-last with density and appearance channels of a
//     plane interleaved in one texel, so the 4 bilinear taps of a sample are 2 contiguous
//     runs of 2 texels and every fetch is a 16-byte vector load; density and appearance
//     features come from the same fetch (one pass over the grid instead of the
//     reference's two);
//   * the SH / RGB decode matrix is folded with the ray's SH basis once per ray into LDS
//     (3 x C_app), so the per-sample decode is 3*C_app FMAs regardless of SH degree.
#include <cstdlib>

#include "hr_kernels.h"
#include "hr_math.h"

#define HR_FMA(a, b, c) __builtin_fmaf((a), (b), (c))

// ZP > 64 (z_channels 65..256): the lanes of one ray span ZP/64 wavefronts of the block, and the steps
// that cross a wavefront go through a 256-float LDS scratch `s_x`.  Returns the value thread
// `src_tid` of the block holds.
__device__ __forceinline__ float hr_block_exchange(float v, int src_tid, float* s_x)
{
    __syncthreads();
    s_x[threadIdx.x] = v;
    __syncthreads();
    return s_x[src_tid];
}

template <int ZP>
__device__ __forceinline__ float hr_bitonic_sort(float v, int k, float* s_x)
{
#pragma unroll
    for (int size = 2; size <= ZP; size <<= 1) {
#pragma unroll
        for (int j = size >> 1; j > 0; j >>= 1) {
            float o;
            if (j < 64) o = __shfl_xor(v, j, 64);
            else o = hr_block_exchange(v, (int)threadIdx.x ^ j, s_x);
            const bool up = ((k & size) == 0);
            const bool lower = ((k & j) == 0);
            const float mn = fminf(v, o), mx = fmaxf(v, o);
            v = (lower == up) ? mn : mx;
        }
    }
    return v;
}

// weighted sum of the 4 bilinear taps for one float4 channel group, in ATen's order
// (nw, ne, sw, se; grid_sampler_2d)
__device__ __forceinline__ float4 hr_bilerp4(const float4 v00, const float4 v01, const float4 v10, const float4 v11,
                                              float w00, float w01, float w10, float w11)
{
    float4 r;
    r.x = HR_FMA(v11.x, w11, HR_FMA(v10.x, w10, HR_FMA(v01.x, w01, v00.x * w00)));
    r.y = HR_FMA(v11.y, w11, HR_FMA(v10.y, w10, HR_FMA(v01.y, w01, v00.y * w00)));
    r.z = HR_FMA(v11.z, w11, HR_FMA(v10.z, w10, HR_FMA(v01.z, w01, v00.z * w00)));
    r.w = HR_FMA(v11.w, w11, HR_FMA(v10.w, w10, HR_FMA(v01.w, w01, v00.w * w00)));
    return r;
}

__device__ __forceinline__ float4 hr_lerp4(const float4 v0, const float4 v1, float w0, float w1)
{
    float4 r;
    r.x = HR_FMA(v1.x, w1, v0.x * w0);
    r.y = HR_FMA(v1.y, w1, v0.y * w0);
    r.z = HR_FMA(v1.z, w1, v0.z * w0);
    r.w = HR_FMA(v1.w, w1, v0.w * w0);
    return r;
}

// One float4 channel group of a sample: plane tap x (line | time-plane) tap, then density partial
// sum (q < cd) or appearance decode through the ray's matrix M.
__device__ __forceinline__ void hr_consume_group(const HrGridPlane& g, int q, int cd, const float4 pa, const float4 pb, const float* M,
                                                 int CA, float& sig_feat, float& pre0, float& pre1, float& pre2)
{
    const float fx = pa.x * pb.x, fy = pa.y * pb.y, fz = pa.z * pb.z, fw = pa.w * pb.w;
    if (q < cd) {
        sig_feat = sig_feat + fx; sig_feat = sig_feat + fy; sig_feat = sig_feat + fz; sig_feat = sig_feat + fw;
    } else {
        const int ch = g.app_off + 4 * (q - cd);
        const float4 m0 = *reinterpret_cast<const float4*>(M + ch);
        const float4 m1 = *reinterpret_cast<const float4*>(M + CA + ch);
        const float4 m2 = *reinterpret_cast<const float4*>(M + 2 * CA + ch);
        pre0 = HR_FMA(m0.w, fw, HR_FMA(m0.z, fz, HR_FMA(m0.y, fy, HR_FMA(m0.x, fx, pre0))));
        pre1 = HR_FMA(m1.w, fw, HR_FMA(m1.z, fz, HR_FMA(m1.y, fy, HR_FMA(m1.x, fx, pre1))));
        pre2 = HR_FMA(m2.w, fw, HR_FMA(m2.z, fz, HR_FMA(m2.y, fy, HR_FMA(m2.x, fx, pre2))));
    }
}

typedef _Float16 hr_half8 __attribute__((ext_vector_type(8)));

// The three plane pairs of a VM decomposition sample the SAME three axis coordinates: plane j spans axes
// (MAT[j][0], MAT[j][1]) and its line / time plane runs along VEC[j] (tensorf_base.py:231-232, tensorf_dynamic.py:48),
// every axis always at that axis' grid size.  So a sample needs three 1-D taps (+ one along the keyframes), computed
// once, instead of three per plane pair.
struct HrAxisTaps {
    hr_axis_tap ax[3];          // x @ grid[0], y @ grid[1], z @ grid[2]
    hr_axis_tap t;              // keyframe axis (video) -- unused otherwise
};

template <int J> struct HrPlaneAxes {
    static constexpr int A0 = (J == 2) ? 1 : 0;           // MAT_MODE[j][0]: 0, 0, 1
    static constexpr int A1 = (J == 0) ? 1 : 2;           // MAT_MODE[j][1]: 1, 2, 2
    static constexpr int V = 2 - J;                       // VEC_MODE[j] = MAT_MODE_TIME[j][0]: 2, 1, 0
};

// How the gather is compiled per kernel variant.  fp32 texels: the lanes of a quad (or pair) cooperate on one sample
// at a time (hr_gather_plane_coop below); float16 texels: every lane gathers its own sample, one 16-byte load per two
// channel groups.  Both at 4 workgroups/CU.  Measured sample stage, ms per 800x800 frame:
//                            own-sample, group-major   own-sample, 2 groups/tap   cooperative
//   DoNeRF Z=32 static              1.18                      1.39                   1.12
//   technicolor Z=32 keyframe       1.13                       -                     0.94
//   immersive Z=32 keyframe         1.58                      1.75                   1.34
//   neural_3d Z=64 keyframe         4.55 (L1 hit rate 61 %)   3.95                   2.93
template <int ZP, bool HALF>
struct HrGatherTune {
#ifdef HR_SAMPLE_MIN_BLOCKS_FP32
    static constexpr int MIN_BLOCKS = HR_SAMPLE_MIN_BLOCKS_FP32;
#else
    static constexpr int MIN_BLOCKS = 4;     // 110-120 VGPRs; 5 workgroups/CU (96) spills in both texel formats
#endif
};

// acc[j] (+)= w * texel[q0 + j] for the first min(nb, G) channel groups of one tap; `off` is the
// BYTE offset of the tap's texel.  FIRST: acc = v * w, else acc = fma(v, w, acc) (ATen's bilinear order).
template <bool HALF, int G, bool FIRST>
__device__ __forceinline__ void hr_tap(const void* base, unsigned off, int q0, int nb, float w, float4 (&acc)[G])
{
    auto put = [&](int j, const float4 v) {
        if (FIRST) {
            acc[j].x = v.x * w; acc[j].y = v.y * w; acc[j].z = v.z * w; acc[j].w = v.w * w;
        } else {
            acc[j].x = HR_FMA(v.x, w, acc[j].x); acc[j].y = HR_FMA(v.y, w, acc[j].y);
            acc[j].z = HR_FMA(v.z, w, acc[j].z); acc[j].w = HR_FMA(v.w, w, acc[j].w);
        }
    };
    if constexpr (!HALF) {
        const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (size_t)(off + 16u * (unsigned)q0));
        float4 v[G];
#pragma unroll
        for (int j = 0; j < G; ++j)
            if (j < nb) v[j] = p[j];
#pragma unroll
        for (int j = 0; j < G; ++j)
            if (j < nb) put(j, v[j]);
    } else {
        const hr_half8* p = reinterpret_cast<const hr_half8*>(reinterpret_cast<const char*>(base) + (size_t)(off + 16u * (unsigned)(q0 >> 1)));
        constexpr int NO = (G + 1) / 2;
        hr_half8 h[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o)
            if (2 * o < nb) h[o] = p[o];
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            if (2 * o < nb) put(2 * o, make_float4((float)h[o][0], (float)h[o][1], (float)h[o][2], (float)h[o][3]));
            if (2 * o + 1 < nb && 2 * o + 1 < G) put(2 * o + 1, make_float4((float)h[o][4], (float)h[o][5], (float)h[o][6], (float)h[o][7]));
        }
    }
}

// All channel groups of one plane pair for a sample at normalised coordinates pn: bilinear
// plane tap x (line | time-plane) tap, density partial sum and appearance decode.
// HALF: float16 texels, one 16-byte load brings two channel groups (half the load instructions and
// half the bytes); values are widened to fp32 before any arithmetic.
// (Measured alternatives, both slower than this plain loop at 5 waves/SIMD: a 4-lanes-per-sample
//  gather with LDS hand-over, 1.66 vs 1.27 ms per frame; compile-time unrolled batches of 12-16
//  loads in flight at 4 waves/SIMD, 1.37 ms.  The gather sits at ~1.1 vector-L1 accesses per
//  clock per CU, i.e. it is bound by the tag-lookup rate for scattered 16-byte reads.)
template <bool HALF, int G, int J>
__device__ __forceinline__ void hr_gather_plane(const HrGridPlane& g, const HrAxisTaps& at, const float* M, int CA,
                                                float& sig_feat, float& pre0, float& pre1, float& pre2)
{
    const int ng = g.cd4 + g.ca4;
    const int cd = g.cd4;
    if (ng == 0) return;
    const int tex = g.tex;
    const hr_axis_tap tx = at.ax[HrPlaneAxes<J>::A0];
    const hr_axis_tap ty = at.ax[HrPlaneAxes<J>::A1];
    // ATen: nw = (x1-ix)(y1-iy), ne = (ix-x0)(y1-iy), sw = (x1-ix)(iy-y0), se = (ix-x0)(iy-y0)
    const float w00 = tx.w0 * ty.w0, w01 = tx.w1 * ty.w0, w10 = tx.w0 * ty.w1, w11 = tx.w1 * ty.w1;
    const bool line = (g.bw == 1);
    // line: grid x == 0 on a width-1 image puts weight exactly 1 on column 0 -> 2 taps along the axis;
    // time plane: x = spatial coordinate, y = keyframe time -> 4 taps
    const hr_axis_tap bxp = at.ax[HrPlaneAxes<J>::V];
    const hr_axis_tap byp = at.t;
    const float v00 = bxp.w0 * byp.w0, v01 = bxp.w1 * byp.w0, v10 = bxp.w0 * byp.w1, v11 = bxp.w1 * byp.w1;
    // BYTE offsets of the taps' texels, unsigned 32-bit: a load is then `uniform base + zero-extended VGPR offset` and
    // needs no 64-bit address arithmetic
    const unsigned esz = HALF ? 2u : 4u, texb = (unsigned)tex * esz;
    const unsigned ia00 = (unsigned)(ty.i0 * g.aw + tx.i0) * texb, ia01 = (unsigned)(ty.i0 * g.aw + tx.i1) * texb;
    const unsigned ia10 = (unsigned)(ty.i1 * g.aw + tx.i0) * texb, ia11 = (unsigned)(ty.i1 * g.aw + tx.i1) * texb;
    const unsigned ib00 = line ? (unsigned)bxp.i0 * texb : (unsigned)(byp.i0 * g.bw + bxp.i0) * texb;
    const unsigned ib01 = line ? (unsigned)bxp.i1 * texb : (unsigned)(byp.i0 * g.bw + bxp.i1) * texb;
    const unsigned ib10 = (unsigned)(byp.i1 * g.bw + bxp.i0) * texb, ib11 = (unsigned)(byp.i1 * g.bw + bxp.i1) * texb;
    if constexpr (HALF) {
        // float16 texels: one 16-byte load brings two channel groups; octet-major order (all taps of an octet, then the
        // next octet) keeps 96 VGPRs without spills and measured 0.86 vs 1.08 ms against the tap-major form below
        const char* A = reinterpret_cast<const char*>(g.a);
        const char* B = reinterpret_cast<const char*>(g.b);
        auto ld = [](const char* p, int o) { return *reinterpret_cast<const hr_half8*>(p + (size_t)(16u * (unsigned)o)); };
        auto lo4 = [](const hr_half8 h) { return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]); };
        auto hi4 = [](const hr_half8 h) { return make_float4((float)h[4], (float)h[5], (float)h[6], (float)h[7]); };
        const int no = (ng + 1) >> 1;
        for (int o = 0; o < no; ++o) {
            const hr_half8 a00 = ld(A + ia00, o), a01 = ld(A + ia01, o), a10 = ld(A + ia10, o), a11 = ld(A + ia11, o);
            const hr_half8 b00 = ld(B + ib00, o), b01 = ld(B + ib01, o);
            hr_half8 b10 = b00, b11 = b01;
            if (!line) { b10 = ld(B + ib10, o); b11 = ld(B + ib11, o); }
            {
                const float4 pa = hr_bilerp4(lo4(a00), lo4(a01), lo4(a10), lo4(a11), w00, w01, w10, w11);
                const float4 pb = line ? hr_lerp4(lo4(b00), lo4(b01), bxp.w0, bxp.w1)
                                       : hr_bilerp4(lo4(b00), lo4(b01), lo4(b10), lo4(b11), v00, v01, v10, v11);
                hr_consume_group(g, 2 * o, cd, pa, pb, M, CA, sig_feat, pre0, pre1, pre2);
            }
            if (2 * o + 1 < ng) {
                const float4 pa = hr_bilerp4(hi4(a00), hi4(a01), hi4(a10), hi4(a11), w00, w01, w10, w11);
                const float4 pb = line ? hr_lerp4(hi4(b00), hi4(b01), bxp.w0, bxp.w1)
                                       : hr_bilerp4(hi4(b00), hi4(b01), hi4(b10), hi4(b11), v00, v01, v10, v11);
                hr_consume_group(g, 2 * o + 1, cd, pa, pb, M, CA, sig_feat, pre0, pre1, pre2);
            }
        }
        return;
    }
    // Tap-major order: per pass, up to G channel groups of ONE tap's texel are read back to back (a
    // contiguous 64-byte run for fp32 texels, 32 bytes for fp16), then the next tap.  The accumulators carry
    // ATen's summation order (nw, ne, sw, se).  Group-major order (all taps of one group, then the next group)
    // touches 6-8 different cache lines per lane between two reads of the same texel and measured an L1 hit
    // rate of only 61 % on the keyframe model with 64 samples per ray (12x the L2 requests of DoNeRF).
    for (int q0 = 0; q0 < ng; q0 += G) {
        const int nb = ng - q0;
        float4 pa[G], pb[G];
        hr_tap<HALF, G, true>(g.a, ia00, q0, nb, w00, pa);
        hr_tap<HALF, G, false>(g.a, ia01, q0, nb, w01, pa);
        hr_tap<HALF, G, false>(g.a, ia10, q0, nb, w10, pa);
        hr_tap<HALF, G, false>(g.a, ia11, q0, nb, w11, pa);
        if (line) {
            hr_tap<HALF, G, true>(g.b, ib00, q0, nb, bxp.w0, pb);
            hr_tap<HALF, G, false>(g.b, ib01, q0, nb, bxp.w1, pb);
        } else {
            hr_tap<HALF, G, true>(g.b, ib00, q0, nb, v00, pb);
            hr_tap<HALF, G, false>(g.b, ib01, q0, nb, v01, pb);
            hr_tap<HALF, G, false>(g.b, ib10, q0, nb, v10, pb);
            hr_tap<HALF, G, false>(g.b, ib11, q0, nb, v11, pb);
        }
#pragma unroll
        for (int j = 0; j < G; ++j)
            if (j < nb) hr_consume_group(g, q0 + j, cd, pa[j], pb[j], M, CA, sig_feat, pre0, pre1, pre2);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Quad-cooperative gather (fp32 texels).  The vector-memory address unit charges a scattered 16-byte lane access
// about a cycle, but serves the 4 lanes of a quad reading 64 CONTIGUOUS bytes at more than twice that rate
// (tools/gather_ubench.hip, L1-resident: 1.4 vs 3.0 lane-loads per clock per CU; L2-resident 1.0 vs 1.7).
// So for the gather the four lanes of a quad stop working on their own samples and take one channel group each
// of ONE sample at a time: the owner's taps are broadcast inside the quad with DPP moves, each lane loads its
// float4 of every tap (one 64-byte run per tap and quad), forms plane x line for its group, and the four partial
// results (density sum, three decode dot products) are summed with two DPP steps and kept by the owner.
// The quad's lanes are consecutive samples of one ray (ZP >= 8), so they share the ray's decode matrix M.
template <int CTRL>
__device__ __forceinline__ float hr_dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int hr_dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

struct HrTaps {                 // one plane pair's taps of one sample: BYTE offsets of the texels (unsigned 32-bit, so
    unsigned ia[4];             //   that a load is `uniform base + zero-extended VGPR offset`: no 64-bit address
    float wa[4];                //   arithmetic per load) and weights.  plane: nw, ne, sw, se
    unsigned ib[4];             // line: low, high (2 used) / time plane: 4
    float wb[4];
};

template <int J>
__device__ __forceinline__ HrTaps hr_make_taps(const HrGridPlane& g, const HrAxisTaps& at)
{
    HrTaps t;
    const unsigned tex = (unsigned)g.tex * 4u;            // bytes per fp32 texel
    const hr_axis_tap tx = at.ax[HrPlaneAxes<J>::A0];
    const hr_axis_tap ty = at.ax[HrPlaneAxes<J>::A1];
    t.wa[0] = tx.w0 * ty.w0; t.wa[1] = tx.w1 * ty.w0; t.wa[2] = tx.w0 * ty.w1; t.wa[3] = tx.w1 * ty.w1;
    t.ia[0] = (unsigned)(ty.i0 * g.aw + tx.i0) * tex; t.ia[1] = (unsigned)(ty.i0 * g.aw + tx.i1) * tex;
    t.ia[2] = (unsigned)(ty.i1 * g.aw + tx.i0) * tex; t.ia[3] = (unsigned)(ty.i1 * g.aw + tx.i1) * tex;
    const bool line = (g.bw == 1);
    const hr_axis_tap bxp = at.ax[HrPlaneAxes<J>::V];
    const hr_axis_tap byp = at.t;
    if (line) {
        t.ib[0] = (unsigned)bxp.i0 * tex; t.ib[1] = (unsigned)bxp.i1 * tex; t.ib[2] = 0; t.ib[3] = 0;
        t.wb[0] = bxp.w0; t.wb[1] = bxp.w1; t.wb[2] = 0.0f; t.wb[3] = 0.0f;
    } else {
        t.ib[0] = (unsigned)(byp.i0 * g.bw + bxp.i0) * tex; t.ib[1] = (unsigned)(byp.i0 * g.bw + bxp.i1) * tex;
        t.ib[2] = (unsigned)(byp.i1 * g.bw + bxp.i0) * tex; t.ib[3] = (unsigned)(byp.i1 * g.bw + bxp.i1) * tex;
        t.wb[0] = bxp.w0 * byp.w0; t.wb[1] = bxp.w1 * byp.w0; t.wb[2] = bxp.w0 * byp.w1; t.wb[3] = bxp.w1 * byp.w1;
    }
    return t;
}

// One owner at a time.  LPS = lanes per sample: 4 (the quad serves owner lane T of the quad) or 2 (each pair serves
// its own lane T; used for plane pairs with exactly two channel groups so that no lane idles).
template <int LPS, int T>
__device__ __forceinline__ void hr_gather_coop_step(const HrGridPlane& g, const HrTaps& mine, bool my_valid, const float* M, int CA,
                                                    float& sig_feat, float& pre0, float& pre1, float& pre2)
{
    // quad_perm selecting the owner: [T,T,T,T] for quads, [T,T,2+T,2+T] for pairs
    constexpr int B = (LPS == 4) ? T * 0x55 : (T | (T << 2) | ((2 + T) << 4) | ((2 + T) << 6));
    const int j = threadIdx.x & (LPS - 1);                // this lane's channel group within a pass
    const bool v = hr_dpp_i<B>(my_valid ? 1 : 0) != 0;
    if (!v) return;                                       // uniform inside the quad / pair
    const bool line = (g.bw == 1);
    HrTaps t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t.ia[i] = (unsigned)hr_dpp_i<B>((int)mine.ia[i]);
        t.wa[i] = hr_dpp_f<B>(mine.wa[i]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        t.ib[i] = (unsigned)hr_dpp_i<B>((int)mine.ib[i]);
        t.wb[i] = hr_dpp_f<B>(mine.wb[i]);
    }
    if (!line) {
#pragma unroll
        for (int i = 2; i < 4; ++i) {
            t.ib[i] = (unsigned)hr_dpp_i<B>((int)mine.ib[i]);
            t.wb[i] = hr_dpp_f<B>(mine.wb[i]);
        }
    }
    const int ng = g.cd4 + g.ca4, cd = g.cd4;
    const char* A = reinterpret_cast<const char*>(g.a);
    const char* Bp = reinterpret_cast<const char*>(g.b);
    float s = 0.0f, p0 = 0.0f, p1 = 0.0f, p2 = 0.0f;
    for (int q = j; q < ng; q += LPS) {                   // this lane's channel group(s)
        const unsigned qb = 16u * (unsigned)q;
        auto ld = [qb](const char* p, unsigned off) { return *reinterpret_cast<const float4*>(p + (size_t)(off + qb)); };
        const float4 pa = hr_bilerp4(ld(A, t.ia[0]), ld(A, t.ia[1]), ld(A, t.ia[2]), ld(A, t.ia[3]), t.wa[0], t.wa[1], t.wa[2], t.wa[3]);
        const float4 pb = line ? hr_lerp4(ld(Bp, t.ib[0]), ld(Bp, t.ib[1]), t.wb[0], t.wb[1])
                               : hr_bilerp4(ld(Bp, t.ib[0]), ld(Bp, t.ib[1]), ld(Bp, t.ib[2]), ld(Bp, t.ib[3]), t.wb[0], t.wb[1], t.wb[2], t.wb[3]);
        hr_consume_group(g, q, cd, pa, pb, M, CA, s, p0, p1, p2);
    }
    // sum over the cooperating lanes: neighbour, then (quads) the other pair
    s += hr_dpp_f<0xB1>(s); p0 += hr_dpp_f<0xB1>(p0); p1 += hr_dpp_f<0xB1>(p1); p2 += hr_dpp_f<0xB1>(p2);
    if (LPS == 4) { s += hr_dpp_f<0x4E>(s); p0 += hr_dpp_f<0x4E>(p0); p1 += hr_dpp_f<0x4E>(p1); p2 += hr_dpp_f<0x4E>(p2); }
    if (j == T) { sig_feat += s; pre0 += p0; pre1 += p1; pre2 += p2; }
}

template <int J>
__device__ __forceinline__ void hr_gather_plane_coop(const HrGridPlane& g, const HrAxisTaps& at, bool valid, const float* M, int CA,
                                                     float& sig_feat, float& pre0, float& pre1, float& pre2)
{
    const int ng = g.cd4 + g.ca4;
    if (ng == 0) return;
    if (ng == 1) {               // a single 16-byte group per texel: nothing to share
        if (valid) hr_gather_plane<false, 1, J>(g, at, M, CA, sig_feat, pre0, pre1, pre2);
        return;
    }
    const HrTaps mine = hr_make_taps<J>(g, at);
    if (ng == 2) {
        hr_gather_coop_step<2, 0>(g, mine, valid, M, CA, sig_feat, pre0, pre1, pre2);
        hr_gather_coop_step<2, 1>(g, mine, valid, M, CA, sig_feat, pre0, pre1, pre2);
    } else {
        hr_gather_coop_step<4, 0>(g, mine, valid, M, CA, sig_feat, pre0, pre1, pre2);
        hr_gather_coop_step<4, 1>(g, mine, valid, M, CA, sig_feat, pre0, pre1, pre2);
        hr_gather_coop_step<4, 2>(g, mine, valid, M, CA, sig_feat, pre0, pre1, pre2);
        hr_gather_coop_step<4, 3>(g, mine, valid, M, CA, sig_feat, pre0, pre1, pre2);
    }
}

template <int ZP, bool HALF>
__global__ __launch_bounds__(256, (HrGatherTune<ZP, HALF>::MIN_BLOCKS)) void hr_sample_kernel(const hr_config* __restrict__ cfgp, const HrSampleArgs a)
{
    // the configuration lives in device memory (2 KB: too large to index dynamically as a by-value kernel argument
    // without the compiler copying it to scratch); uniform reads of it become scalar loads
    const hr_config& cfg = *cfgp;
    constexpr int RPB = 256 / ZP;   // rays per block
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Z = cfg.z_channels;
    const int P = cfg.preds_per_z;
    const int CA = a.ca_total;                   // padded appearance slots (multiple of 4)
    const int HS = a.nq * 4 + 4;                   // LDS row stride of a head row (+4: conflict-free float4 fills)
    const int RPR = a.rows_per_ray;                // head rows per ray (1 unless the head comes from a point MLP)
    float* s_head = lds;                           // [RPB * RPR][HS]
    float* s_M = lds + RPB * RPR * HS;             // [RPB][3][CA]
    float* s_x = s_M + RPB * 3 * CA;               // [256] cross-wave scratch, ZP > 64 only
    constexpr int ZW = (ZP < 64) ? ZP : 64;        // lanes of a ray inside one wavefront
    constexpr int WPR = (ZP + 63) / 64;            // wavefronts per ray

    const int tid = threadIdx.x;
    const int rib = tid / ZP;
    const int k = tid % ZP;
    // XCD-aware block order: the dispatcher places block b on XCD b % 8, so consecutive
    // blocks (neighbouring rays, overlapping texel footprints) would land on 8 different L2s.
    // Give each XCD a contiguous range of the ray list instead (bijective for any grid size).
    unsigned bid = blockIdx.x;
    if (a.dbg_mode != 2) {
        const unsigned nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int64_t ray_base = (int64_t)bid * RPB;
    const int64_t ray = ray_base + rib;
    const bool ray_ok = ray < a.n_rays;
    const bool lane_ok = ray_ok && (k < Z);

    // ---- stage this block's head into LDS: per feature quad the block's RPB rays are RPB x 16
    //      contiguous bytes in the HQ layout (RPB divides 64, so a block never straddles a 64-ray group)
    if (RPR == 1) {
        const float4* src4 = reinterpret_cast<const float4*>(a.head) + ((size_t)(ray_base >> 6) * a.nq << 6) + (ray_base & 63);
        const int total = a.nq * RPB;
        for (int i = tid; i < total; i += 256) {
            const int q = i / RPB, r = i - q * RPB;
            *reinterpret_cast<float4*>(s_head + r * HS + 4 * q) = src4[((size_t)q << 6) + r];
        }
    } else {                                       // cascade: RPB * RPR rows of the point MLP's head, any alignment
        const float4* src4 = reinterpret_cast<const float4*>(a.head);
        const int NR = RPB * RPR;
        const int64_t row0 = ray_base * RPR, n_rows = a.n_rays * RPR;
        for (int i = tid; i < a.nq * NR; i += 256) {
            const int q = i / NR, r = i - q * NR;
            const int64_t row = row0 + r;
            if (row < n_rows) *reinterpret_cast<float4*>(s_head + r * HS + 4 * q) = src4[hr_head_index(row, 4 * q, a.nq) >> 2];
        }
    }

    // ---- per-ray quantities (computed redundantly by the ray's lanes)
    float ro[3] = {0.f, 0.f, 0.f}, rd[3] = {0.f, 0.f, 1.f}, vd[3] = {0.f, 0.f, 1.f};
    float t_ray = 0.0f;
    if (ray_ok) {
        const float* r = a.rays + ray * cfg.ray_dim;
        ro[0] = r[0] - cfg.isect_origin[0];        // base.py:143-149
        ro[1] = r[1] - cfg.isect_origin[1];
        ro[2] = r[2] - cfg.isect_origin[2];
        rd[0] = r[3]; rd[1] = r[4]; rd[2] = r[5];
        vd[0] = r[3]; vd[1] = r[4]; vd[2] = r[5];  // viewdirs, point.py:868-869
        t_ray = r[cfg.ray_dim - 1];                // rays[..., -1], point.py:783
    }

    // ---- decode matrix of this ray: M[c][ch] (RGB: basis_mat rows; SH: sum_j sh_j(d) * basis row c*9+j)
    {
        float sh[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (cfg.shading == HR_SHADING_SH) hr_sh_deg2(vd[0], vd[1], vd[2], sh);
        float* M = s_M + rib * 3 * CA;
        const int nat = a.n_basis_cols;
        for (int c = 0; c < 3; ++c)
        for (int pos = k; pos < CA; pos += ZP) {          // (no integer division by the runtime CA)
            const int e = c * CA + pos;
            // padded slot -> column of basis_mat (the reference concatenates only real channels)
            int col = -1;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int rel = pos - a.planes[j].app_off;
                if (rel >= 0 && rel < a.planes[j].app_real && a.planes[j].ca4 > 0) col = a.planes[j].app_real_off + rel;
            }
            float v = 0.0f;
            if (col >= 0) {
                if (cfg.shading == HR_SHADING_SH) {
#pragma unroll
                    for (int j = 0; j < 9; ++j) v = HR_FMA(sh[j], a.basis[(c * 9 + j) * nat + col], v);
                } else {
                    v = a.basis[c * nat + col];
                }
            }
            M[e] = v;
        }
    }
    __syncthreads();

    // sample k's P head values: row k / M of the ray, columns (k % M) * P ..  (M == Z when RPR == 1)
    const int kk = lane_ok ? k : 0;
    const float* hk = s_head + rib * HS + kk * P;
    if (RPR != 1) {                                        // cascades only: integer division by a runtime value
        const int Mz = Z / RPR;
        hk = s_head + (rib * RPR + kk / Mz) * HS + (kk % Mz) * P;
    }

    // ---- distances: intersect + mask, then sort along the ray (base.py:152-210)
    float dist = __builtin_inff();
    if (lane_ok) dist = hr_sample_distance(cfg, hk, k, ro, rd);
    if (cfg.sort) dist = hr_bitonic_sort<ZP>(dist, k, s_x);

    // ---- points, contraction, advect, offset
    float oc[3] = {0.f, 0.f, 0.f};
    if (cfg.contract_type != HR_CONTRACT_IDENTITY) hr_contract_point(cfg, ro[0], ro[1], ro[2], oc);
    float base_t = 0.0f, time_off = 0.0f;
    if (cfg.advect) {
        base_t = hr_base_time(cfg, t_ray);
        time_off = t_ray - base_t;
    }
    float p[3] = {0.f, 0.f, 0.f};
    float dist_c = 0.0f;
    if (lane_ok) hr_sample_point(cfg, hk, dist, ro, rd, oc, time_off, p, &dist_c);

    // deltas (tensorf_no_sample.py:137-144)
    float dist_next;
    if constexpr (ZP > 64) dist_next = hr_block_exchange(dist_c, min(tid + 1, 255), s_x);
    else dist_next = __shfl_down(dist_c, 1, 64);
    const float delta = (k == Z - 1) ? 1e10f : (dist_next - dist_c);

    // ---- feature gather
    const bool valid = lane_ok && hr_sample_valid(cfg, p, dist_c) && (a.dbg_mode != 1) && (a.rows_out == nullptr);
    float sig_feat = 0.0f;
    float pre0 = 0.0f, pre1 = 0.0f, pre2 = 0.0f;
    if constexpr (!HALF) {      // all lanes take part: the quad's lanes serve each other's samples
        float pn[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (valid) {
            pn[0] = hr_normalize_coord(cfg, p[0], 0);
            pn[1] = hr_normalize_coord(cfg, p[1], 1);
            pn[2] = hr_normalize_coord(cfg, p[2], 2);
            pn[3] = cfg.video ? hr_normalize_time(cfg, base_t) : 0.0f;
        }
        const float* M = s_M + rib * 3 * CA;
        HrAxisTaps at;
        at.ax[0] = hr_make_tap(pn[0], cfg.grid[0]);
        at.ax[1] = hr_make_tap(pn[1], cfg.grid[1]);
        at.ax[2] = hr_make_tap(pn[2], cfg.grid[2]);
        at.t = hr_make_tap(pn[3], cfg.video ? cfg.num_keyframes : 2);
        hr_gather_plane_coop<0>(a.planes[0], at, valid, M, CA, sig_feat, pre0, pre1, pre2);
        hr_gather_plane_coop<1>(a.planes[1], at, valid, M, CA, sig_feat, pre0, pre1, pre2);
        hr_gather_plane_coop<2>(a.planes[2], at, valid, M, CA, sig_feat, pre0, pre1, pre2);
    } else if (valid) {
        float pn[4];
        pn[0] = hr_normalize_coord(cfg, p[0], 0);
        pn[1] = hr_normalize_coord(cfg, p[1], 1);
        pn[2] = hr_normalize_coord(cfg, p[2], 2);
        pn[3] = cfg.video ? hr_normalize_time(cfg, base_t) : 0.0f;
        const float* M = s_M + rib * 3 * CA;
        HrAxisTaps at;
        at.ax[0] = hr_make_tap(pn[0], cfg.grid[0]);
        at.ax[1] = hr_make_tap(pn[1], cfg.grid[1]);
        at.ax[2] = hr_make_tap(pn[2], cfg.grid[2]);
        at.t = hr_make_tap(pn[3], cfg.video ? cfg.num_keyframes : 2);
        hr_gather_plane<HALF, 1, 0>(a.planes[0], at, M, CA, sig_feat, pre0, pre1, pre2);
        hr_gather_plane<HALF, 1, 1>(a.planes[1], at, M, CA, sig_feat, pre0, pre1, pre2);
        hr_gather_plane<HALF, 1, 2>(a.planes[2], at, M, CA, sig_feat, pre0, pre1, pre2);
    }

    // ---- density -> alpha -> transmittance -> weight (raw2alpha, tensorf_utils.py:242-253)
    const float sigma = valid ? hr_density(cfg, sig_feat) : 0.0f;
    const float alpha = lane_ok ? (1.0f - HR_EXP(-sigma * (delta * cfg.distance_scale))) : 0.0f;
    float inc = lane_ok ? ((1.0f - alpha) + 1e-10f) : 1.0f;
    const int kw = k & (ZW - 1);                   // position inside this wavefront's part of the ray
#pragma unroll
    for (int d = 1; d < ZW; d <<= 1) {
        const float o = __shfl_up(inc, d, 64);
        if (kw >= d) inc = inc * o;
    }
    float before = 1.0f;                           // product over the ray's earlier wavefronts
    if constexpr (ZP > 64) {
        __syncthreads();
        if ((tid & 63) == 63) s_x[tid >> 6] = inc;
        __syncthreads();
        const int w = tid >> 6, w0 = (w / WPR) * WPR;
        for (int i = w0; i < w; ++i) before = before * s_x[i];
        inc = inc * before;
    }
    float T = __shfl_up(inc, 1, 64);
    if (kw == 0) T = before;                       // == 1 for the ray's first sample
    const float weight = alpha * T;

    // ---- colour decode (+ per-sample scale/shift) and front-to-back sum
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    if (lane_ok) {
        float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
        if (weight > cfg.weight_thresh) {          // app_mask, tensorf_no_sample.py:201
            if (cfg.shading == HR_SHADING_SH) {    // SHRender, tensorf_utils.py:334-338
                r0 = fmaxf(pre0 + 0.5f, 0.0f); r1 = fmaxf(pre1 + 0.5f, 0.0f); r2 = fmaxf(pre2 + 0.5f, 0.0f);
            } else {                               // RGBRender, tensorf_utils.py:341-343
                r0 = HR_RCP(1.0f + HR_EXP(-pre0)); r1 = HR_RCP(1.0f + HR_EXP(-pre1)); r2 = HR_RCP(1.0f + HR_EXP(-pre2));
            }
        }
        if (cfg.f_color_scale.offset >= 0) {       // scale_shift_color_all, tensorf_utils.py:267-273
            const hr_head_field& fs = cfg.f_color_scale;
            const hr_head_field& fh = cfg.f_color_shift;
            r0 = r0 * (hr_apply_act_post(fs.act, hk[fs.offset + 0]) + 1.0f) + hr_apply_act_post(fh.act, hk[fh.offset + 0]);
            r1 = r1 * (hr_apply_act_post(fs.act, hk[fs.offset + 1]) + 1.0f) + hr_apply_act_post(fh.act, hk[fh.offset + 1]);
            r2 = r2 * (hr_apply_act_post(fs.act, hk[fs.offset + 2]) + 1.0f) + hr_apply_act_post(fh.act, hk[fh.offset + 2]);
        }
        c0 = weight * r0; c1 = weight * r1; c2 = weight * r2;
    }
    float acc_w = lane_ok ? weight : 0.0f;
#pragma unroll
    for (int d = ZW >> 1; d > 0; d >>= 1) {
        c0 += __shfl_xor(c0, d, 64);
        c1 += __shfl_xor(c1, d, 64);
        c2 += __shfl_xor(c2, d, 64);
        acc_w += __shfl_xor(acc_w, d, 64);
    }
    if constexpr (ZP > 64) {                       // add the ray's wavefronts in order
        __syncthreads();
        if ((tid & 63) == 0) {
            float* o = s_x + 4 * (tid >> 6);
            o[0] = c0; o[1] = c1; o[2] = c2; o[3] = acc_w;
        }
        __syncthreads();
        const int w0 = ((tid >> 6) / WPR) * WPR;
        c0 = s_x[4 * w0 + 0]; c1 = s_x[4 * w0 + 1]; c2 = s_x[4 * w0 + 2]; acc_w = s_x[4 * w0 + 3];
        for (int i = 1; i < WPR; ++i) {
            c0 += s_x[4 * (w0 + i) + 0]; c1 += s_x[4 * (w0 + i) + 1]; c2 += s_x[4 * (w0 + i) + 2]; acc_w += s_x[4 * (w0 + i) + 3];
        }
    }
    if (ray_ok && k == 0 && a.rows_out == nullptr) {
        if (cfg.white_bg) {                        // tensorf_no_sample.py:236-237
            const float bg = 1.0f - acc_w;
            c0 += bg; c1 += bg; c2 += bg;
        }
        if (cfg.f_color_scale_global.offset >= 0) {   // scale_shift_color_one (tensorf_utils.py:275-281): sample 0's head
            const hr_head_field& fs = cfg.f_color_scale_global;
            const hr_head_field& fh = cfg.f_color_shift_global;
            c0 = c0 * (hr_apply_act(fs.act, hk[fs.offset + 0]) + 1.0f) + hr_apply_act(fh.act, hk[fh.offset + 0]);
            c1 = c1 * (hr_apply_act(fs.act, hk[fs.offset + 1]) + 1.0f) + hr_apply_act(fh.act, hk[fh.offset + 1]);
            c2 = c2 * (hr_apply_act(fs.act, hk[fs.offset + 2]) + 1.0f) + hr_apply_act(fh.act, hk[fh.offset + 2]);
        }
        else if (a.color_table) {                     // transform_color_one (tensorf_utils.py:308-320, point.py:588-594)
            // camera id = round(rays[..., -2]); ids outside the table are clamped (the reference would raise)
            int id = (int)rintf(a.rays[ray * cfg.ray_dim + cfg.ray_dim - 2]);
            id = min(max(id, 0), cfg.color_table_views - 1);
            const float* e = a.color_table + 12 * id;
            float t[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) t[i] = hr_apply_act(cfg.color_table_t_act, e[i]);
            const float n0 = c0 + ((c0 * t[0] + c1 * t[1]) + c2 * t[2]);
            const float n1 = c1 + ((c0 * t[3] + c1 * t[4]) + c2 * t[5]);
            const float n2 = c2 + ((c0 * t[6] + c1 * t[7]) + c2 * t[8]);
            c0 = n0 + hr_apply_act(cfg.color_table_s_act, e[9]);
            c1 = n1 + hr_apply_act(cfg.color_table_s_act, e[10]);
            c2 = n2 + hr_apply_act(cfg.color_table_s_act, e[11]);
        }
        a.rgb[ray * 3 + 0] = fminf(fmaxf(c0, 0.0f), 1.0f);   // eval-mode clamp, :246-247
        a.rgb[ray * 3 + 1] = fminf(fmaxf(c1, 0.0f), 1.0f);
        a.rgb[ray * 3 + 2] = fminf(fmaxf(c2, 0.0f), 1.0f);
    }

    // ---- coarse pass of a cascade: emit the point MLP's input row of this (sorted) sample (point.py:142-157)
    if (a.rows_out && lane_ok) {
        float* row = a.rows_out + (ray * Z + k) * a.row_dim;
        const float* r = a.rays + ray * cfg.ray_dim;
        int c = 0;
        for (int i = 0; i < a.n_row_inputs; ++i) {
            const int kind = a.row_kind[i];
            for (int j = 0; j < a.row_len[i]; ++j)
                row[c++] = (kind == HR_PIN_POINTS) ? (j == 0 ? p[0] : j == 1 ? p[1] : p[2])
                           : (kind == HR_PIN_VIEWDIRS) ? r[3 + j] : (kind == HR_PIN_ORIGINS) ? r[j] : r[cfg.ray_dim - 1];
        }
    }

    // ---- optional diagnostics
    if (lane_ok) {
        const int64_t s = ray * Z + k;
        if (a.fields.distances_dev) a.fields.distances_dev[s] = dist_c;
        if (a.fields.points_dev) {
            a.fields.points_dev[s * 3 + 0] = p[0];
            a.fields.points_dev[s * 3 + 1] = p[1];
            a.fields.points_dev[s * 3 + 2] = p[2];
        }
        if (a.fields.sigma_dev) a.fields.sigma_dev[s] = sigma;
        if (a.fields.weights_dev) a.fields.weights_dev[s] = weight;
    }
}

static size_t hr_sample_lds_bytes(int nq, int ca_total, int ZP, int rows_per_ray)
{
    const int RPB = 256 / ZP;
    return ((size_t)RPB * rows_per_ray * (nq * 4 + 4) + (size_t)RPB * 3 * ca_total + (ZP > 64 ? 256 : 0)) * sizeof(float);
}

void hr_launch_samples(const hr_config& cfg, const HrSampleArgs& args, hipStream_t stream)
{
    if (args.n_rays <= 0) return;
    const int Z = cfg.z_channels;
    int ZP = 8;
    while (ZP < Z) ZP <<= 1;
    const int RPB = 256 / ZP;
    const unsigned blocks = (unsigned)((args.n_rays + RPB - 1) / RPB);
    const size_t lds = hr_sample_lds_bytes(args.nq, args.ca_total, ZP, args.rows_per_ray);
    static const int dbg = [] { const char* e = getenv("HR_SAMPLE_DBG"); return e ? atoi(e) : 0; }();
    HrSampleArgs args2 = args;
    args2.dbg_mode = dbg;
    // few samples x many head columns can exceed the 64 KiB a kernel gets by default (e.g. 32 rays x 8 x 64 floats)
    const bool big_lds = lds > 64 * 1024;
#define HR_LAUNCH_SAMPLES(Z_) \
    do { \
        if (cfg.grid_dtype == HR_GRID_FP16) { \
            if (big_lds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hr_sample_kernel<Z_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((hr_sample_kernel<Z_, true>), dim3(blocks), dim3(256), lds, stream, args2.cfg_dev, args2); \
        } else { \
            if (big_lds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hr_sample_kernel<Z_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((hr_sample_kernel<Z_, false>), dim3(blocks), dim3(256), lds, stream, args2.cfg_dev, args2); \
        } \
    } while (0)
    switch (ZP) {
        case 8: HR_LAUNCH_SAMPLES(8); break;
        case 16: HR_LAUNCH_SAMPLES(16); break;
        case 32: HR_LAUNCH_SAMPLES(32); break;
        case 64: HR_LAUNCH_SAMPLES(64); break;
        case 128: HR_LAUNCH_SAMPLES(128); break;
        case 256: HR_LAUNCH_SAMPLES(256); break;
        default: break;  // Z > 256 is rejected by hr_model_create
    }
#undef HR_LAUNCH_SAMPLES
}
