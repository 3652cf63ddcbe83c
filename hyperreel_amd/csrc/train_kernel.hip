// Training path on the device: the three phases of hr_train.h, plus the layout kernels around them --
// MLP input features for the caller's autograd MLP, and the packed texel gradients back to the reference's (C, H, W)
// parameter layout.
//
// Mapping: a training batch is 16 384 rays (conf/experiment/training/*.yaml: batch_size), i.e. 0.5 M samples against
// the 20 M of a rendered frame, so the step is latency- and atomics-bound rather than bandwidth-bound.
//   phase A (forward, compositing backward): one thread per ray, 64-thread workgroups -> one wavefront on each of the
//     256 CUs; the ray's samples are walked with the intermediates in registers / scratch;
//   phase B (gather backward): 16 lanes per (ray, sample), ONE TEXEL CHANNEL PER LANE: texel gradients go straight to
//     HBM through hardware fp32 atomics (global_atomic_add_f32) on the channel-last packed layout, and the channels of a
//     tap are one contiguous 32..64-byte run per instruction.  The memory system retires atomics per cache-line request,
//     not per lane: tools/atomic_ubench.hip measures 331 G atomics/s this way against 19.5 G/s for 64 lanes on 64
//     different lines, and the phase measured 12.2 ms (one thread per ray) and 10.4 ms (one thread per sample, channels
//     in a loop) per DoNeRF step before.  The ray's decode matrix and its gradient live in LDS (ds_add_f32);
//   phase C (intersection backward): one thread per (ray, sample), elementwise.
#include "hr_kernels.h"
#include "hr_mask.h"
#define HR_GATHER_FENCED 1        // (the lane-per-sample forward: 90 registers instead of 102, sample_core.inc)
#include "sample_core.inc"      // the render kernels' lane-per-sample building blocks (sort, cooperative gather)
// The deterministic build of the sample-stage training kernels (train_det_kernel.hip: HR_TRAIN_DET, hr_acc_t = 64-bit fixed point, see
// hr_train.h) compiles this file a second time: everything that depends on hr_acc_t lives in its own namespace there, the exports that
// do not (rows, features, occupancy) are compiled once, in the default build.
#ifdef HR_TRAIN_DET
namespace hr_det {
#endif
#include "hr_train.h"

// Phase A walks a ray serially and is bound by the latency of that walk (every sample's gather waits on its point), not by
// issue slots: a batch of 16 384 rays in full wavefronts is ONE wavefront per CU with nothing to hide the latency behind.
// HR_TRAIN_RPW rays per wavefront (the other lanes idle) gives every SIMD several wavefronts instead.
#ifndef HR_TRAIN_RPW
#define HR_TRAIN_RPW 16
#endif
template <int ZP>
__global__ __launch_bounds__(64) void hr_train_kernel(const hr_config* __restrict__ cfgp, const HrTrainArgs a)
{
    HR_FX_ENTER(a);
    if (threadIdx.x >= HR_TRAIN_RPW) return;
    const int64_t ray = (int64_t)blockIdx.x * HR_TRAIN_RPW + threadIdx.x;
    if (ray >= a.n_rays) return;
    hr_ray_train<ZP>(*cfgp, a, ray);
}

// ---------------------------------------------------------------------------------------------------------
// Phase A, one LANE per (ray, sample) -- the mapping of the render kernels (sample_core.inc): the ZP samples of a ray sit in
// adjacent lanes, the sort is a bitonic network over lane exchanges, the transmittance an inclusive product scan, the colour a
// butterfly sum, and the compositing backward one suffix-sum scan.  Same arithmetic as hr_ray_train (which stays: ZP > 64,
// and the host build the CPU tests check against torch.autograd), with the forward's gather done by the render path's
// cooperative gather (hr_gather_844 / hr_gather_plane_coop: same values).  32x the lanes of the one-thread-per-ray walk and
// no per-lane scratch arrays.
template <int ZP>
__device__ __forceinline__ void hr_bitonic_sort_kv(float& v, int& id, int k)
{
    // ascending in (value, original index): a total order, so the result is the stable sort of hr_ray_train / sort_z
#pragma unroll
    for (int size = 2; size <= ZP; size <<= 1) {
#pragma unroll
        for (int j = size >> 1; j > 0; j >>= 1) {
            const float o = __shfl_xor(v, j, 64);
            const int oi = __shfl_xor(id, j, 64);
            const bool up = ((k & size) == 0), lower = ((k & j) == 0);
            const bool o_less = (o < v) || (o == v && oi < id);
            if ((lower == up) ? o_less : !o_less) { v = o; id = oi; }
        }
    }
}

template <int ZP, int NB, int PC>
__global__ __launch_bounds__(256) void hr_train_lanes_kernel(const hr_config* __restrict__ cfgp, const HrTrainArgs a)
{
    static_assert(ZP <= 64, "a ray inside one wavefront");
    HR_FX_ENTER(a);
    const hr_config& c = *cfgp;
    constexpr int RPB = 256 / ZP;
    extern __shared__ float lds[];                 // [RPB][3 * CA] decode matrices
    const int CA = a.ca_total, Z = c.z_channels, P = c.preds_per_z;
    const int tid = threadIdx.x;
    const int rib = tid / ZP, k = tid % ZP;
    const int64_t ray0 = (int64_t)blockIdx.x * RPB;
    const int64_t ray = ray0 + rib;
    const bool ray_ok = ray < a.n_rays;
    for (int e = tid; e < RPB * 3 * CA; e += 256) {
        const int r = e / (3 * CA), i = e - r * 3 * CA;
        float v = 0.0f;
        if (ray0 + r < a.n_rays) {
            float sh[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const float* rr = a.rays + (ray0 + r) * c.ray_dim;
            if (c.shading == HR_SHADING_SH) hr_sh_deg2(rr[3], rr[4], rr[5], sh);
            v = hr_train_decode_coef(c, a, sh, i / CA, i % CA);
        }
        lds[e] = v;
    }
    __shared__ __attribute__((aligned(16))) float s_ones[HR_GATHER_ONES];
    hr_gather_ones_init(s_ones);
    __syncthreads();
    const float* M = lds + rib * 3 * CA;
    const bool lane_ok = ray_ok && k < Z;
    const int64_t rr_ = ray_ok ? ray : 0;
    const float* r = a.rays + rr_ * c.ray_dim;
    const float* head = a.head + rr_ * (int64_t)Z * P;
    const float* hk = head + (lane_ok ? k : 0) * P;
    const float ro[3] = {r[0] - c.isect_origin[0], r[1] - c.isect_origin[1], r[2] - c.isect_origin[2]};
    const float rd[3] = {r[3], r[4], r[5]};
    const float t_ray = r[c.ray_dim - 1];

    // ---- forward
    float dist = __builtin_inff();
    int src = k;
    if (lane_ok) dist = hr_sample_distance(c, hk, k, ro, rd);
    if (c.sort) hr_bitonic_sort_kv<ZP>(dist, src, k);
    float oc[3] = {0.f, 0.f, 0.f};
    if (c.contract_type != HR_CONTRACT_IDENTITY) hr_contract_point(c, ro[0], ro[1], ro[2], oc);
    float base_t = 0.0f, time_off = 0.0f;
    if (c.advect) { base_t = hr_base_time(c, t_ray); time_off = t_ray - base_t; }
    float p[3] = {0.f, 0.f, 0.f};
    float dist_c = 0.0f;
    if (lane_ok) hr_sample_point(c, hk, dist, ro, rd, oc, time_off, p, &dist_c);
    const float dist_next = __shfl_down(dist_c, 1, 64);
    const float delta = (k == Z - 1) ? 1e10f : (dist_next - dist_c);
    const bool valid = lane_ok && hr_sample_valid(c, p, dist_c);
    float feat = 0.0f, pre0 = 0.0f, pre1 = 0.0f, pre2 = 0.0f;
    {
        float pn[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (valid) {
            pn[0] = hr_normalize_coord(c, p[0], 0);
            pn[1] = hr_normalize_coord(c, p[1], 1);
            pn[2] = hr_normalize_coord(c, p[2], 2);
            pn[3] = c.video ? hr_normalize_time(c, base_t) : 0.0f;
            if (a.d_rgb && a.tape.taps) {          // what phase B needs of this sample's position: the three axis taps
                const int64_t NS = a.n_rays * Z, s = ray * Z + k;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const hr_axis_tap_g g = hr_make_tap_g(pn[i], c.grid[i]);
                    float* t = a.tape.taps + (size_t)(6 * i) * NS + s;
                    t[0] = __builtin_bit_cast(float, g.t.i0); t[NS] = __builtin_bit_cast(float, g.t.i1);
                    t[2 * NS] = g.t.w0; t[3 * NS] = g.t.w1; t[4 * NS] = g.s0; t[5 * NS] = g.s1;
                }
            }
        }
        if constexpr (PC != 0) {
            HrAxisTapsC at;
            at.ax[0] = hr_make_tap_c(pn[0], c.grid[0]);
            at.ax[1] = hr_make_tap_c(pn[1], c.grid[1]);
            at.ax[2] = hr_make_tap_c(pn[2], c.grid[2]);
            at.t = hr_make_tap_c(pn[3], (NB == 4 && c.video) ? c.num_keyframes : 2);
            hr_gather_844<NB, PC>(a.planes, at, valid, M, s_ones, feat, pre0, pre1, pre2);
        } else {
            HrAxisTaps at;
            at.ax[0] = hr_make_tap(pn[0], c.grid[0]);
            at.ax[1] = hr_make_tap(pn[1], c.grid[1]);
            at.ax[2] = hr_make_tap(pn[2], c.grid[2]);
            at.t = hr_make_tap(pn[3], c.video ? c.num_keyframes : 2);
            hr_gather_plane_coop<0, 1, NB>(a.planes[0], at, valid, M, CA, feat, pre0, pre1, pre2);
            hr_gather_plane_coop<1, 1, NB>(a.planes[1], at, valid, M, CA, feat, pre0, pre1, pre2);
            hr_gather_plane_coop<2, 1, NB>(a.planes[2], at, valid, M, CA, feat, pre0, pre1, pre2);
        }
    }
    const float sigma = valid ? hr_density(c, feat) : 0.0f;
    const float alpha = lane_ok ? (1.0f - expf(-sigma * (delta * c.distance_scale))) : 0.0f;
    const float inc = lane_ok ? ((1.0f - alpha) + 1e-10f) : 1.0f;
    float prod = inc;                              // inclusive product over the ray's earlier lanes
#pragma unroll
    for (int d = 1; d < ZP; d <<= 1) {
        const float o = __shfl_up(prod, d, 64);
        if (k >= d) prod = prod * o;
    }
    float T = __shfl_up(prod, 1, 64);
    if (k == 0) T = 1.0f;
    const float wgt = lane_ok ? alpha * T : 0.0f;
    if (lane_ok) {                                 // optional diagnostics / regulariser inputs, same definitions as the render path's hr_fields
        const int64_t s = ray * Z + k;
        if (a.f_dist) a.f_dist[s] = dist_c;
        if (a.f_points) { a.f_points[3 * s] = p[0]; a.f_points[3 * s + 1] = p[1]; a.f_points[3 * s + 2] = p[2]; }
        if (a.f_weights) a.f_weights[s] = wgt;
    }
    const bool app = lane_ok && (wgt > c.weight_thresh);
    const float pre[3] = {pre0, pre1, pre2};
    float raw[3] = {0.f, 0.f, 0.f}, sc[3] = {1.f, 1.f, 1.f}, rr[3] = {0.f, 0.f, 0.f};
    if (lane_ok) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (app) raw[i] = (c.shading == HR_SHADING_SH) ? fmaxf(pre[i] + 0.5f, 0.0f) : 1.0f / (1.0f + expf(-pre[i]));
            rr[i] = raw[i];
            if (c.f_color_scale.offset >= 0) {
                sc[i] = hr_apply_act(c.f_color_scale.act, hk[c.f_color_scale.offset + i]) + 1.0f;
                rr[i] = raw[i] * sc[i] + hr_apply_act(c.f_color_shift.act, hk[c.f_color_shift.offset + i]);
            }
        }
    }
    float c0 = wgt * rr[0], c1 = wgt * rr[1], c2 = wgt * rr[2], acc_w = wgt;
#pragma unroll
    for (int d = ZP >> 1; d > 0; d >>= 1) {        // every lane of the ray ends up with the ray's sums
        c0 += __shfl_xor(c0, d, 64);
        c1 += __shfl_xor(c1, d, 64);
        c2 += __shfl_xor(c2, d, 64);
        acc_w += __shfl_xor(acc_w, d, 64);
    }
    if (a.white_bg) { const float bg = 1.0f - acc_w; c0 += bg; c1 += bg; c2 += bg; }
    const float cpre[3] = {c0, c1, c2};            // the composited colour before the per-ray scale / shift
    float gscale[3] = {1.0f, 1.0f, 1.0f};
    float tcol[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int cam = 0;
    const bool head_transform = c.f_color_scale_global.offset >= 0 && c.f_color_scale_global.channels == 9;
    if (head_transform) {                          // transform_color_one, the matrix from the head (`color_transform_global`): sample 0's nine values
        const hr_head_field& fs = c.f_color_scale_global;
        const hr_head_field& fh = c.f_color_shift_global;
#pragma unroll
        for (int i = 0; i < 9; ++i) tcol[i] = hr_apply_act(fs.act, head[fs.offset + i]);
        const float n0 = c0 + ((c0 * tcol[0] + c1 * tcol[1]) + c2 * tcol[2]);
        const float n1 = c1 + ((c0 * tcol[3] + c1 * tcol[4]) + c2 * tcol[5]);
        const float n2 = c2 + ((c0 * tcol[6] + c1 * tcol[7]) + c2 * tcol[8]);
        c0 = n0 + hr_apply_act(fh.act, head[fh.offset + 0]);
        c1 = n1 + hr_apply_act(fh.act, head[fh.offset + 1]);
        c2 = n2 + hr_apply_act(fh.act, head[fh.offset + 2]);
    } else if (c.f_color_scale_global.offset >= 0) {      // scale_shift_color_one (tensorf_utils.py:275-281): sample 0's head values
        const hr_head_field& fs = c.f_color_scale_global;
        const hr_head_field& fh = c.f_color_shift_global;
#pragma unroll
        for (int i = 0; i < 3; ++i) gscale[i] = hr_apply_act(fs.act, head[fs.offset + i]) + 1.0f;
        c0 = c0 * gscale[0] + hr_apply_act(fh.act, head[fh.offset + 0]);
        c1 = c1 * gscale[1] + hr_apply_act(fh.act, head[fh.offset + 1]);
        c2 = c2 * gscale[2] + hr_apply_act(fh.act, head[fh.offset + 2]);
    } else if (a.color_table) {                    // transform_color_one (tensorf_utils.py:308-320, point.py:588-594)
        cam = (int)rintf(r[c.ray_dim - 2]);
        cam = cam < 0 ? 0 : (cam > c.color_table_views - 1 ? c.color_table_views - 1 : cam);
        const float* e = a.color_table + 12 * cam;
#pragma unroll
        for (int i = 0; i < 9; ++i) tcol[i] = hr_apply_act(c.color_table_t_act, e[i]);
        const float n0 = c0 + ((c0 * tcol[0] + c1 * tcol[1]) + c2 * tcol[2]);
        const float n1 = c1 + ((c0 * tcol[3] + c1 * tcol[4]) + c2 * tcol[5]);
        const float n2 = c2 + ((c0 * tcol[6] + c1 * tcol[7]) + c2 * tcol[8]);
        c0 = n0 + hr_apply_act(c.color_table_s_act, e[9]);
        c1 = n1 + hr_apply_act(c.color_table_s_act, e[10]);
        c2 = n2 + hr_apply_act(c.color_table_s_act, e[11]);
    }
    if (a.rgb && ray_ok && k == 0) { a.rgb[ray * 3 + 0] = c0; a.rgb[ray * 3 + 1] = c1; a.rgb[ray * 3 + 2] = c2; }
    if (!a.d_rgb) return;

    // ---- backward of the compositing
    float g[3] = {0.f, 0.f, 0.f};
    if (ray_ok) { g[0] = a.d_rgb[ray * 3 + 0]; g[1] = a.d_rgb[ray * 3 + 1]; g[2] = a.d_rgb[ray * 3 + 2]; }
    float* dhead = a.d_head + rr_ * (int64_t)Z * P;
    float* dhk = dhead + (lane_ok ? k : 0) * P;
    if (lane_ok)
        for (int i = 0; i < P; ++i) dhk[i] = 0.0f;
    if (head_transform) {
        const hr_head_field& fs = c.f_color_scale_global;
        const hr_head_field& fh = c.f_color_shift_global;
        float gn[3] = {g[0], g[1], g[2]};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (lane_ok && k == 0) dhk[fh.offset + i] += g[i] * hr_act_grad(fh.act, head[fh.offset + i]);     // sample 0's row is this lane's own
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (lane_ok && k == 0) dhk[fs.offset + 3 * i + j] += g[i] * cpre[j] * hr_act_grad(fs.act, head[fs.offset + 3 * i + j]);
                gn[j] += g[i] * tcol[3 * i + j];
            }
        }
        g[0] = gn[0]; g[1] = gn[1]; g[2] = gn[2];
    } else if (c.f_color_scale_global.offset >= 0) {
        const hr_head_field& fs = c.f_color_scale_global;
        const hr_head_field& fh = c.f_color_shift_global;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (lane_ok && k == 0) {               // sample 0's row is this lane's own
                dhk[fs.offset + i] += g[i] * cpre[i] * hr_act_grad(fs.act, head[fs.offset + i]);
                dhk[fh.offset + i] += g[i] * hr_act_grad(fh.act, head[fh.offset + i]);
            }
            g[i] = g[i] * gscale[i];               // everything below sees the gradient of the un-scaled colour
        }
    } else if (a.color_table) {
        const float* e = a.color_table + 12 * cam;
        hr_acc_t* de = a.d_color_table + 12 * cam;
        float gn[3] = {g[0], g[1], g[2]};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (ray_ok && k == 0) HR_ATOMIC_ADD(de + 9 + i, g[i] * hr_act_grad(c.color_table_s_act, e[9 + i]));
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (ray_ok && k == 0) HR_ATOMIC_ADD(de + 3 * i + j, g[i] * cpre[j] * hr_act_grad(c.color_table_t_act, e[3 * i + j]));
                gn[j] += g[i] * tcol[3 * i + j];
            }
        }
        g[0] = gn[0]; g[1] = gn[1]; g[2] = gn[2];
    }
    const float gsum = a.white_bg ? (g[0] + g[1] + g[2]) : 0.0f;
    float dpre[3] = {0.f, 0.f, 0.f};
    if (lane_ok) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float dr = wgt * g[i];
            if (c.f_color_scale.offset >= 0) {
                const float hs = hk[c.f_color_scale.offset + i], hh = hk[c.f_color_shift.offset + i];
                dhk[c.f_color_scale.offset + i] += dr * raw[i] * hr_act_grad(c.f_color_scale.act, hs);
                dhk[c.f_color_shift.offset + i] += dr * hr_act_grad(c.f_color_shift.act, hh);
            }
            const float draw = dr * sc[i];
            if (!app) dpre[i] = 0.0f;
            else if (c.shading == HR_SHADING_SH) dpre[i] = (pre[i] + 0.5f > 0.0f) ? draw : 0.0f;
            else dpre[i] = draw * raw[i] * (1.0f - raw[i]);
        }
    }
    const float dw = lane_ok ? ((g[0] * rr[0] + g[1] * rr[1] + g[2] * rr[2]) - gsum) : 0.0f;
    float suf = dw * wgt;                          // inclusive suffix sum of dw_j * w_j over the ray's later lanes
#pragma unroll
    for (int d = 1; d < ZP; d <<= 1) {
        const float o = __shfl_down(suf, d, 64);
        if (k + d < ZP) suf += o;
    }
    const float S = suf - dw * wgt;                // sum over j > k
    const float dalpha = dw * T - S / inc;
    const float e1 = 1.0f - alpha;
    const float dsigma = dalpha * e1 * (delta * c.distance_scale);
    const float dfeat = valid ? dsigma * hr_density_grad(c, feat) : 0.0f;
    const float ddelta = (lane_ok && k < Z - 1) ? dalpha * e1 * sigma * c.distance_scale : 0.0f;
    float ddelta_prev = __shfl_up(ddelta, 1, 64);
    if (k == 0) ddelta_prev = 0.0f;
    if (lane_ok) {                                 // hand the per-sample upstream gradients to phases B and C
        const int64_t NS = a.n_rays * Z;
        const int64_t s = ray * Z + k;
        a.tape.ds[s] = dist;
        a.tape.src[s] = src;
        a.tape.dfeat[s] = dfeat;
        a.tape.dpre[s] = dpre[0]; a.tape.dpre[NS + s] = dpre[1]; a.tape.dpre[2 * NS + s] = dpre[2];
        a.tape.ddc[s] = ddelta_prev - ddelta;
    }
}

template <int ZP>
static void hr_launch_train_lanes(const hr_config& cfg, const HrTrainArgs& args, hipStream_t stream)
{
    constexpr int RPB = 256 / ZP;
    const unsigned blocks = (unsigned)((args.n_rays + RPB - 1) / RPB);
    const size_t lds = sizeof(float) * RPB * 3 * args.ca_total;
    const int pc = hr_plane_class(args.planes, 0, args.ca_total);
    if (pc == 1 && !cfg.video) hipLaunchKernelGGL((hr_train_lanes_kernel<ZP, 2, 1>), dim3(blocks), dim3(256), lds, stream, args.cfg_dev, args);
    else if (pc == 1) hipLaunchKernelGGL((hr_train_lanes_kernel<ZP, 4, 1>), dim3(blocks), dim3(256), lds, stream, args.cfg_dev, args);
    else if (pc == 2) hipLaunchKernelGGL((hr_train_lanes_kernel<ZP, 4, 2>), dim3(blocks), dim3(256), lds, stream, args.cfg_dev, args);
    else hipLaunchKernelGGL((hr_train_lanes_kernel<ZP, 4, 0>), dim3(blocks), dim3(256), lds, stream, args.cfg_dev, args);
}

// Phase B.  HR_TRAIN_LPS = 16 adjacent lanes per sample, one texel channel each (a plane pair has 8 or 16 channels per
// texel in every shipped model), so that the atomics of one tap are one contiguous run; a workgroup is 16 such groups
// and walks the samples of RPB whole rays (1 ray when it has 16 samples or more).
#define HR_TRAIN_LPS 16
// rows of basis_mat: 3 (RGB) or 27 (SH: 3 colours x 9 basis functions)
__device__ __forceinline__ int hr_train_basis_rows(const hr_config& c) { return c.shading == HR_SHADING_SH ? 27 : 3; }

template <int ZP>
__global__ __launch_bounds__(256, 4) void hr_train_gather_bwd_kernel(const hr_config* __restrict__ cfgp, const HrTrainArgs a)
{
    HR_FX_ENTER(a);
    const hr_config& c = *cfgp;
    constexpr int GROUPS = 256 / HR_TRAIN_LPS;
    constexpr int RPB = (ZP >= GROUPS) ? 1 : GROUPS / ZP;
    extern __shared__ __attribute__((aligned(16))) float lds[];    // [RPB][3 * CA] decode matrix (float) | [RPB][3 * CA] its gradient | basis_mat's gradient, this workgroup's share (hr_acc_t)
    const int CA = a.ca_total, Z = c.z_channels;
    hr_acc_t* dMs = reinterpret_cast<hr_acc_t*>(lds + RPB * 3 * CA);
    hr_acc_t* bacc = dMs + RPB * 3 * CA;
    const int nb = hr_train_basis_rows(c) * a.n_basis_cols;
    for (int e = threadIdx.x; e < nb; e += 256) bacc[e] = HR_ACC_ZERO;
    const int grp = threadIdx.x / HR_TRAIN_LPS, lane = threadIdx.x % HR_TRAIN_LPS;
    // persistent: the workgroup walks ray blocks blockIdx.x, + gridDim.x, ... and adds its basis_mat gradient to the global one once
    for (int64_t ray0 = (int64_t)blockIdx.x * RPB; ray0 < a.n_rays; ray0 += (int64_t)gridDim.x * RPB) {
        __syncthreads();
        for (int e = threadIdx.x; e < RPB * 3 * CA; e += 256) {
            const int r = e / (3 * CA), i = e - r * 3 * CA;
            float v = 0.0f;
            if (ray0 + r < a.n_rays) {
                float sh[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const float* rr = a.rays + (ray0 + r) * c.ray_dim;
                if (c.shading == HR_SHADING_SH) hr_sh_deg2(rr[3], rr[4], rr[5], sh);
                v = hr_train_decode_coef(c, a, sh, i / CA, i % CA);
            }
            lds[e] = v;
            dMs[e] = HR_ACC_ZERO;
        }
        __syncthreads();
        for (int si = grp; si < RPB * Z; si += GROUPS) {
            const int r = si / Z, k = si - r * Z;
            if (ray0 + r >= a.n_rays) continue;
            if (a.tape.taps) hr_sample_train_bwd_taps(c, a, ray0 + r, k, lds + r * 3 * CA, dMs + r * 3 * CA, lane, HR_TRAIN_LPS, nullptr);
            else hr_sample_train_bwd(c, a, ray0 + r, k, lds + r * 3 * CA, dMs + r * 3 * CA, lane, HR_TRAIN_LPS);
        }
        __syncthreads();
        for (int e = threadIdx.x; e < RPB * 3 * CA; e += 256) {
            const int r = e / (3 * CA), i = e - r * 3 * CA;
            if (ray0 + r >= a.n_rays) continue;
            float sh[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const float* rr = a.rays + (ray0 + r) * c.ray_dim;
            if (c.shading == HR_SHADING_SH) hr_sh_deg2(rr[3], rr[4], rr[5], sh);
            hr_train_fold_basis(c, a, sh, i / CA, i % CA, HR_ACC_VALUE(dMs[e]), bacc);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nb; e += 256) {
        const hr_acc_t v = bacc[e];
        if (v != HR_ACC_ZERO) HR_ATOMIC_ADD_ACC(a.d_basis + e, v);
    }
}

#ifndef HR_TRAIN_DET       // the class-specialised, LDS-windowed phase B: float accumulators (the deterministic build takes the generic kernel above)
// rays per trip: four samples per 16-lane group between the barriers (the per-trip staging of the decode matrices, its barriers and
// the fold into basis_mat's gradient are then a quarter; measured 1 / 2 / 4 / 8 samples: DoNeRF sample-stage backward 0.84 / 0.80 /
// 0.78 / 0.77 ms, immersive 1.46 / 1.32 / 1.24 / 1.21, neural_3d 2.56 / 2.32 / 2.19 / 2.14 -- profiles/r03_c_train_experiments.txt)
// ---- phase B of the shipped decompositions ([8, 4, 4]: PC = 1, [8, 0, 0]: PC = 2), taps on the tape.
// The generic path (hr_sample_train_bwd_taps) walks the three plane pairs one after the other with 16 lanes each -- pairs 1 and 2
// have 8 channels, so half the lanes idle twice -- and keeps three run-time plane descriptors live: ~200 uniform values against ~100
// scalar registers, i.e. 263 of its 896 VALU instructions per sample step were v_readlane_b32 restoring spilled scalars, ~190 more
// 64-bit address arithmetic.  Here a lane has two SLOTS with compile-time texel sizes: slot A = channel `lane` of pair 0 (uniform
// descriptor, scalar base + 32-bit offsets), slot B = channel `lane & 7` of pair 1 (lanes 0-7) or pair 2 (lanes 8-15), whose
// descriptor is per-lane data fixed for the whole kernel.  One load round, one atomic round, one lane reduction per sample.
struct HrBwdSlot {
    const float* A;     // plane texels
    const float* B;     // line / time-plane texels
    float* GA;          // their gradients
    float* GB;
    float* acc;         // workgroup-private window of GB in LDS (nullptr: none), texels [lo, lo + n)
    int lo, n;
    int aw, bw;         // plane width; time-plane width (line: unused)
    int ch;             // this lane's channel of the texel
    int mslot;          // appearance slot of the channel in the ray's decode matrix, -1: a density channel
};

template <int TEX, bool KEYED>
__device__ __forceinline__ void hr_bwd_slot(const HrBwdSlot& q, const hr_axis_tap_g& gx, const hr_axis_tap_g& gy, const hr_axis_tap_g& gv,
                                            const hr_axis_tap& at, const float* M, float* dM, int CA, float dfeat, const float* dpre, float* d3)
{
    // taps (hr_train_taps) and texels (hr_train_gather_bwd_channel), same arithmetic in the same order
    const float wa0 = gx.t.w0 * gy.t.w0, wa1 = gx.t.w1 * gy.t.w0, wa2 = gx.t.w0 * gy.t.w1, wa3 = gx.t.w1 * gy.t.w1;
    const unsigned ia0 = (unsigned)((gy.t.i0 * q.aw + gx.t.i0) * TEX + q.ch), ia1 = (unsigned)((gy.t.i0 * q.aw + gx.t.i1) * TEX + q.ch);
    const unsigned ia2 = (unsigned)((gy.t.i1 * q.aw + gx.t.i0) * TEX + q.ch), ia3 = (unsigned)((gy.t.i1 * q.aw + gx.t.i1) * TEX + q.ch);
    int ib[4];
    float wb[4];
    if (KEYED) {
        ib[0] = at.i0 * q.bw + gv.t.i0; ib[1] = at.i0 * q.bw + gv.t.i1; ib[2] = at.i1 * q.bw + gv.t.i0; ib[3] = at.i1 * q.bw + gv.t.i1;
        wb[0] = gv.t.w0 * at.w0; wb[1] = gv.t.w1 * at.w0; wb[2] = gv.t.w0 * at.w1; wb[3] = gv.t.w1 * at.w1;
    } else {
        ib[0] = gv.t.i0; ib[1] = gv.t.i1; ib[2] = 0; ib[3] = 0;
        wb[0] = gv.t.w0; wb[1] = gv.t.w1; wb[2] = 0.0f; wb[3] = 0.0f;
    }
    const float a00 = q.A[ia0], a01 = q.A[ia1], a10 = q.A[ia2], a11 = q.A[ia3];
    const float b0 = q.B[(unsigned)(ib[0] * TEX + q.ch)], b1 = q.B[(unsigned)(ib[1] * TEX + q.ch)];
    float b2 = 0.0f, b3 = 0.0f;
    if (KEYED) { b2 = q.B[(unsigned)(ib[2] * TEX + q.ch)]; b3 = q.B[(unsigned)(ib[3] * TEX + q.ch)]; }
    const float pa = fmaf(a11, wa3, fmaf(a10, wa2, fmaf(a01, wa1, a00 * wa0)));
    const float pb = fmaf(b3, wb[3], fmaf(b2, wb[2], fmaf(b1, wb[1], b0 * wb[0])));
    const float f = pa * pb;
    float u;
    if (q.mslot < 0) {
        u = dfeat;
    } else {
        u = dpre[0] * M[q.mslot] + dpre[1] * M[CA + q.mslot] + dpre[2] * M[2 * CA + q.mslot];
        HR_ATOMIC_ADD_RAY(dM + q.mslot, dpre[0] * f); HR_ATOMIC_ADD_RAY(dM + CA + q.mslot, dpre[1] * f); HR_ATOMIC_ADD_RAY(dM + 2 * CA + q.mslot, dpre[2] * f);
    }
    if (u == 0.0f) return;
    const float dpa = u * pb, dpb = u * pa;
    if (wa0 != 0.0f) HR_ATOMIC_ADD(q.GA + ia0, dpa * wa0);
    if (wa1 != 0.0f) HR_ATOMIC_ADD(q.GA + ia1, dpa * wa1);
    if (wa2 != 0.0f) HR_ATOMIC_ADD(q.GA + ia2, dpa * wa2);
    if (wa3 != 0.0f) HR_ATOMIC_ADD(q.GA + ia3, dpa * wa3);
#pragma unroll
    for (int i = 0; i < (KEYED ? 4 : 2); ++i) {
        if (wb[i] == 0.0f) continue;
        const int rel = ib[i] - q.lo;
        if (q.acc && (unsigned)rel < (unsigned)q.n) HR_ATOMIC_ADD_RAY(q.acc + (unsigned)(rel * TEX + q.ch), dpb * wb[i]);
        else HR_ATOMIC_ADD(q.GB + (unsigned)(ib[i] * TEX + q.ch), dpb * wb[i]);
    }
    d3[0] += dpa * ((a00 * gx.s0 + a01 * gx.s1) * gy.t.w0 + (a10 * gx.s0 + a11 * gx.s1) * gy.t.w1);
    d3[1] += dpa * ((a00 * gx.t.w0 + a01 * gx.t.w1) * gy.s0 + (a10 * gx.t.w0 + a11 * gx.t.w1) * gy.s1);
    if (!KEYED) d3[2] += dpb * (b0 * gv.s0 + b1 * gv.s1);
    else d3[2] += dpb * ((b0 * gv.s0 + b1 * gv.s1) * at.w0 + (b2 * gv.s0 + b3 * gv.s1) * at.w1);
}

// one sample, its 16 lanes together.  `sa` / `sb`: this lane's slots (sb per-lane); `on_a` / `on_b`: the pass differentiates pair 0 / pairs 1 + 2
template <bool KEYED, int PC>
__device__ __forceinline__ void hr_bwd_class_sample(const hr_config& c, const HrTrainArgs& a, int64_t ray, int k, const float* M, float* dM, int lane,
                                                    const HrBwdSlot& sa, const HrBwdSlot& sb, bool on_a, bool on_b, int add_dp, const hr_axis_tap& at)
{
    const int Z = c.z_channels, CA = a.ca_total;
    const int64_t NS = a.n_rays * Z;
    const unsigned s = (unsigned)(ray * Z + k);
    const float dfeat = a.tape.dfeat[s];
    const float dpre[3] = {a.tape.dpre[s], a.tape.dpre[NS + s], a.tape.dpre[2 * NS + s]};
    float dp[3] = {0.f, 0.f, 0.f};
    if (dfeat != 0.0f || dpre[0] != 0.0f || dpre[1] != 0.0f || dpre[2] != 0.0f) {     // only samples that were valid
        hr_axis_tap_g ax[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float* t = a.tape.taps + (int64_t)(6 * i) * NS;
            ax[i].t.i0 = __builtin_bit_cast(int, t[s]);
            ax[i].t.i1 = __builtin_bit_cast(int, t[NS + s]);
            ax[i].t.w0 = t[2 * NS + s]; ax[i].t.w1 = t[3 * NS + s];
            ax[i].s0 = t[4 * NS + s]; ax[i].s1 = t[5 * NS + s];
            ax[i].mult = 0.5f * (float)(c.grid[i] - 1);
        }
        float dpn[3] = {0.f, 0.f, 0.f};
        if (on_a) {                                  // pair 0: plane (x, y), line / time plane along z
            float d3[3] = {0.f, 0.f, 0.f};
            hr_bwd_slot<16, KEYED>(sa, ax[0], ax[1], ax[2], at, M, dM, CA, dfeat, dpre, d3);
            dpn[0] += d3[0] * ax[0].mult; dpn[1] += d3[1] * ax[1].mult; dpn[2] += d3[2] * ax[2].mult;
        }
        if (PC == 1 && on_b) {                       // pair 1 (lanes 0-7): plane (x, z), line along y; pair 2 (lanes 8-15): plane (y, z), line along x
            const bool p1 = lane < 8;
            hr_axis_tap_g gx = p1 ? ax[0] : ax[1], gv = p1 ? ax[1] : ax[0];
            float d3[3] = {0.f, 0.f, 0.f};
            hr_bwd_slot<8, KEYED>(sb, gx, ax[2], gv, at, M, dM, CA, dfeat, dpre, d3);
            const float cx = d3[0] * gx.mult, cz = d3[1] * ax[2].mult, cv = d3[2] * gv.mult;
            dpn[0] += p1 ? cx : cv; dpn[1] += p1 ? cv : cx; dpn[2] += cz;
        }
        for (int i = 0; i < 3; ++i) dp[i] = HR_LANE_SUM(dpn[i], HR_TRAIN_LPS) * c.inv_size[i];
    }
    if (lane != 0) return;
    if (add_dp) { dp[0] += a.tape.dp[s]; dp[1] += a.tape.dp[NS + s]; dp[2] += a.tape.dp[2 * NS + s]; }
    a.tape.dp[s] = dp[0]; a.tape.dp[NS + s] = dp[1]; a.tape.dp[2 * NS + s] = dp[2];
}

#ifndef HR_TRAIN_TRIP_MULT
#define HR_TRAIN_TRIP_MULT 4
#endif
#define HR_TRAIN_LINES_RPB(ZP) (HR_TRAIN_TRIP_MULT * (((1024 / HR_TRAIN_LPS) + (ZP) - 1) / (ZP)))
// Phase B with the contended part of the gradient in LDS.  Static nets (KEYED = false): the three lines, whole -- a few hundred
// texels that every sample of the batch hits.  Keyframe nets (KEYED = true): the two rows of each time plane that the rays of
// one keyframe interval blend between -- the batch's rays are walked in the order of tape.perm (grouped by that row,
// hr_train_bucket_kernel), every workgroup takes a contiguous range of that order and moves its window (adding it to the global
// gradient) when the row changes: 12 keyframes x a few hundred texels take the time-plane adds of all 524 288 samples of a batch,
// as global atomics they were 2-3 ms of the keyframe families' step.  `pairs`: the plane pairs of this pass (HrTrainWindow).
template <int ZP, bool KEYED, int PC>
__global__ __launch_bounds__(1024) void hr_train_gather_bwd_lines_kernel(const hr_config* __restrict__ cfgp, const HrTrainArgs a, const unsigned pairs, const int add_dp)
{
    HR_FX_ENTER(a);
    const hr_config& c = *cfgp;
    constexpr int GROUPS = 1024 / HR_TRAIN_LPS;
    constexpr int RPB = HR_TRAIN_LINES_RPB(ZP);
    extern __shared__ float lds[];                 // [RPB][3 * CA] decode matrix | [RPB][3 * CA] its gradient | the windows | basis_mat's gradient | [RPB][4] time taps (PC != 0)
    const int CA = a.ca_total, Z = c.z_channels;
    HrTrainWindow win;
    win.pairs = pairs;
    win.add_dp = add_dp;
    float* bacc;
    float* s_tt;
    const int nb = hr_train_basis_rows(c) * a.n_basis_cols;
    {
        float* p = lds + 2 * RPB * 3 * CA;
        for (int j = 0; j < 3; ++j) {
            const HrGridPlane& g = a.planes[j];
            const bool on = ((pairs >> j) & 1u) && (g.cd4 + g.ca4) > 0;
            win.acc[j] = on ? p : nullptr;
            win.lo[j] = 0;
            win.n[j] = on ? (KEYED ? 2 * g.bw : g.bh) : 0;
            p += win.n[j] * g.tex;
        }
        bacc = p;
        p += nb;
        for (float* q = lds + 2 * RPB * 3 * CA + threadIdx.x; q < p; q += 1024) *q = 0.0f;
        s_tt = p;                                  // [RPB][4] the rays' time taps (class path)
    }
    // class path: this lane's two slots (see HrBwdSlot)
    const int lane16 = threadIdx.x % HR_TRAIN_LPS;
    HrBwdSlot sa, sb;
    {
        const HrGridPlane& g0 = a.planes[0];
        sa.A = reinterpret_cast<const float*>(g0.a); sa.B = reinterpret_cast<const float*>(g0.b); sa.GA = a.g_a[0]; sa.GB = a.g_b[0];
        sa.acc = win.acc[0]; sa.lo = 0; sa.n = win.n[0]; sa.aw = g0.aw; sa.bw = g0.bw;
        sa.ch = lane16; sa.mslot = lane16 < 8 ? -1 : lane16 - 8;
        const int jb = lane16 < 8 ? 1 : 2;
        const HrGridPlane& g1 = a.planes[1];
        const HrGridPlane& g2 = a.planes[2];
        sb.A = reinterpret_cast<const float*>(jb == 1 ? g1.a : g2.a); sb.B = reinterpret_cast<const float*>(jb == 1 ? g1.b : g2.b);
        sb.GA = jb == 1 ? a.g_a[1] : a.g_a[2]; sb.GB = jb == 1 ? a.g_b[1] : a.g_b[2];
        sb.acc = jb == 1 ? win.acc[1] : win.acc[2]; sb.lo = 0; sb.n = jb == 1 ? win.n[1] : win.n[2];
        sb.aw = jb == 1 ? g1.aw : g2.aw; sb.bw = jb == 1 ? g1.bw : g2.bw;
        sb.ch = lane16 & 7; sb.mslot = (lane16 & 7) < 4 ? -1 : (jb == 1 ? 8 : 12) + (lane16 & 7) - 4;
    }
    const bool on_a = (pairs & 1u) != 0, on_b = (pairs & 6u) == 6u;
    // add the windows to the global gradient and clear them (the caller has a barrier on both sides)
    auto flush = [&]() {
        for (int j = 0; j < 3; ++j) {
            if (!win.acc[j]) continue;
            const int tex = a.planes[j].tex, cnt = win.n[j] * tex;
            float* dst = a.g_b[j] + (int64_t)win.lo[j] * tex;       // a row outside the plane was never added to: stays zero, is never written
            for (int i = threadIdx.x; i < cnt; i += 1024) {
                const float v = win.acc[j][i];
                if (v != 0.0f) { HR_ATOMIC_ADD(dst + i, v); win.acc[j][i] = 0.0f; }
            }
        }
    };
    const int grp = threadIdx.x / HR_TRAIN_LPS, lane = threadIdx.x % HR_TRAIN_LPS;
    const int64_t trips = (a.n_rays + RPB - 1) / RPB;
    // static: trips blockIdx.x, + gridDim.x, ...; keyed: a contiguous range of the grouped order, so that the window moves rarely
    const int64_t t_begin = KEYED ? (trips * blockIdx.x) / gridDim.x : blockIdx.x;
    const int64_t t_end = KEYED ? (trips * (blockIdx.x + 1)) / gridDim.x : trips;
    const int64_t t_step = KEYED ? 1 : gridDim.x;
    int row = -0x7fffffff;
    for (int64_t trip = t_begin; trip < t_end; trip += t_step) {
        const int64_t ray0 = trip * RPB;
        __syncthreads();
        if (KEYED) {
            const int r0 = hr_train_time_row(c, a.rays + (int64_t)a.tape.perm[ray0] * c.ray_dim);
            if (r0 != row) {                       // uniform over the workgroup
                if (row != -0x7fffffff) flush();
                row = r0;
                for (int j = 0; j < 3; ++j) win.lo[j] = r0 * a.planes[j].bw;
                sa.lo = r0 * sa.bw;
                sb.lo = r0 * sb.bw;
            }
        }
        if (PC != 0 && threadIdx.x < RPB) {        // the rays' time taps (hr_train_ray's tap_t), once per trip instead of once per sample
            hr_axis_tap t = {0, 0, 0.0f, 0.0f};
            if (ray0 + threadIdx.x < a.n_rays) {
                const int64_t ray = KEYED ? (int64_t)a.tape.perm[ray0 + threadIdx.x] : ray0 + threadIdx.x;
                float base_t = 0.0f;
                if (c.advect) base_t = hr_base_time(c, a.rays[ray * c.ray_dim + c.ray_dim - 1]);
                t = hr_make_tap(c.video ? hr_normalize_time(c, base_t) : 0.0f, c.video ? c.num_keyframes : 2);
            }
            s_tt[4 * threadIdx.x] = __builtin_bit_cast(float, t.i0); s_tt[4 * threadIdx.x + 1] = __builtin_bit_cast(float, t.i1);
            s_tt[4 * threadIdx.x + 2] = t.w0; s_tt[4 * threadIdx.x + 3] = t.w1;
        }
        for (int e = threadIdx.x; e < RPB * 3 * CA; e += 1024) {
            const int r = e / (3 * CA), i = e - r * 3 * CA;
            float v = 0.0f;
            if (ray0 + r < a.n_rays) {
                float sh[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const float* rr = a.rays + (KEYED ? (int64_t)a.tape.perm[ray0 + r] : ray0 + r) * c.ray_dim;
                if (c.shading == HR_SHADING_SH) hr_sh_deg2(rr[3], rr[4], rr[5], sh);
                v = hr_train_decode_coef(c, a, sh, i / CA, i % CA);
            }
            lds[e] = v;
            lds[RPB * 3 * CA + e] = 0.0f;
        }
        __syncthreads();
        for (int si = grp; si < RPB * Z; si += GROUPS) {
            const int r = si / Z, k = si - r * Z;
            if (ray0 + r >= a.n_rays) continue;
            const int64_t ray = KEYED ? (int64_t)a.tape.perm[ray0 + r] : ray0 + r;
            if (PC != 0) {
                hr_axis_tap at;
                at.i0 = __builtin_bit_cast(int, s_tt[4 * r]); at.i1 = __builtin_bit_cast(int, s_tt[4 * r + 1]);
                at.w0 = s_tt[4 * r + 2]; at.w1 = s_tt[4 * r + 3];
                hr_bwd_class_sample<KEYED, PC>(c, a, ray, k, lds + r * 3 * CA, lds + (RPB + r) * 3 * CA, lane, sa, sb, on_a, on_b, add_dp, at);
            } else if (a.tape.taps) hr_sample_train_bwd_taps(c, a, ray, k, lds + r * 3 * CA, lds + (RPB + r) * 3 * CA, lane, HR_TRAIN_LPS, &win);
            else hr_sample_train_bwd(c, a, ray, k, lds + r * 3 * CA, lds + (RPB + r) * 3 * CA, lane, HR_TRAIN_LPS, &win);
        }
        __syncthreads();
        for (int e = threadIdx.x; e < RPB * 3 * CA; e += 1024) {
            const int r = e / (3 * CA), i = e - r * 3 * CA;
            if (ray0 + r >= a.n_rays) continue;
            float sh[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const float* rr = a.rays + (KEYED ? (int64_t)a.tape.perm[ray0 + r] : ray0 + r) * c.ray_dim;
            if (c.shading == HR_SHADING_SH) hr_sh_deg2(rr[3], rr[4], rr[5], sh);
            hr_train_fold_basis(c, a, sh, i / CA, i % CA, lds[RPB * 3 * CA + e], bacc);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nb; e += 1024) {
        const float v = bacc[e];
        if (v != 0.0f) HR_ATOMIC_ADD(a.d_basis + e, v);
    }
    flush();
}

// Keyframe nets: tape.perm = the batch's ray indices grouped by hr_train_time_row (a counting sort in one workgroup: a batch is
// 16 384 rays; the order inside a group is whatever the LDS counters hand out).  LDS: num_keyframes + 1 counters.
__global__ __launch_bounds__(1024) void hr_train_bucket_kernel(const hr_config* __restrict__ cfgp, const float* __restrict__ rays, int64_t n, int* __restrict__ perm)
{
    const hr_config& c = *cfgp;
    extern __shared__ int cnt[];
    const int nb = (c.video ? c.num_keyframes : 2) + 1;         // rows -1 .. K - 1
    for (int i = threadIdx.x; i < nb; i += 1024) cnt[i] = 0;
    __syncthreads();
    for (int64_t r = threadIdx.x; r < n; r += 1024) atomicAdd(&cnt[hr_train_time_row(c, rays + r * c.ray_dim) + 1], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < nb; ++b) { const int t = cnt[b]; cnt[b] = acc; acc += t; }
    }
    __syncthreads();
    for (int64_t r = threadIdx.x; r < n; r += 1024) perm[atomicAdd(&cnt[hr_train_time_row(c, rays + r * c.ray_dim) + 1], 1)] = (int)r;
}

// LDS bytes of the windows of plane pairs `pairs`; keyed: two rows of each time plane, else the whole lines
static size_t hr_train_window_bytes(const HrTrainArgs& args, unsigned pairs, bool keyed)
{
    size_t n = 0;
    for (int j = 0; j < 3; ++j) {
        const HrGridPlane& g = args.planes[j];
        if (g.cd4 + g.ca4 == 0 || !((pairs >> j) & 1u)) continue;
        n += sizeof(float) * (size_t)(keyed ? 2 * g.bw : g.bh) * g.tex;
    }
    return n;
}

// 1: the [8, 4, 4] decomposition, 2: [8, 0, 0] (what hr_bwd_class_sample is compiled for), 0: anything else
static int hr_train_plane_class(const HrTrainArgs& a, const hr_config& cfg)
{
    const HrGridPlane* pl = a.planes;
    auto ok = [&](int j, int cd4, int off) {
        return pl[j].cd4 == cd4 && pl[j].ca4 == cd4 && pl[j].tex == 8 * cd4 && pl[j].app_off == off &&
               (int64_t)pl[j].aw * pl[j].ah * pl[j].tex < (1ll << 30) && (int64_t)pl[j].bw * pl[j].bh * pl[j].tex < (1ll << 30);
    };
    if (!a.tape.taps || !a.tape.dp || a.n_rays * cfg.z_channels >= (1ll << 30) || !ok(0, 2, 0)) return 0;
    if (ok(1, 1, 8) && ok(2, 1, 12) && a.ca_total == 16) return 1;
    if (pl[1].cd4 + pl[1].ca4 == 0 && pl[2].cd4 + pl[2].ca4 == 0 && a.ca_total == 8) return 2;
    return 0;
}

template <int ZP, bool KEYED>
static void hr_launch_lines_kernel(int pc, unsigned blocks, size_t lds, hipStream_t stream, const HrTrainArgs& args, unsigned pairs, int add_dp)
{
    if (pc == 1) hipLaunchKernelGGL((hr_train_gather_bwd_lines_kernel<ZP, KEYED, 1>), dim3(blocks), dim3(1024), lds, stream, args.cfg_dev, args, pairs, add_dp);
    else if (pc == 2) hipLaunchKernelGGL((hr_train_gather_bwd_lines_kernel<ZP, KEYED, 2>), dim3(blocks), dim3(1024), lds, stream, args.cfg_dev, args, pairs, add_dp);
    else hipLaunchKernelGGL((hr_train_gather_bwd_lines_kernel<ZP, KEYED, 0>), dim3(blocks), dim3(1024), lds, stream, args.cfg_dev, args, pairs, add_dp);
}
template <int ZP, bool KEYED>
static bool hr_lines_opt_in(int pc, size_t lds)
{
    static HrLdsOptIn opt[3];
    const void* fn = pc == 1 ? reinterpret_cast<const void*>(&hr_train_gather_bwd_lines_kernel<ZP, KEYED, 1>)
                   : pc == 2 ? reinterpret_cast<const void*>(&hr_train_gather_bwd_lines_kernel<ZP, KEYED, 2>)
                             : reinterpret_cast<const void*>(&hr_train_gather_bwd_lines_kernel<ZP, KEYED, 0>);
    return hr_lds_opt_in(opt[pc], fn, lds);
}

#endif
template <int ZP>
static bool hr_launch_gather_bwd_lines(const hr_config& cfg, const HrTrainArgs& args, hipStream_t stream)
{
#if defined(HR_TRAIN_NO_WINDOWS) || defined(HR_TRAIN_DET)      // measurement builds / the deterministic build: the global-atomics kernel for everything
    (void)cfg; (void)args; (void)stream;
    return false;
#else
    constexpr int RPB = HR_TRAIN_LINES_RPB(ZP);
    const int pc = hr_train_plane_class(args, cfg);
    bool any = false, keyed = false;
    for (int j = 0; j < 3; ++j)
        if (args.planes[j].cd4 + args.planes[j].ca4 > 0) { any = true; keyed = keyed || args.planes[j].bw != 1; }
    if (!any) return false;
    const size_t base = sizeof(float) * (2 * RPB * 3 * args.ca_total + 27 * args.n_basis_cols + 4 * RPB);
    const size_t cap = 150 * 1024;
    const int64_t iters = (args.n_rays + RPB - 1) / RPB;
    const int cus = hr_current_device_cus();
    const unsigned blocks = (unsigned)(iters < cus ? iters : cus);
    if (!keyed) {
        const size_t lds = base + hr_train_window_bytes(args, 7u, false);
        if (lds > cap) return false;
        if (!hr_lines_opt_in<ZP, false>(pc, lds)) return false;
        hr_launch_lines_kernel<ZP, false>(pc, blocks, lds, stream, args, 7u, 0);
        return true;
    }
    // keyframe net: needs the taps on the tape (two passes re-read them) and the grouped order; all pairs in one pass if their
    // rows fit, else pair 0 (the wide one) and pairs 1 + 2
    if (!args.tape.taps || !args.tape.dp || !args.tape.perm || args.n_rays > 0x7fffffff || !cfg.video || cfg.num_keyframes < 2 || cfg.num_keyframes > 8192) return false;
    unsigned passes[2] = {7u, 0u};
    if (base + hr_train_window_bytes(args, 7u, true) > cap) { passes[0] = 1u; passes[1] = 6u; }
    size_t lds_max = 0;
    for (int p = 0; p < 2; ++p) {
        if (!passes[p]) continue;
        const size_t lds = base + hr_train_window_bytes(args, passes[p], true);
        if (lds > cap) return false;
        lds_max = lds > lds_max ? lds : lds_max;
    }
    if (!hr_lines_opt_in<ZP, true>(pc, lds_max)) return false;
    hipLaunchKernelGGL(hr_train_bucket_kernel, dim3(1), dim3(1024), sizeof(int) * (cfg.num_keyframes + 1), stream, args.cfg_dev, args.rays, args.n_rays, args.tape.perm);
    for (int p = 0; p < 2; ++p) {
        if (!passes[p]) continue;
        const size_t lds = base + hr_train_window_bytes(args, passes[p], true);
        hr_launch_lines_kernel<ZP, true>(pc, blocks, lds, stream, args, passes[p], p);
    }
    return true;
#endif
}

// Tail of phase B (taps path): one thread per sorted sample
__global__ __launch_bounds__(256) void hr_train_point_bwd_kernel(const hr_config* __restrict__ cfgp, const HrTrainArgs a)
{
    HR_FX_ENTER(a);
    const hr_config& c = *cfgp;
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_rays * c.z_channels) return;
    hr_sample_train_point_bwd(c, a, s / c.z_channels, (int)(s % c.z_channels));
}

// Phase C
__global__ __launch_bounds__(256) void hr_train_dist_bwd_kernel(const hr_config* __restrict__ cfgp, const HrTrainArgs a)
{
    HR_FX_ENTER(a);
    const hr_config& c = *cfgp;
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_rays * c.z_channels) return;
    hr_sample_train_dist_bwd(c, a, s / c.z_channels, (int)(s % c.z_channels));
}

#ifdef HR_TRAIN_DET
static void hr_launch_train_impl(const hr_config& cfg, const HrTrainArgs& args_in, hipStream_t stream)
#else
void hr_launch_train(const hr_config& cfg, const HrTrainArgs& args_in, hipStream_t stream)
#endif
{
    if (args_in.n_rays <= 0) return;
    int ZP = 8;
    while (ZP < cfg.z_channels) ZP <<= 1;
    HrTrainArgs args = args_in;
    if (ZP > 64) { args.tape.taps = nullptr; args.tape.dp = nullptr; }     // the one-thread-per-ray phase A leaves no taps
    const unsigned blocks = (unsigned)((args.n_rays + HR_TRAIN_RPW - 1) / HR_TRAIN_RPW);
    switch (ZP) {                                  // phase A: a lane per sample where a ray fits one wavefront
        case 8: hr_launch_train_lanes<8>(cfg, args, stream); break;
        case 16: hr_launch_train_lanes<16>(cfg, args, stream); break;
        case 32: hr_launch_train_lanes<32>(cfg, args, stream); break;
        case 64: hr_launch_train_lanes<64>(cfg, args, stream); break;
        case 128: hipLaunchKernelGGL(hr_train_kernel<128>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 256: hipLaunchKernelGGL(hr_train_kernel<256>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        default: break;
    }
    if (!args.d_rgb) return;
    bool done = false;
    switch (ZP) {
        case 8: done = hr_launch_gather_bwd_lines<8>(cfg, args, stream); break;
        case 16: done = hr_launch_gather_bwd_lines<16>(cfg, args, stream); break;
        case 32: done = hr_launch_gather_bwd_lines<32>(cfg, args, stream); break;
        case 64: done = hr_launch_gather_bwd_lines<64>(cfg, args, stream); break;
        case 128: done = hr_launch_gather_bwd_lines<128>(cfg, args, stream); break;
        case 256: done = hr_launch_gather_bwd_lines<256>(cfg, args, stream); break;
        default: break;
    }
    const int GROUPS = 256 / HR_TRAIN_LPS;
    const int RPB = (ZP >= GROUPS) ? 1 : GROUPS / ZP;
    const int64_t nblocks = (args.n_rays + RPB - 1) / RPB;
    const int64_t resident = 16 * (int64_t)hr_current_device_cus();          // four 256-thread workgroups per CU are resident (128 registers); four rounds of them: the blocks differ in cost
    const unsigned bblocks = (unsigned)(nblocks < resident ? nblocks : resident);
    const size_t lds = sizeof(float) * (RPB * 3 * args.ca_total) + sizeof(hr_acc_t) * (RPB * 3 * args.ca_total + 27 * args.n_basis_cols);
    if (!done) switch (ZP) {
        case 8: hipLaunchKernelGGL(hr_train_gather_bwd_kernel<8>, dim3(bblocks), dim3(256), lds, stream, args.cfg_dev, args); break;
        case 16: hipLaunchKernelGGL(hr_train_gather_bwd_kernel<16>, dim3(bblocks), dim3(256), lds, stream, args.cfg_dev, args); break;
        case 32: hipLaunchKernelGGL(hr_train_gather_bwd_kernel<32>, dim3(bblocks), dim3(256), lds, stream, args.cfg_dev, args); break;
        case 64: hipLaunchKernelGGL(hr_train_gather_bwd_kernel<64>, dim3(bblocks), dim3(256), lds, stream, args.cfg_dev, args); break;
        case 128: hipLaunchKernelGGL(hr_train_gather_bwd_kernel<128>, dim3(bblocks), dim3(256), lds, stream, args.cfg_dev, args); break;
        case 256: hipLaunchKernelGGL(hr_train_gather_bwd_kernel<256>, dim3(bblocks), dim3(256), lds, stream, args.cfg_dev, args); break;
        default: break;
    }
    const int64_t ns = args.n_rays * cfg.z_channels;
    if (args.tape.taps) hipLaunchKernelGGL(hr_train_point_bwd_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream, args.cfg_dev, args);
    hipLaunchKernelGGL(hr_train_dist_bwd_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream, args.cfg_dev, args);
}

#ifdef HR_TRAIN_DET
// the step's fixed-point unit: max |d_rgb| -> 2^32 units (hr_train.h); one workgroup, before the step's first accumulating kernel.  ALWAYS
// run (n = 0 without a d_rgb): the flag of the step before must not leak into this one
__global__ __launch_bounds__(1024) void hr_fx_scale_kernel(const float* __restrict__ d_rgb, int64_t n, HrFxUnit* __restrict__ fx)
{
    __shared__ float s_max[16];
    __shared__ unsigned s_bad;
    if (threadIdx.x == 0) s_bad = 0u;
    __syncthreads();
    float mx = 0.0f;
    bool bad = false;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float v = fabsf(d_rgb[i]);
        if (!(v <= 3.0e38f)) bad = true; else mx = fmaxf(mx, v);
    }
    for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    if (bad) s_bad = 1u;
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) mx = fmaxf(mx, s_max[w]);
        int e = 0;
        if (mx > 0.0f) (void)frexpf(mx, &e);          // mx = f * 2^e, f in [0.5, 1)
        int sh = 32 - e;                                // mx * 2^sh in [2^31, 2^32)
        sh = sh < -60 ? -60 : (sh > 120 ? 120 : sh);
        fx->one = ldexpf(1.0f, sh);
        fx->inv = ldexpf(1.0f, -sh);
        fx->bad = s_bad;
    }
}
}   // namespace hr_det
// the caller's HrTrainArgs is the default build's (float pointers); the structs differ in pointee types only
void hr_launch_train_det(const hr_config& cfg, const void* args_flt, size_t args_bytes, hipStream_t stream)
{
    hr_det::HrTrainArgs a;
    if (args_bytes != sizeof(a) || !reinterpret_cast<const hr_det::HrTrainArgs*>(args_flt)->fx) return;
    memcpy(&a, args_flt, sizeof(a));
    const bool have = a.d_rgb && a.n_rays > 0;
    hipLaunchKernelGGL(hr_det::hr_fx_scale_kernel, dim3(1), dim3(1024), 0, stream, have ? a.d_rgb : nullptr, have ? a.n_rays * 3 : 0, a.fx);
    hr_det::hr_launch_train_impl(cfg, a, stream);
}
#else
// Coarse level of a point_prediction cascade (hr_ray_rows ... in hr_train.h): rows forward per ray, then per sample the
// point backward and the intersection backward.  Elementwise work, no gather.
template <int ZP>
__global__ __launch_bounds__(64) void hr_rows_kernel(const hr_config* __restrict__ cfgp, const HrRowsArgs a)
{
    const int64_t ray = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (ray >= a.n_rays) return;
    hr_ray_rows<ZP>(*cfgp, a, ray);
}

__global__ __launch_bounds__(256) void hr_rows_bwd_kernel(const hr_config* __restrict__ cfgp, const HrRowsArgs a, int phase)
{
    const hr_config& c = *cfgp;
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_rays * c.z_channels) return;
    if (phase == 0) hr_sample_rows_bwd(c, a, s / c.z_channels, (int)(s % c.z_channels));
    else hr_sample_rows_dist_bwd(c, a, s / c.z_channels, (int)(s % c.z_channels));
}

void hr_launch_rows(const hr_config& cfg, const HrRowsArgs& args, hipStream_t stream)
{
    if (args.n_rays <= 0) return;
    int ZP = 8;
    while (ZP < cfg.z_channels) ZP <<= 1;
    const unsigned blocks = (unsigned)((args.n_rays + 63) / 64);
    switch (ZP) {
        case 8: hipLaunchKernelGGL(hr_rows_kernel<8>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 16: hipLaunchKernelGGL(hr_rows_kernel<16>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 32: hipLaunchKernelGGL(hr_rows_kernel<32>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 64: hipLaunchKernelGGL(hr_rows_kernel<64>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 128: hipLaunchKernelGGL(hr_rows_kernel<128>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        case 256: hipLaunchKernelGGL(hr_rows_kernel<256>, dim3(blocks), dim3(64), 0, stream, args.cfg_dev, args); break;
        default: break;
    }
    if (!args.d_rows) return;
    const int64_t ns = args.n_rays * cfg.z_channels;
    const unsigned sb = (unsigned)((ns + 255) / 256);
    hipLaunchKernelGGL(hr_rows_bwd_kernel, dim3(sb), dim3(256), 0, stream, args.cfg_dev, args, 0);
    hipLaunchKernelGGL(hr_rows_bwd_kernel, dim3(sb), dim3(256), 0, stream, args.cfg_dev, args, 1);
}

// rays (n, ray_dim) -> MLP input features (n, mlp_in): ray parameterisation + positional encoding
// (nlf/param.py:87-115,244-253; nlf/pe.py:53-66,210-221), what RayPredictionEmbedding feeds its net (embedding/ray.py:316-330)
__global__ __launch_bounds__(256) void hr_features_kernel(const hr_config* __restrict__ cfgp, const float* __restrict__ rays, int64_t n,
                                                         float* __restrict__ out)
{
    const int64_t ray = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (ray >= n) return;
    const hr_config& c = *cfgp;
    float f[HR_MAX_MLP_IN];
    const int m = hr_ray_features(c, rays + ray * c.ray_dim, f);
    for (int i = 0; i < m; ++i) out[ray * c.mlp_in + i] = f[i];
}

void hr_launch_features(const hr_config* cfg_dev, const float* rays, int64_t n, float* out, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(hr_features_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, cfg_dev, rays, n, out);
}

// Occupancy of the grids (hr_mask.h): one thread per lattice point, z fastest -- neighbouring threads read neighbouring
// line texels and the same plane texel rows.  8 M points (200^3) x 16 density channels x 6 taps: a few hundred microseconds,
// twice per training run (update_AlphaMask_list).
__global__ __launch_bounds__(256) void hr_dense_alpha_kernel(const hr_config* __restrict__ cfgp, const HrMaskArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n12 = (int64_t)a.n[1] * a.n[2];
    if (i >= n12 * a.n[0]) return;
    const int x = (int)(i / n12), y = (int)((i - (int64_t)x * n12) / a.n[2]), z = (int)(i % a.n[2]);
    a.alpha[i] = hr_point_alpha(*cfgp, a, x, y, z);
}

void hr_launch_dense_alpha(const HrMaskArgs& args, hipStream_t stream)
{
    const int64_t n = (int64_t)args.n[0] * args.n[1] * args.n[2];
    if (n <= 0) return;
    hipLaunchKernelGGL(hr_dense_alpha_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, args.cfg_dev, args);
}
#endif   // !HR_TRAIN_DET
