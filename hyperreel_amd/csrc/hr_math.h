// Per-ray / per-sample arithmetic of the HyperReel forward path, written once and
// used by the HIP kernels.  Every function is straight-line fp32 with the reference's
// operation order (file:line cited), so that decisions the reference takes on
// exact comparisons (masks, branch selection) come out the same way.  The file is also
// compilable by a plain host C++ compiler (HR_FN expands to `static inline`), which the
// CPU test-suite uses to check these formulas against the oracle without a GPU
// (tests/host_math).  The product only ever runs them on the device.
#ifndef HR_MATH_H
#define HR_MATH_H

#include <math.h>
#include <stdint.h>

#include "../../include/hyperreel_hip.h"

#if defined(__HIPCC__)
#define HR_FN __device__ __forceinline__
#define HR_UNROLL _Pragma("unroll")      // small fixed-trip loops over float[3]: keep the arrays in registers
#else
#define HR_FN static inline
#define HR_UNROLL
#endif

// Transcendentals and divisions.  The defaults are the compiler's IEEE forms (correctly rounded `/` and sqrtf, libm-grade
// expf / sincosf) so that thresholds the reference takes on exact comparisons (`dist <= near`, `disc <= 0`, the aabb test)
// fall the same way as in its CPU ops.  Three opt-in groups of 1-ulp hardware approximations exist for measurements:
//   HR_FAST_EXP     v_exp_f32 (+ v_rcp_f32 inside sigmoid / tanh): values only, never compared against a threshold
//   HR_FAST_POST    v_rcp_f32 / v_sqrt_f32 / fast tanh+sigmoid AFTER the near/far mask only (contraction of the points,
//                   point offsets, flow, colour scale): continuous quantities, a 1e-7 relative change moves RGB by ~1e-7
//   HR_FAST_DIV     the same approximations in the distance arithmetic too (measurements only, see below)
//   HR_FAST_SINCOS  v_sin_f32 / v_cos_f32 in the windowed positional encoding
// Measured on the 800x800 frames: HR_FAST_DIV flipped a `dist <= near` decision on 1 ray in 20 000 of the cylinder scene
// (RGB error 5e-2 on that ray) for 2 % of the sample kernel's time, so it is off in the shipped build (hyperreel_amd/build.py).
#if defined(HR_FAST_MATH)
#define HR_FAST_EXP 1
#define HR_FAST_DIV 1
#define HR_FAST_POST 1
#define HR_FAST_SINCOS 1
#endif
#if defined(__HIPCC__) && defined(HR_FAST_DIV)
#define HR_DIV(a, b) ((a) * __builtin_amdgcn_rcpf(b))
#define HR_SQRT(x) __builtin_amdgcn_sqrtf(x)
#else
#define HR_DIV(a, b) ((a) / (b))
#define HR_SQRT(x) sqrtf(x)
#endif
#if defined(__HIPCC__) && (defined(HR_FAST_POST) || defined(HR_FAST_DIV))
#define HR_DIVP(a, b) ((a) * __builtin_amdgcn_rcpf(b))
#define HR_SQRTP(x) __builtin_amdgcn_sqrtf(x)
#else
#define HR_DIVP(a, b) ((a) / (b))
#define HR_SQRTP(x) sqrtf(x)
#endif
#if defined(__HIPCC__) && defined(HR_FAST_EXP)
#define HR_EXP(x) __expf(x)
#define HR_RCP(x) __builtin_amdgcn_rcpf(x)
HR_FN float hr_tanh(float x)
{
    const float e = __expf(2.0f * fminf(fmaxf(x, -15.0f), 15.0f));
    return (e - 1.0f) * __builtin_amdgcn_rcpf(e + 1.0f);
}
#else
#define HR_EXP(x) expf(x)
#define HR_RCP(x) (1.0f / (x))
HR_FN float hr_tanh(float x) { return tanhf(x); }
#endif
#if defined(__HIPCC__) && defined(HR_FAST_SINCOS)
#define HR_SINCOS(x, s, c) do { *(s) = __sinf(x); *(c) = __cosf(x); } while (0)   // v_sin/v_cos, |err| ~ 1e-6
#else
#define HR_SINCOS(x, s, c) sincosf((x), (s), (c))
#endif

// reciprocal for MARGINS only (1 ulp is nothing against their factor 4)
#if defined(__HIPCC__)
#define HR_RCP_BAND(x) __builtin_amdgcn_rcpf(x)
#else
#define HR_RCP_BAND(x) (1.0f / (x))
#endif

// ---------------------------------------------------------------- activations
// y = act(x*inner + shift)*outer   (nlf/activations.py:53-69,121-137,163-178)
HR_FN float hr_apply_act(const hr_act& a, float x)
{
    float y = x * a.inner + a.shift;
    // IEEE forms on purpose: head activations feed the sample distances, which the path compares against thresholds
    if (a.type == HR_ACT_SIGMOID) {
        y = 1.0f / (1.0f + expf(-y));
    } else if (a.type == HR_ACT_TANH) {
        y = tanhf(y);
    }
    // add: the (1 - w) * start_value term of an EaseValue inside its window, else 0 (one fma either way; with add == 0 it
    // rounds exactly like the plain product)
    return fmaf(y, a.outer, a.add);
}

// The same activation for heads that only feed continuous quantities (point offsets, flow, colour scale / shift)
HR_FN float hr_apply_act_post(const hr_act& a, float x)
{
#if defined(__HIPCC__) && defined(HR_FAST_POST)
    // the shipped colour scale / shift heads and the outer offset / flow activations are the identity with unit parameters: x itself (three
    // instructions per value otherwise; the parameters are uniform, their bit patterns compare on the scalar unit).  Only -0 -> +0 is lost.
    if (a.type == HR_ACT_IDENTITY && __builtin_bit_cast(unsigned, a.inner) == 0x3f800000u && __builtin_bit_cast(unsigned, a.shift) == 0u &&
        __builtin_bit_cast(unsigned, a.outer) == 0x3f800000u && __builtin_bit_cast(unsigned, a.add) == 0u) return x;
    float y = __builtin_fmaf(x, a.inner, a.shift);          // (one rounding instead of two: these heads feed continuous quantities only)
    if (a.type == HR_ACT_SIGMOID) {
        y = __builtin_amdgcn_rcpf(1.0f + __expf(-y));
    } else if (a.type == HR_ACT_TANH) {
        const float e = __expf(2.0f * fminf(fmaxf(y, -15.0f), 15.0f));
        y = (e - 1.0f) * __builtin_amdgcn_rcpf(e + 1.0f);
    }
    return __builtin_fmaf(y, a.outer, a.add);
#else
    return hr_apply_act(a, x);
#endif
}

// ---------------------------------------------------------------- ray features (MLP input)
// utils/intersect_utils.py:127-150  (|d| < 1e-5 -> 1e12)
HR_FN float hr_axis_plane_t(float val, float o, float d)
{
    float dd = (fabsf(d) < 1e-5f) ? 1e12f : d;
    return HR_DIV(val - o, dd);
}

// |d hr_axis_plane_t / d val| -- margins only (HrRisk)
HR_FN float hr_axis_plane_amp(float d)
{
    return (fabsf(d) < 1e-5f) ? 1e-12f : HR_RCP_BAND(fabsf(d));
}

// nlf/param.py:244-253 (pluecker), :87-115 (two_plane), :20-24 (identity) followed by
// nlf/pe.py:210-221 (windowed) / :53-66 (basic).  Writes mlp_in floats to `out`
// (stride 1).  Returns the number written.
// part / nparts: the work can be shared by `nparts` callers on the same ray (the MLP kernels' 4 wavefronts): every caller
// evaluates the cheap parameterisation, caller `part` writes the identity columns if part == 0 and the sin/cos columns
// whose running index is congruent to `part` -- the libm-grade sincos chains are what the prologue's time goes into.
HR_FN int hr_ray_features(const hr_config& c, const float* ray, float* out, int part = 0, int nparts = 1)
{
    int n_out = 0;
    int pe_idx = 0;
    for (int g = 0; g < c.n_groups; ++g) {
        const hr_param_group& pg = c.groups[g];
        // up to 8 parameterised values, kept in registers (all loops over them are unrolled)
        float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f, x4 = 0.f, x5 = 0.f, x6 = 0.f, x7 = 0.f;
        int nx = 0;
        if (pg.fn == HR_PARAM_PLUECKER) {
            float ox = ray[pg.start + 0] - pg.origin[0];
            float oy = ray[pg.start + 1] - pg.origin[1];
            float oz = ray[pg.start + 2] - pg.origin[2];
            float dx = ray[pg.start + 3], dy = ray[pg.start + 4], dz = ray[pg.start + 5];
            float nrm = HR_SQRT(dx * dx + dy * dy + dz * dz);    // F.normalize(p=2, eps=1e-12)
            nrm = fmaxf(nrm, 1e-12f);
            dx = HR_DIV(dx, nrm); dy = HR_DIV(dy, nrm); dz = HR_DIV(dz, nrm);
            float mx = oy * dz - oz * dy;                          // torch.cross(o, d)
            float my = oz * dx - ox * dz;
            float mz = ox * dy - oy * dx;
            x0 = dx * pg.a; x1 = dy * pg.a; x2 = dz * pg.a;
            x3 = mx * pg.b; x4 = my * pg.b; x5 = mz * pg.b;
            nx = 6;
        } else if (pg.fn == HR_PARAM_TWO_PLANE) {
            float ox = ray[pg.start + 0] - pg.origin[0];
            float oy = ray[pg.start + 1] - pg.origin[1];
            float oz = ray[pg.start + 2] - pg.origin[2];
            float dx = ray[pg.start + 3], dy = ray[pg.start + 4], dz = ray[pg.start + 5];
            float t1 = hr_axis_plane_t(pg.a, oz, dz);
            float t2 = hr_axis_plane_t(pg.b, oz, dz);
            x0 = ox + dx * t1; x1 = oy + dy * t1;
            x2 = ox + dx * t2; x3 = oy + dy * t2;
            nx = 4;
        } else {
            nx = pg.end - pg.start;
            if (nx > 0) x0 = ray[pg.start + 0];
            if (nx > 1) x1 = ray[pg.start + 1];
            if (nx > 2) x2 = ray[pg.start + 2];
            if (nx > 3) x3 = ray[pg.start + 3];
            if (nx > 4) x4 = ray[pg.start + 4];
            if (nx > 5) x5 = ray[pg.start + 5];
            if (nx > 6) x6 = ray[pg.start + 6];
            if (nx > 7) x7 = ray[pg.start + 7];
        }
#define HR_X(i) ((i) == 0 ? x0 : (i) == 1 ? x1 : (i) == 2 ? x2 : (i) == 3 ? x3 : (i) == 4 ? x4 : (i) == 5 ? x5 : (i) == 6 ? x6 : x7)
        const bool ident = (pg.pe_type == HR_PE_NONE) || (pg.pe_type == HR_PE_BASIC) || !pg.pe_exclude_identity;
        if (ident) {
            if (part == 0)
                for (int i = 0; i < nx; ++i) out[n_out + i] = HR_X(i);
            n_out += nx;
        }
        if (pg.pe_type == HR_PE_WINDOWED) {
            float f = 1.0f;
            for (int j = 0; j < pg.pe_n_freqs; ++j) {
                f = f * pg.pe_freq_mult;                           // freq_multiplier ** (j+1)
                float bf = pg.pe_base_mult * f;
                const float w = pg.pe_weight[j];                   // WindowedPE.weight(j), pe.py:186-208 (1 after the window)
                for (int i = 0; i < nx; ++i, ++pe_idx) {          // [sin(all i), cos(all i)] per frequency
                    if (pe_idx % nparts != part) continue;
                    float sv, cv;
                    HR_SINCOS(bf * HR_X(i), &sv, &cv);
                    out[n_out + i] = w * sv;
                    out[n_out + nx + i] = w * cv;
                }
                n_out += 2 * nx;
            }
        } else if (pg.pe_type == HR_PE_BASIC) {  // [x, sin(f_j x_i) (i-major, j-minor), cos(...)]
            for (int i = 0; i < nx; ++i) {
                float f = 1.0f;
                for (int j = 0; j < pg.pe_n_freqs; ++j, ++pe_idx) { f = f * pg.pe_freq_mult; if (pe_idx % nparts == part) out[n_out] = sinf(f * HR_X(i)); ++n_out; }
            }
            for (int i = 0; i < nx; ++i) {
                float f = 1.0f;
                for (int j = 0; j < pg.pe_n_freqs; ++j, ++pe_idx) { f = f * pg.pe_freq_mult; if (pe_idx % nparts == part) out[n_out] = cosf(f * HR_X(i)); ++n_out; }
            }
        }
#undef HR_X
    }
    return n_out;
}

// ---------------------------------------------------------------- contraction (nlf/contract.py)
// torch.pow(x, scalar) as ATen evaluates it on float32 (pow_tensor_scalar_optimized_kernel): 0.5 -> sqrt, 2 -> x * x, 3 -> x * x * x, else powf
// (its negative special cases -0.5 / -1 / -2 cannot occur: validate() requires positive powers)
HR_FN float hr_pow_scalar(float x, float e)
{
    if (e == 0.5f) return HR_SQRT(x);
    if (e == 2.0f) return x * x;
    if (e == 3.0f) return (x * x) * x;
    return powf(x, e);
}

// inverse_contract_distance: MIPNeRFContract contract.py:143-158 (identity distance_activation);
// BBoxContract :78-79 / ZDepthContract :104-105 (d * fac); DoNeRFContract :226-230
HR_FN float hr_inverse_contract_distance(const hr_config& c, float distance)
{
    if (c.contract_type == HR_CONTRACT_AFFINE) return distance * c.c_aff_fac;
    distance = (distance * 0.5f) * 2.0f;             // x/2*2, exact either way
    distance = fminf(fmaxf(distance, -2.0f), 2.0f);
    if (c.contract_type == HR_CONTRACT_DONERF) {
        const float sgn0 = (distance > 0.0f) ? 1.0f : ((distance < 0.0f) ? -1.0f : 0.0f);
        return HR_DIV(hr_pow_scalar(fabsf(distance) + 1e-8f, c.c_pow_power) * sgn0, c.c_pow_fac);
    }
    float t = 2.0f - fabsf(distance);
    float inv = HR_DIV(t, c.c_d_scale) + c.c_d_inv_end;
    float sgn = (distance > 0.0f) ? 1.0f : ((distance < 0.0f) ? -1.0f : 0.0f);
    float r = (fabsf(distance) < 1.0f) ? distance : sgn * HR_DIV(1.0f, inv);
    return r * c.c_d0;
}

// contract_points: MIPNeRFContract contract.py:178-192; BBoxContract :84-85 / ZDepthContract :110-111
HR_FN void hr_contract_point(const hr_config& c, float px, float py, float pz, float* q)
{
    if (c.contract_type == HR_CONTRACT_AFFINE) {
        q[0] = HR_DIVP(px - c.c_aff_min[0], c.c_aff_size[0]);
        q[1] = HR_DIVP(py - c.c_aff_min[1], c.c_aff_size[1]);
        q[2] = HR_DIVP(pz - c.c_aff_min[2], c.c_aff_size[2]);
        return;
    }
    if (c.contract_type == HR_CONTRACT_DONERF) {     // contract.py:238-240: (p / |p|) * (|p| fac + 1e-8)^(1/power); 0/0 at the origin, as there
        const float dn = HR_SQRTP(px * px + py * py + pz * pz);
        const float s = hr_pow_scalar(dn * c.c_pow_fac + 1e-8f, c.c_pow_inv_power);
        q[0] = HR_DIVP(px, dn) * s; q[1] = HR_DIVP(py, dn) * s; q[2] = HR_DIVP(pz, dn) * s;
        return;
    }
    px = HR_DIVP(px, c.c_r0); py = HR_DIVP(py, c.c_r0); pz = HR_DIVP(pz, c.c_r0);
    float dist = HR_SQRTP(px * px + py * py + pz * pz);
    if (dist < 1.0f) {
        q[0] = px; q[1] = py; q[2] = pz;
    } else {
        float inv = HR_DIVP(1.0f, fabsf(dist));
        float t = (inv - c.c_r_inv_end) * c.c_r_scale;
        float s = 2.0f - t;
        q[0] = HR_DIVP(px, dist) * s; q[1] = HR_DIVP(py, dist) * s; q[2] = HR_DIVP(pz, dist) * s;
    }
}

// ---------------------------------------------------------------- decisions at risk (the verified fast path)
// The per-sample stage is continuous in the MLP's head EXCEPT at a handful of comparisons (near / far mask, the quadratic's discriminant and
// root choice, the bounding box, a positive weight threshold).  A cheaper MLP arithmetic moves the head a little and every continuous
// quantity with it -- invisible at the 1e-4 bar -- but a comparison whose two sides are closer than that error may fall the other way and
// change a pixel by 1e-2.  With a non-NULL HrRisk the functions below also report whether any comparison they made was inside its margin;
// the sample kernel collects the rays that have such a sample and hr_render renders them again with the reference-grade arithmetic.
//
// The margins are derived per SAMPLE from one number the model is calibrated for (api.hip: calibrate_band; band_kernel.hip):
//   band_zc  how far the two arithmetics' values of  z * scale + anchor  (process_z_vals before the inverse contraction: a smooth, bounded-
//            slope function of one head column) may differ -- 4 x the largest difference measured on the calibration rays;
// pushed through the derivatives of what follows it:
//   dlen = |d length / d zc|      of the inverse contraction (1 for none; c_aff_fac; c_d0 r^2 / c_d_scale beyond the unit ball),
//   amp  = |d distance / d length| of the intersection (1 / |d_axis| for a plane; 2 |r| / sqrt(discriminant) for sphere and cylinder),
// so that a length is at risk within band_zc dlen and a distance within band_zc dlen amp.  Points are compared (bounding box) after the
// point contraction, which undoes dlen: their margin is band_q x the largest amp among the ray's samples (+ band_off for the offset and
// flow heads), both measured the same way.  The measurement is taken over rays whose live samples all have amp <= amp_cut (a ray grazing a
// plane or tangent to a sphere conditions everything badly, the MLP's input features included, under any arithmetic); a ray with a live
// sample beyond amp_cut is therefore at risk by that alone.  NULL (the default): nothing is computed.
struct HrRisk {
    float band_zc;
    float band_q;
    float band_off;
    float amp_cut;  // a live sample with amp beyond this is at risk by itself: the margins were measured on rays without such samples (0: no such rule)
    bool hit;
    // out: this sample's zc, dlen and amp (0 where the intersection misses) -- what the margins were built from, and what the band probe reads
    float zc, dlen, amp;
#ifdef HR_DEBUG_HSUM
    float dbg[8];   // measurement builds: intermediates of the sphere intersection (tools/hsum_bisect.py)
#endif
};
#define HR_RISK_INIT(zc_, q_, off_, cut_) HrRisk{(zc_), (q_), (off_), (cut_), false, 0.0f, 1.0f, 0.0f}
#define HR_RISK_ABS(risk, x, y) do { if (risk) (risk)->hit = (risk)->hit || (fabsf((x) - (y)) <= (risk)->band_zc * (risk)->dlen); } while (0)

// ---------------------------------------------------------------- ray / primitive intersection
// utils/intersect_utils.py:45-84 (sphere) and :86-125 (cylinder: the xz components)
HR_FN float hr_quadratic_t(float oo, float dd, float od, float radius, HrRisk* risk = nullptr)
{
    float a = dd;
    float b = 2.0f * od;
    float cc = oo - radius * radius;
    float disc = b * b - 4.0f * a * cc;
    // at risk: the discriminant's sign -- only the radius depends on the head, d disc = 8 a r d r -- and the radius' sign
    float band_r = 0.0f;
    if (risk) {
        band_r = risk->band_zc * risk->dlen;
        risk->hit = risk->hit || (fabsf(disc) <= 8.0f * a * fabsf(radius) * band_r) || (fabsf(radius) <= band_r);
    }
#if defined(HR_DEBUG_HSUM) && HR_DEBUG_HSUM != 2
    if (risk) risk->dbg[4] = disc;
#endif
    disc = (disc < 0.0f) ? 0.0f : disc;
    float sq = HR_SQRT(disc + 1e-8f);
    float t1 = HR_DIV(-b + sq, 2.0f * a);
    float t2 = HR_DIV(-b - sq, 2.0f * a);
#if defined(HR_DEBUG_HSUM) && HR_DEBUG_HSUM != 2
    if (risk) { risk->dbg[5] = sq; risk->dbg[6] = t1; risk->dbg[7] = t2; }
#endif
    if (risk) {
        // d t / d r = +- 2 r / sq for both roots (a cancels); ... and the sign of the near root (which root is returned)
        risk->amp = (disc <= 0.0f) ? 0.0f : 2.0f * fabsf(radius) * HR_RCP_BAND(sq);
        risk->hit = risk->hit || (disc > 0.0f && fabsf(t2) <= band_r * risk->amp);
    }
    t1 = (disc <= 0.0f) ? 0.0f : t1;
    t2 = (disc <= 0.0f) ? 0.0f : t2;
    return ((t2 < 0.0f) || (radius < 0.0f)) ? t1 : t2;
}

HR_FN float hr_sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

// F.normalize(p=2, eps=1e-12) of a 3-vector, in place
HR_FN void hr_normalize3(float* v)
{
    float n = fmaxf(HR_SQRT(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
    v[0] = HR_DIV(v[0], n); v[1] = HR_DIV(v[1], n); v[2] = HR_DIV(v[2], n);
}

HR_FN void hr_cross3(const float* a, const float* b, float* r)
{
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}

// pluecker_pos (nlf/param.py:297-307): the point of the line (o, d) closest to the origin,
// d_hat x (o x d_hat) with d_hat = normalize(d)
HR_FN void hr_pluecker_pos(const float* o, const float* d, float* pos)
{
    float dn[3] = {d[0], d[1], d[2]};
    hr_normalize3(dn);
    float m[3];
    hr_cross3(o, dn, m);
    hr_cross3(dn, m, pos);
}

// sign(d . diff) * |diff|  (primitive.py:171-173, :532-534)
HR_FN float hr_signed_base_distance(const float* d, const float* diff)
{
    float dt = d[0] * diff[0] + d[1] * diff[1] + d[2] * diff[2];
    return hr_sign(dt) * HR_SQRT(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
}

// z_vals channel `ch` of a sample: head activation -> intersect activation * (1 - sigma)  (base.py:161-162)
HR_FN float hr_zval(const hr_config& c, const float* hk, int ch, float one_m)
{
    return hr_apply_act(c.z_act, hr_apply_act(c.f_z_vals.act, hk[c.f_z_vals.offset + ch])) * one_m;
}

// |d hr_inverse_contract_distance / d distance| at `zc` (r: the value it returned) -- margins only
HR_FN float hr_inverse_contract_slope(const hr_config& c, float zc, float r)
{
    if (c.contract_type == HR_CONTRACT_AFFINE) return fabsf(c.c_aff_fac);
    if (c.contract_type == HR_CONTRACT_DONERF) return 1.0f;       // (not derived: such models are not verified, api.hip can_verify)
    if (fabsf(zc) < 1.0f) return fabsf(c.c_d0);
    const float ru = r * HR_RCP_BAND(c.c_d0);                      // 1 / inv
    return fabsf(c.c_d0 * ru * ru * HR_RCP_BAND(c.c_d_scale));
}

// process_z_vals (base.py:128-140): anchor + scale, then back from the contracted sample space
HR_FN float hr_process_z(const hr_config& c, float z, float scale, float anchor, HrRisk* risk = nullptr)
{
    z = z * scale + anchor;
    const float zc = z;
    if (c.contract_samples) z = hr_inverse_contract_distance(c, z);
    if (risk) {
        risk->zc = zc;
        risk->dlen = c.contract_samples ? hr_inverse_contract_slope(c, zc, z) : 1.0f;
    }
    return z;
}

// IntersectSphereNew / IntersectCylinderNew .intersect (primitive.py:498-545, :313-363): the ray is
// moved into the primitive's frame, intersected, and samples whose primitive the ray misses are
// recycled as offsets from the ray's closest point to the axis/centre.
HR_FN float hr_isect_new(const hr_config& c, const float* hk, int k, float one_m, const float* ro, const float* rd, HrRisk* risk = nullptr)
{
    float org[3] = {0.0f, 0.0f, 0.0f};
    if (c.origin_scale != 0.0f)
        HR_UNROLL
        for (int i = 0; i < 3; ++i) org[i] = hr_zval(c, hk, i, one_m) * c.origin_scale;
    float rs[3] = {c.resize_initial[0], c.resize_initial[1], c.resize_initial[2]};
    if (c.resize_scale != 0.0f)
        HR_UNROLL
        for (int i = 0; i < 3; ++i) rs[i] = hr_zval(c, hk, 3 + i, one_m) * c.resize_scale + c.resize_initial[i];
    const float raw = hr_process_z(c, hr_zval(c, hk, 6, one_m), c.z_scale, c.samples[k]);
    const float radius = hr_process_z(c, hr_zval(c, hk, 7, one_m), c.z_scale, c.samples[k], risk);
    float o[3], d[3];
    HR_UNROLL
    for (int i = 0; i < 3; ++i) { o[i] = (ro[i] - org[i]) * rs[i]; d[i] = rd[i] * rs[i]; }
    const float dnorm = HR_SQRT(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);     // torch.norm(rays_d)
    float dn[3] = {d[0], d[1], d[2]};
    hr_normalize3(dn);
    float t, min_radius, base_distance;
    if (c.isect_type == HR_ISECT_SPHERE_NEW) {
        t = hr_quadratic_t(o[0] * o[0] + o[1] * o[1] + o[2] * o[2], dn[0] * dn[0] + dn[1] * dn[1] + dn[2] * dn[2],
                           o[0] * dn[0] + o[1] * dn[1] + o[2] * dn[2], radius, risk);
        float pos[3];
        hr_pluecker_pos(o, dn, pos);                                           // also min_sphere_radius' vector
        min_radius = HR_SQRT(pos[0] * pos[0] + pos[1] * pos[1] + pos[2] * pos[2]);
        const float diff[3] = {pos[0] - o[0], pos[1] - o[1], pos[2] - o[2]};
        base_distance = hr_signed_base_distance(dn, diff);
    } else {
        t = hr_quadratic_t(o[0] * o[0] + o[2] * o[2], dn[0] * dn[0] + dn[2] * dn[2], o[0] * dn[0] + o[2] * dn[2], radius, risk);
        const float oc[3] = {o[0], 0.0f, o[2]}, dc[3] = {dn[0], 0.0f, dn[2]};
        float pos[3];
        hr_pluecker_pos(oc, dc, pos);
        min_radius = HR_SQRT(pos[0] * pos[0] + pos[1] * pos[1] + pos[2] * pos[2]);
        const float diff[3] = {pos[0] - oc[0], pos[1] - oc[1], pos[2] - oc[2]};
        base_distance = HR_DIV(hr_signed_base_distance(dc, diff), HR_SQRT(dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2]));
    }
    HR_RISK_ABS(risk, fabsf(radius), min_radius + 4.0f * c.z_scale);
    if (fabsf(radius) < min_radius + 4.0f * c.z_scale) t = raw + base_distance;
    return HR_DIV(t, dnorm + 1e-5f);
}

// Pre-sort distance of sample k (Intersect.forward, intersect/base.py:142-203):
// head activation -> z activation * (1 - sigma) -> anchors/scale -> inverse contraction
// -> closed-form intersection -> near/far mask.  `hk` points at the P raw head values of
// sample k; `ro`/`rd` are the ray origin (minus intersect origin) and direction.
// The ray's terms of the sphere / cylinder quadratic (primitive.py:425-431, intersect_utils.py:45-125) for a primitive of centre-scale
// (sx, sy, sz): q = {o.o, d.d, o.d} of the scaled origin and direction.  With the shipped origin_scale_factor of 0 the scale is a constant of
// the model and q a constant of the RAY: the sample kernel computes it once per ray (sample_core.inc, hr_ray_constants) and hands it to
// hr_sample_distance, which otherwise calls this per sample -- the same operations in the same order either way.
HR_FN void hr_quadratic_ray_terms(const hr_config& c, const float* ro, const float* rd, float sx, float sy, float sz, float* q)
{
    float ox = ro[0] * sx, oy = ro[1] * sy, oz = ro[2] * sz;     // primitive.py:425-431
    float dx = rd[0] * sx, dy = rd[1] * sy, dz = rd[2] * sz;
    if (c.isect_type == HR_ISECT_SPHERE) {
        q[0] = ox * ox + oy * oy + oz * oz;
        q[1] = dx * dx + dy * dy + dz * dz;
        q[2] = ox * dx + oy * dy + oz * dz;
    } else {
        q[0] = ox * ox + oz * oz;
        q[1] = dx * dx + dz * dz;
        q[2] = ox * dx + oz * dz;
    }
}

// quad: hr_quadratic_ray_terms of this ray for the model's constant centre-scale, or NULL (read only where origin_scale == 0)
HR_FN float hr_sample_distance(const hr_config& c, const float* hk, int k, const float* ro, const float* rd, HrRisk* risk = nullptr,
                               const float* quad = nullptr)
{
    float sigma = 0.0f;
    if (c.f_isect_sigma.offset >= 0) sigma = hr_apply_act(c.f_isect_sigma.act, hk[c.f_isect_sigma.offset]);
    float one_m = 1.0f - sigma;
    float dist;
    if (risk) risk->amp = 1.0f;
    if (c.isect_type == HR_ISECT_Z_PLANE) {
        float z = hr_process_z(c, hr_zval(c, hk, 0, one_m), c.z_scale, c.samples[k], risk);   // base.py:129
        dist = hr_axis_plane_t(z, ro[2], rd[2]);                     // z.py:88-95
        if (risk) risk->amp = hr_axis_plane_amp(rd[2]);
    } else if (c.isect_type == HR_ISECT_SPHERE || c.isect_type == HR_ISECT_CYLINDER) {
        // origins = z[:3] * origin_scale_factor + origin_initial (primitive.py:410-412).  With the
        // shipped origin_scale_factor of 0 the three channels are multiplied by zero; they are then
        // not read at all (and hr_model_finalize drops those columns from the last Linear).
        float sx = c.origin_initial[0], sy = c.origin_initial[1], sz = c.origin_initial[2];
        if (c.origin_scale != 0.0f) {
            sx = hr_zval(c, hk, 0, one_m) * c.origin_scale + c.origin_initial[0];
            sy = hr_zval(c, hk, 1, one_m) * c.origin_scale + c.origin_initial[1];
            sz = hr_zval(c, hk, 2, one_m) * c.origin_scale + c.origin_initial[2];
        }
        float radius = hr_process_z(c, hr_zval(c, hk, 3, one_m), c.z_scale, c.samples[k], risk);
#if defined(HR_DEBUG_HSUM) && HR_DEBUG_HSUM != 2
        if (risk) { risk->dbg[0] = hr_zval(c, hk, 3, one_m); risk->dbg[1] = radius; }
#endif
        float q_[3];
        if (quad && c.origin_scale == 0.0f) { q_[0] = quad[0]; q_[1] = quad[1]; q_[2] = quad[2]; }
        else hr_quadratic_ray_terms(c, ro, rd, sx, sy, sz, q_);
#if defined(HR_DEBUG_HSUM) && HR_DEBUG_HSUM != 2
        if (risk) { risk->dbg[2] = q_[0]; risk->dbg[3] = q_[2]; }
#elif defined(HR_DEBUG_HSUM)
        if (risk) { risk->dbg[6] = q_[0]; risk->dbg[7] = q_[2]; }
#endif
        dist = hr_quadratic_t(q_[0], q_[1], q_[2], radius, risk);
    } else if (c.isect_type == HR_ISECT_SPHERE_NEW || c.isect_type == HR_ISECT_CYLINDER_NEW) {
        dist = hr_isect_new(c, hk, k, one_m, ro, rd, risk);
    } else if (c.isect_type == HR_ISECT_VOXEL_GRID) {
        // voxel.py:72-112: samples are (Z/3, 3) axis planes; sample k is a plane orthogonal to axis k % 3
        const int axis = k % 3;
        float z = hr_process_z(c, hr_zval(c, hk, 0, one_m), c.voxel_scale[axis], c.samples[k], risk);
        const float o = (axis == 0) ? ro[0] : (axis == 1) ? ro[1] : ro[2];
        const float d = (axis == 0) ? rd[0] : (axis == 1) ? rd[1] : rd[2];
        if (c.isect_outward) z = z * hr_sign(d);
        dist = hr_axis_plane_t(z, o, d);                             // intersect_utils.py:152-179
        if (risk) risk->amp = hr_axis_plane_amp(d);
    } else if (c.isect_type == HR_ISECT_DEFORMABLE_VOXEL_GRID) {
        // voxel.py:184-213: a plane per sample, normal = normalize(z[:3]*scale + start_normal[k % axes]),
        // offset = processed z[3]; intersect_plane (intersect_utils.py:210-236)
        const int axis = k % c.dvg_axes;
        float n[3];
        HR_UNROLL
        for (int i = 0; i < 3; ++i) n[i] = c.dvg_normals[3 * axis + i];
        if (c.dvg_normal_scale != 0.0f)
            HR_UNROLL
            for (int i = 0; i < 3; ++i) n[i] = hr_zval(c, hk, i, one_m) * c.dvg_normal_scale + c.dvg_normals[3 * axis + i];
        hr_normalize3(n);
        const float dplane = hr_process_z(c, hr_zval(c, hk, 3, one_m), c.z_scale, c.samples[k], risk);
        const float o_n = (ro[0] * n[0] + ro[1] * n[1]) + ro[2] * n[2];
        float d_n = (rd[0] * n[0] + rd[1] * n[1]) + rd[2] * n[2];
        HR_RISK_ABS(risk, fabsf(d_n), 1e-5f);
        d_n = (fabsf(d_n) < 1e-5f) ? 1e12f : d_n;
        dist = HR_DIV(dplane - o_n, d_n);
        if (risk) risk->amp = HR_RCP_BAND(fabsf(d_n));
    } else {                                                         // euclidean_distance_unified, primitive.py:162-176
        float z = hr_process_z(c, hr_zval(c, hk, 0, one_m), c.z_scale, c.samples[k], risk);
        float pos[3];
        hr_pluecker_pos(ro, rd, pos);
        const float diff[3] = {pos[0] - ro[0], pos[1] - ro[1], pos[2] - ro[2]};
        dist = z + hr_signed_base_distance(rd, diff);
    }
    if (!c.isect_mask_off) {
        if (risk) {
            const float band_d = risk->band_zc * risk->dlen * risk->amp;
            risk->hit = risk->hit || (fabsf(dist - c.near) <= band_d) || (fabsf(dist - c.far) <= band_d);
        }
        bool mask = (dist <= c.near) || (dist >= c.far);             // base.py:194
        dist = mask ? 0.0f : dist;
    }
    if (risk) risk->hit = risk->hit || (dist != 0.0f && risk->amp_cut > 0.0f && risk->amp > risk->amp_cut);
    return dist;
}

// get_base_time, utils/flow_utils.py:10-35 (jitter off).  rintf == torch.round (half to even).
HR_FN float hr_base_time(const hr_config& c, float t)
{
    if (c.num_keyframes <= 0) return 0.0f;
    float tt = t * c.flow_fac;
    tt = fminf(fmaxf(tt, 0.0f), c.flow_kmax);
    return rintf(tt - 1e-5f) * c.flow_inv_fac;
}

// Everything between the sort and the colour net for the sample of sorted rank k
// (base.py:212-257 points + contraction + re-mask; point.py:780-831 advect;
// point.py:371-396 offset).  `oc` is the contracted ray origin (hr_contract_point of ro).
// Outputs the final point and the final distance.
HR_FN void hr_sample_point(const hr_config& c, const float* hk, float dist_sorted, const float* ro, const float* rd,
                           const float* oc, float time_offset, float* p, float* dist_out)
{
    bool zero = (dist_sorted == 0.0f);
    float px = ro[0] + rd[0] * dist_sorted;
    float py = ro[1] + rd[1] * dist_sorted;
    float pz = ro[2] + rd[2] * dist_sorted;
    float dist = dist_sorted;
    if (c.contract_type != HR_CONTRACT_IDENTITY) {
        float q[3];
        hr_contract_point(c, px, py, pz, q);
        float ex = q[0] - oc[0], ey = q[1] - oc[1], ez = q[2] - oc[2];
        dist = HR_SQRTP(ex * ex + ey * ey + ez * ez);                 // contract.py:43-50
        px = q[0]; py = q[1]; pz = q[2];
    }
    dist = zero ? 0.0f : dist;                                       // base.py:246
    if (c.advect && c.use_spatial_flow) {
        const hr_head_field& f = c.f_spatial_flow;
        HR_UNROLL
        for (int i = 0; i < 3; ++i) {
            float fl = hr_apply_act_post(c.flow_act, hr_apply_act_post(f.act, hk[f.offset + i]));
            float add = fl * time_offset;
            if (i == 0) px = px + add; else if (i == 1) py = py + add; else pz = pz + add;
        }
    }
    if (c.point_offset) {
        float sig = 0.0f;
        if (c.f_offset_sigma.offset >= 0) sig = hr_apply_act_post(c.f_offset_sigma.act, hk[c.f_offset_sigma.offset]);
        float om = 1.0f - sig;
        const hr_head_field& f = c.f_point_offset;
        float o0 = hr_apply_act_post(c.offset_act, hr_apply_act_post(f.act, hk[f.offset + 0])) * om;
        float o1 = hr_apply_act_post(c.offset_act, hr_apply_act_post(f.act, hk[f.offset + 1])) * om;
        float o2 = hr_apply_act_post(c.offset_act, hr_apply_act_post(f.act, hk[f.offset + 2])) * om;
        px = px + o0; py = py + o1; pz = pz + o2;
    }
    p[0] = px; p[1] = py; p[2] = pz;
    *dist_out = dist;
}

// valid_mask (tensorf_base.py:349-353) & (distances > 0) (tensorf_no_sample.py:156)
HR_FN bool hr_sample_valid(const hr_config& c, const float* p, float dist, HrRisk* risk = nullptr, float band_p = 0.0f)
{
    if (risk && dist > 0.0f) {                  // a live sample within the points' margin of a face of the box (band_p: the caller's, from the ray's largest amp)
        float m = fminf(fabsf(p[0] - c.aabb[0]), fabsf(p[0] - c.aabb[3]));
        m = fminf(m, fminf(fabsf(p[1] - c.aabb[1]), fabsf(p[1] - c.aabb[4])));
        m = fminf(m, fminf(fabsf(p[2] - c.aabb[2]), fabsf(p[2] - c.aabb[5])));
        risk->hit = risk->hit || (m <= band_p);
    }
    bool out = (c.aabb[0] > p[0]) || (p[0] > c.aabb[3]) || (c.aabb[1] > p[1]) || (p[1] > c.aabb[4]) ||
               (c.aabb[2] > p[2]) || (p[2] > c.aabb[5]);
    return (!out) && (dist > 0.0f);
}

// normalize_coord (tensorf_base.py:308-309)
HR_FN float hr_normalize_coord(const hr_config& c, float v, int axis)
{
    return (v - c.aabb[axis]) * c.inv_size[axis] - 1.0f;
}

// normalize_time_coord (tensorf_dynamic.py:615-616)
HR_FN float hr_normalize_time(const hr_config& c, float base_t)
{
    return (base_t * c.time_scale + c.time_offset) * 2.0f - 1.0f;
}

// feature2density (tensorf_no_sample.py:82-88 / tensorf_dynamic.py:386-391)
HR_FN float hr_density(const hr_config& c, float f)
{
    if (c.density_act == HR_DENSITY_RELU) return fmaxf(f, 0.0f);
    if (c.density_act == HR_DENSITY_RELU_ABS) return fabsf(f);
    float z = f + c.density_shift;                                   // F.softplus, threshold 20
    return (z > 20.0f) ? z : log1pf(HR_EXP(z));
}

// eval_sh_bases(2, d) (utils/sh_utils.py:94-119)
HR_FN void hr_sh_deg2(float x, float y, float z, float* sh)
{
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C20 = 1.0925484305920792f, C21 = -1.0925484305920792f, C22 = 0.31539156525252005f;
    const float C23 = -1.0925484305920792f, C24 = 0.5462742152960396f;
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    sh[0] = C0;
    sh[1] = -C1 * y;
    sh[2] = C1 * z;
    sh[3] = -C1 * x;
    sh[4] = C20 * xy;
    sh[5] = C21 * yz;
    sh[6] = C22 * (2.0f * zz - xx - yy);
    sh[7] = C23 * xz;
    sh[8] = C24 * (xx - yy);
}

// One axis of F.grid_sample(align_corners=True, padding_mode='zeros'): unnormalise,
// floor, the two weights in ATen's form ((x1 - ix), (ix - x0)), and validity of the two taps.
struct hr_axis_tap {
    int i0;            // clamped index of the low tap
    int i1;            // clamped index of the high tap
    float w0, w1;      // weights, already zeroed for out-of-range taps
};
HR_FN hr_axis_tap hr_make_tap(float g, int n)
{
    hr_axis_tap t;
    float ix = ((g + 1.0f) / 2.0f) * (float)(n - 1);
    float f0 = floorf(ix);
    float f1 = f0 + 1.0f;
    float w0 = f1 - ix;
    float w1 = ix - f0;
    int i0 = (int)f0;
    int i1 = i0 + 1;
    bool ok0 = (i0 >= 0) && (i0 < n);
    bool ok1 = (i1 >= 0) && (i1 < n);
    t.w0 = ok0 ? w0 : 0.0f;
    t.w1 = ok1 ? w1 : 0.0f;
    t.i0 = ok0 ? i0 : 0;
    t.i1 = ok1 ? i1 : 0;
    return t;
}

// The same two taps in "clamped" form for kernels that address the taps as base, base + 1: i0 is clamped to [0, n-2] so
// that both texels exist, and each weight of hr_make_tap is moved to the slot where its texel now sits (a tap that ATen
// drops as out of range leaves a zero weight).  The bilinear sum is unchanged bit for bit: the terms that move carry
// their own weight, the vacated slots contribute v * 0.  Needs n >= 2.
struct hr_axis_tap_c {
    int i0;            // low tap; the high tap is i0 + 1
    float w0, w1;
};
HR_FN hr_axis_tap_c hr_make_tap_c(float g, int n)
{
    // same arithmetic as hr_make_tap for ix and the two weights (ATen's (x1 - ix), (ix - x0)), evaluated once
    float ix = ((g + 1.0f) / 2.0f) * (float)(n - 1);
    float f0 = floorf(ix);
    float f1 = f0 + 1.0f;
    const int i0 = (int)f0;
    const float w0 = ((i0 >= 0) && (i0 < n)) ? (f1 - ix) : 0.0f;          // zeroed for a tap ATen drops as out of range
    const float w1 = ((i0 + 1 >= 0) && (i0 + 1 < n)) ? (ix - f0) : 0.0f;
    int ic = i0 < 0 ? 0 : i0;
    ic = ic > n - 2 ? n - 2 : ic;
    hr_axis_tap_c t;
    t.i0 = ic;
    t.w0 = (i0 == ic) ? w0 : ((i0 + 1 == ic) ? w1 : 0.0f);
    t.w1 = (i0 == ic) ? w1 : ((i0 == ic + 1) ? w0 : 0.0f);
    return t;
}

// hr_make_tap_c for a coordinate that hr_sample_valid has let through (or the centre, 0, that masked samples are given): the
// point is inside the aabb, so ix lies in [0, n - 1] up to rounding and floor(ix) is one of -1, 0, .., n - 1 -- the three
// cases below.  Bit for bit the weights of hr_make_tap_c on that domain (tests/test_host_math.py), 7 instructions fewer per axis.
HR_FN hr_axis_tap_c hr_make_tap_in(float g, int n)
{
    const float ix = ((g + 1.0f) / 2.0f) * (float)(n - 1);
    const float f0 = floorf(ix);
    const float f1 = f0 + 1.0f;
    const int i0 = (int)f0;
    const float a = f1 - ix, b = ix - f0;
    const bool lo = i0 < 0, hi = i0 > n - 2;
    hr_axis_tap_c t;
    t.i0 = lo ? 0 : (hi ? n - 2 : i0);
    t.w0 = lo ? b : (hi ? 0.0f : a);            // floor == -1: the only tap in range is texel 0, reached as the base
    t.w1 = lo ? 0.0f : (hi ? a : b);            // floor == n - 1: the only tap in range is texel n - 1, reached as base + 1
    return t;
}

// ---------------------------------------------------------------- display pack
// to8b (utils/__init__.py:47): (255 * clip(x, 0, 1)).astype(uint8) -- the product is formed in fp32 and truncated
HR_FN uint8_t hr_to8b(float x)
{
    x = fminf(fmaxf(x, 0.0f), 1.0f);
    return (uint8_t)(255.0f * x);
}

// Source pixel of output pixel (y, x) of the viewer's buffer: NeRFGUI.test_step optionally transposes the (H, W) image
// and then flips it vertically (utils/gui_utils.py:199-205).  (h, w): the RENDERED image; the output is (w, h) when
// transposed.  Returns the row-major pixel index into the rendered image.
HR_FN int64_t hr_display_src_pixel(int y, int x, int h, int w, int transpose, int flip)
{
    const int oh = transpose ? w : h;
    if (flip) y = oh - 1 - y;
    return transpose ? (int64_t)x * w + y : (int64_t)y * w + x;
}

#endif  // HR_MATH_H
