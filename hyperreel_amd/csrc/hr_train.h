// Training path of the sample stage: forward WITHOUT the eval-mode clamp and the reverse-mode derivative
// of everything between the MLP's raw output and the pixel colour -- head activations, ray/primitive intersection,
// near/far mask, per-ray sort, contraction, flow / point offsets, VM feature gather, density, alpha compositing,
// colour decode and per-sample colour scale/shift.  It is what torch.autograd derives for the reference when
// INRSystem.training_step calls manual_backward (nlf/__init__.py:634-709) on
//   Intersect.forward (nlf/intersect/base.py:142-259), AdvectPointsEmbedding / PointOffsetEmbedding
//   (nlf/embedding/point.py:780-831, :371-396), TensorVMNoSample.forward (nlf/nets/tensorf_no_sample.py:128-280),
//   TensorVMKeyframeTime.forward (nlf/nets/tensorf_dynamic.py:645-839), raw2alpha (utils/tensorf_utils.py:242-253)
// with F.grid_sample(align_corners=True, bilinear, zeros) differentiated as ATen's grid_sampler_2d_backward does
// (values: weighted scatter-add into the 4 taps; coordinates: +-tap values, taps outside the image skipped).
//
// Three phases, each a plain function of one ray or one sample, so that the same source is compiled for the host by the
// CPU test-suite (tests/host_math) and checked there against torch.autograd on the CPU restatement of the reference:
//   A  hr_ray_train          per ray: forward, and the backward of the compositing (the only part that couples a ray's
//                            samples); leaves per-sample upstream gradients on a small tape
//   B  hr_sample_train_bwd   per (ray, sorted sample): feature-gather backward (texel scatter-adds, point gradient),
//                            contraction / flow / offset backward
//   C  hr_sample_distance_bwd  per (ray, original sample): intersection + head-activation backward
// The device runs A with one thread per ray and B / C with one thread per sample (train_kernel.hip): 32x the threads
// where the scatter-adds are.  Gradients of shared parameters are accumulated with HR_ATOMIC_ADD (hardware fp32
// atomics on the device, plain adds in the single-threaded host build).
#ifndef HR_TRAIN_H
#define HR_TRAIN_H

#include "hr_grid.h"
#include "hr_math.h"

// Gradient accumulators (texel gradients, basis_mat's, the colour table's, the per-ray decode-matrix gradient in LDS) are `hr_acc_t`.
// Default build: float, hardware fp32 atomics -- fast, and the sum depends on the order the memory system retires them in (two runs of
// the same step differ in the last bits).  HR_TRAIN_DET (train_det_kernel.hip, HR_OPT_TRAIN_DETERMINISTIC): 64-bit FIXED POINT
// through integer atomics -- integer addition is associative, so every run of a step produces the same bits whatever the order; the totals
// are converted to float once, after the last add.  The unit is a power of two chosen PER STEP from the step's largest |dL/d rgb|
// (hr_fx_scale_kernel: max |d_rgb| = 2^32 units): every gradient is linear in d_rgb, so a contribution 2^-32 of the largest still has its
// leading bit and a sum may reach 2^31 times the largest -- a fixed 2^-40 unit (round 4) dropped most bits of late-training gradients
// (d_rgb = 2 err / 3B ~ 1e-10).  A non-finite contribution, or one beyond 2^62 units, raises the unit's `bad` word and the step's
// totals convert to NaN (the fp32 path would have produced inf / NaN or a huge value there; __float2ll_rn alone turns NaN into 0 and
// saturates silently).
// The unit {units per 1.0, its inverse, the step's non-finite flag} lives in a small PER-MODEL device buffer (HrTrainArgs::fx, allocated next
// to the model's fixed-point accumulators; ADVICE r5: module-global __device__ variables were shared by every model and stream of the
// process -- two models training deterministically at once converted with each other's units).  hr_fx_scale_kernel writes it at the start
// of every step; every accumulating kernel copies it into the workgroup's LDS on entry (HR_FX_ENTER), where the macros below find it.
struct HrFxUnit { float one, inv; unsigned bad, pad_; };
#if defined(__HIPCC__) && defined(HR_TRAIN_DET)
typedef long long hr_acc_t;
__shared__ float hr_fx_one;            // units per 1.0, a power of two (this step's, this model's)
__shared__ float hr_fx_inv;
__shared__ unsigned* hr_fx_bad_p;
#define HR_FX_ENTER(a) do { if (threadIdx.x == 0) { hr_fx_one = (a).fx->one; hr_fx_inv = (a).fx->inv; hr_fx_bad_p = &(a).fx->bad; } __syncthreads(); } while (0)
__device__ __forceinline__ long long hr_to_fixed(float v)
{
    const float u = v * hr_fx_one;
    // NaN, infinity, or a contribution beyond the accumulator's range (a pathological Jacobian: __float2ll_rn would saturate silently and the
    // integer sum could wrap) -- the fp32 path would hold inf / NaN or a huge value there: the step's totals convert to NaN.  (A plain store of
    // the same value from any lane: no atomic needed.)
    if (!(fabsf(u) < 4.6e18f)) *hr_fx_bad_p = 1u;                 // 2^62
    return __float2ll_rn(u);
}
#define HR_ACC_VALUE(x) ((float)((double)(x) * (double)hr_fx_inv))
#define HR_ATOMIC_ADD(p, v) (void)atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)hr_to_fixed(v))
#define HR_ATOMIC_ADD_ACC(p, x) (void)atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)(x))      // an accumulated total onto another (exact)
#define HR_ATOMIC_ADD_RAY(p, v) (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) long long*)(p), hr_to_fixed(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define HR_ACC_ZERO 0ll
#elif defined(__HIPCC__)
typedef float hr_acc_t;
#define HR_FX_ENTER(a) do {} while (0)
#define HR_ACC_VALUE(x) (x)
#define HR_ACC_ZERO 0.0f
#define HR_ATOMIC_ADD(p, v) unsafeAtomicAdd((p), (v))
#define HR_ATOMIC_ADD_ACC(p, x) unsafeAtomicAdd((p), (x))
// per-ray / per-workgroup accumulators in LDS, shared by the sample threads: the pointer IS an LDS address, say so (a pointer
// picked from a runtime-indexed array otherwise becomes a flat atomic)
#define HR_ATOMIC_ADD_RAY(p, v) (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) float*)(p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#else
typedef float hr_acc_t;
#define HR_FX_ENTER(a) do {} while (0)
#define HR_ACC_VALUE(x) (x)
#define HR_ACC_ZERO 0.0f
#define HR_ATOMIC_ADD(p, v) (*(p) += (v))
#define HR_ATOMIC_ADD_ACC(p, x) (*(p) += (x))
#define HR_ATOMIC_ADD_RAY(p, v) (*(p) += (v))
#endif
#if defined(__HIPCC__)
// sum of v over the `lanes` adjacent lanes that work on one sample (all of them active), returned to every one of them
__device__ __forceinline__ float hr_lane_sum(float v, int lanes)
{
    if (lanes == 16) {        // one DPP row: rotate-and-add, every lane ends with the row's sum (no trip through the LDS crossbar)
#define HR_ROR_ADD(n) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, true))
        HR_ROR_ADD(8); HR_ROR_ADD(4); HR_ROR_ADD(2); HR_ROR_ADD(1);
#undef HR_ROR_ADD
        return v;
    }
    for (int d = lanes >> 1; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
#define HR_LANE_SUM(v, lanes) hr_lane_sum((v), (lanes))
#else
#define HR_LANE_SUM(v, lanes) (v)
#endif

#define HR_TRAIN_MAX_CA 64      // padded appearance slots the per-ray decode matrix holds

// Per-sample values handed from phase A to phases B and C, each (n_rays * Z)
struct HrTrainTape {
    float* ds;        // sorted pre-contraction distance of rank k
    int* src;         // original sample index of rank k
    float* dfeat;     // dL / d density feature
    float* dpre;      // dL / d decoded colour pre-activation, 3 planes of n_rays * Z
    float* ddc;       // dL / d final distance
    float* dts;       // dL / d pre-sort distance by ORIGINAL sample index (written by phase B)
    // device, lane-per-sample phase A only (NULL otherwise): what phase B would recompute per 16-lane group
    float* taps;      // 18 planes of n_rays * Z: per axis x, y, z the grid_sample tap {i0, i1 (int bits), w0, w1, s0, s1}
    float* dp;        // dL / d point, 3 planes of n_rays * Z (phase B -> hr_sample_train_point_bwd)
    int* perm;        // keyframe nets, device: the batch's ray indices grouped by keyframe row (hr_train_bucket_kernel), n_rays
};

// Workgroup-private accumulators (LDS, device only) for a texel range [lo, lo + n) of each plane pair's line / time plane: the
// whole line of a static net, the two keyframe rows a group of rays blends between for a keyframe net.  Taps outside the
// range, and pairs without an accumulator, go to the global gradient.  `pairs`: the plane pairs this pass differentiates
// (a keyframe net whose rows do not fit one workgroup's LDS together is walked in two passes; the later one adds to tape.dp).
struct HrTrainWindow {
    hr_acc_t* acc[3];
    int lo[3], n[3];
    unsigned pairs;
    int add_dp;
};

struct HrTrainArgs {
    const hr_config* cfg_dev;   // the caller's configuration (no head-column pruning: `head` is the user's layout)
    const float* rays;          // (n, ray_dim)
    const float* head;          // (n, Z * P) raw output of the last Linear, row-major
    int64_t n_rays;
    float* rgb;                 // (n, 3) forward result, not clamped (tensorf_no_sample.py:246 clamps in eval only); may be NULL
    const float* d_rgb;         // (n, 3) dL/d rgb; NULL: forward only
    float* d_head;              // (n, Z * P) dL/d head, written
    HrGridPlane planes[3];      // packed parameter values (fp32 texels)
    hr_acc_t* g_a[3];           // packed gradient accumulators, same texel layout as planes[j].a / .b
    hr_acc_t* g_b[3];
    const float* basis;         // (app_dim, n_basis_cols)
    hr_acc_t* d_basis;          // accumulated
    int n_basis_cols;
    int ca_total;
    int white_bg;               // this step's background decision: white_bg or (training and rand < 0.5), :236
    const float* color_table;   // (color_table_views, 12) per-camera [3x3 | shift] (ColorTransformEmbedding) or NULL
    hr_acc_t* d_color_table;    // accumulated
    HrTrainTape tape;
    // optional per-sample outputs of the forward, by sorted rank like hr_fields of the render path (NULL: not wanted): what a
    // field-consuming regulariser of the training loop reads (nlf/__init__.py:658-690) without a second, inference pass
    float* f_dist;              // (n, Z) final (contracted) distances
    float* f_points;            // (n, Z, 3)
    float* f_weights;           // (n, Z) render weights
    HrFxUnit* fx;               // deterministic build: this model's fixed-point unit for the step (NULL in the default build)
};

// d/dx of hr_apply_act
HR_FN float hr_act_grad(const hr_act& a, float x)
{
    const float u = x * a.inner + a.shift;
    float d = 1.0f;
    if (a.type == HR_ACT_SIGMOID) {
        const float s = 1.0f / (1.0f + expf(-u));
        d = s * (1.0f - s);
    } else if (a.type == HR_ACT_TANH) {
        const float t = tanhf(u);
        d = 1.0f - t * t;
    }
    return d * a.inner * a.outer;
}

// d/dx of outer(inner(x)) for the two stacked activations the embeddings apply (head activation, then the stage's own)
HR_FN float hr_act2_grad(const hr_act& outer, const hr_act& inner, float x)
{
    return hr_act_grad(outer, hr_apply_act(inner, x)) * hr_act_grad(inner, x);
}

// d hr_density / d feature  (relu: 0 at 0 like torch; |.|: sign; softplus with threshold 20)
HR_FN float hr_density_grad(const hr_config& c, float f)
{
    if (c.density_act == HR_DENSITY_RELU) return (f > 0.0f) ? 1.0f : 0.0f;
    if (c.density_act == HR_DENSITY_RELU_ABS) return hr_sign(f);
    const float z = f + c.density_shift;
    return (z > 20.0f) ? 1.0f : 1.0f / (1.0f + expf(-z));
}

// d hr_inverse_contract_distance / d distance
HR_FN float hr_inverse_contract_distance_grad(const hr_config& c, float distance)
{
    if (c.contract_type == HR_CONTRACT_AFFINE) return c.c_aff_fac;
    if (distance < -2.0f || distance > 2.0f) return 0.0f;           // clamp
    if (c.contract_type == HR_CONTRACT_DONERF)                      // d/dd sign(d) (|d| + 1e-8)^power / fac
        return c.c_pow_power * powf(fabsf(distance) + 1e-8f, c.c_pow_power - 1.0f) / c.c_pow_fac;
    if (fabsf(distance) < 1.0f) return c.c_d0;
    const float inv = (2.0f - fabsf(distance)) / c.c_d_scale + c.c_d_inv_end;
    return c.c_d0 / (inv * inv * c.c_d_scale);
}

// Transposed Jacobian of hr_contract_point applied to dq
HR_FN void hr_contract_point_bwd(const hr_config& c, float px, float py, float pz, const float* dq, float* dp)
{
    if (c.contract_type == HR_CONTRACT_AFFINE) {
        dp[0] = dq[0] / c.c_aff_size[0]; dp[1] = dq[1] / c.c_aff_size[1]; dp[2] = dq[2] / c.c_aff_size[2];
        return;
    }
    if (c.contract_type == HR_CONTRACT_DONERF) {
        // q = u s(n), u = p / n, s = (n fac + 1e-8)^(1/power), s' = fac / power * (n fac + 1e-8)^(1/power - 1)
        const float n = sqrtf(px * px + py * py + pz * pz);
        const float b = n * c.c_pow_fac + 1e-8f;
        const float s = powf(b, c.c_pow_inv_power);
        const float sp = c.c_pow_fac * c.c_pow_inv_power * powf(b, c.c_pow_inv_power - 1.0f);
        const float ux = px / n, uy = py / n, uz = pz / n;
        const float ud = ux * dq[0] + uy * dq[1] + uz * dq[2];
        const float k = s / n;
        dp[0] = k * (dq[0] - ux * ud) + sp * ux * ud;
        dp[1] = k * (dq[1] - uy * ud) + sp * uy * ud;
        dp[2] = k * (dq[2] - uz * ud) + sp * uz * ud;
        return;
    }
    const float x = px / c.c_r0, y = py / c.c_r0, z = pz / c.c_r0;
    const float n = sqrtf(x * x + y * y + z * z);
    if (n < 1.0f) {
        dp[0] = dq[0] / c.c_r0; dp[1] = dq[1] / c.c_r0; dp[2] = dq[2] / c.c_r0;
        return;
    }
    // q = (x / n) * s(n), s = 2 - (1/n - r_inv_end) * r_scale, s' = r_scale / n^2
    const float s = 2.0f - (1.0f / n - c.c_r_inv_end) * c.c_r_scale;
    const float sp = c.c_r_scale / (n * n);
    const float ux = x / n, uy = y / n, uz = z / n;
    const float ud = ux * dq[0] + uy * dq[1] + uz * dq[2];
    const float k = s / n;
    dp[0] = (k * (dq[0] - ux * ud) + sp * ux * ud) / c.c_r0;
    dp[1] = (k * (dq[1] - uy * ud) + sp * uy * ud) / c.c_r0;
    dp[2] = (k * (dq[2] - uz * ud) + sp * uz * ud) / c.c_r0;
}

// d hr_quadratic_t / d radius  (everything else of the quadratic is a function of the ray only)
HR_FN float hr_quadratic_t_grad_radius(float oo, float dd, float od, float radius)
{
    const float a = dd, b = 2.0f * od, cc = oo - radius * radius;
    float disc = b * b - 4.0f * a * cc;
    if (disc <= 0.0f) return 0.0f;                    // clamped to 0 and t replaced by the constant 0
    const float sq = sqrtf(disc + 1e-8f);
    const float t2 = (-b - sq) / (2.0f * a);
    // d disc / d radius = 8 a r; d t1 = +(1 / (2 sq)) / (2 a) d disc, d t2 = -(...)
    const float g = 2.0f * radius / sq;
    return ((t2 < 0.0f) || (radius < 0.0f)) ? g : -g;
}

// ---------------------------------------------------------------- forward-mode duals for the long intersections
// sphere_new / cylinder_new (primitive.py:305-363, 490-545: primitive frame from 6 head channels, sample recycling),
// deformable_voxel_grid (voxel.py:178-213: learned plane normals) and sphere / cylinder with learned origins
// (primitive.py:410-431) map up to 8 activated head channels of a sample to one distance.  Their derivative is taken
// by carrying the 8 partials through the same arithmetic as hr_sample_distance (value part identical), not by a
// hand-derived adjoint.
#define HR_DN 8
struct hr_dual {
    float v;
    float d[HR_DN];
};
HR_FN hr_dual hr_dconst(float v)
{
    hr_dual r;
    r.v = v;
    for (int i = 0; i < HR_DN; ++i) r.d[i] = 0.0f;
    return r;
}
HR_FN hr_dual hr_dvar(float v, int i)
{
    hr_dual r = hr_dconst(v);
    r.d[i] = 1.0f;
    return r;
}
// r = f(x) with known f(x.v) and f'(x.v)
HR_FN hr_dual hr_dchain(const hr_dual& x, float fv, float fp)
{
    hr_dual r;
    r.v = fv;
    for (int i = 0; i < HR_DN; ++i) r.d[i] = fp * x.d[i];
    return r;
}
HR_FN hr_dual operator+(const hr_dual& a, const hr_dual& b)
{
    hr_dual r;
    r.v = a.v + b.v;
    for (int i = 0; i < HR_DN; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
HR_FN hr_dual operator-(const hr_dual& a, const hr_dual& b)
{
    hr_dual r;
    r.v = a.v - b.v;
    for (int i = 0; i < HR_DN; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
HR_FN hr_dual operator*(const hr_dual& a, const hr_dual& b)
{
    hr_dual r;
    r.v = a.v * b.v;
    for (int i = 0; i < HR_DN; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
HR_FN hr_dual operator/(const hr_dual& a, const hr_dual& b)
{
    hr_dual r;
    r.v = a.v / b.v;
    for (int i = 0; i < HR_DN; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
    return r;
}
HR_FN hr_dual operator+(const hr_dual& a, float b) { hr_dual r = a; r.v = a.v + b; return r; }
HR_FN hr_dual operator-(const hr_dual& a, float b) { hr_dual r = a; r.v = a.v - b; return r; }
HR_FN hr_dual operator-(float a, const hr_dual& b) { return hr_dconst(a) - b; }
HR_FN hr_dual operator*(const hr_dual& a, float b) { return hr_dchain(a, a.v * b, b); }
HR_FN hr_dual operator*(float a, const hr_dual& b) { return hr_dchain(b, a * b.v, a); }
HR_FN hr_dual operator-(const hr_dual& a) { return hr_dchain(a, -a.v, -1.0f); }
HR_FN hr_dual hr_dsqrt(const hr_dual& a)
{
    const float s = sqrtf(a.v);
    return hr_dchain(a, s, 0.5f / s);
}
HR_FN hr_dual hr_ddot3(const hr_dual* a, const hr_dual* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
// torch.norm of a 3-vector: the sub-gradient at 0 is 0
HR_FN hr_dual hr_dnorm3(const hr_dual* a)
{
    const hr_dual ss = hr_ddot3(a, a);
    const float n = sqrtf(ss.v);
    return hr_dchain(ss, n, n > 0.0f ? 0.5f / n : 0.0f);
}
// F.normalize(p=2, eps=1e-12)
HR_FN void hr_dnormalize3(const hr_dual* a, hr_dual* r)
{
    const hr_dual n = hr_dnorm3(a);
    if (n.v > 1e-12f) { r[0] = a[0] / n; r[1] = a[1] / n; r[2] = a[2] / n; }
    else { r[0] = a[0] * 1e12f; r[1] = a[1] * 1e12f; r[2] = a[2] * 1e12f; }
}
HR_FN void hr_dcross3(const hr_dual* a, const hr_dual* b, hr_dual* r)
{
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}
HR_FN void hr_dpluecker_pos(const hr_dual* o, const hr_dual* d, hr_dual* pos)
{
    hr_dual dn[3], m[3];
    hr_dnormalize3(d, dn);
    hr_dcross3(o, dn, m);
    hr_dcross3(dn, m, pos);
}
HR_FN hr_dual hr_dsigned_base_distance(const hr_dual* d, const hr_dual* diff)
{
    return hr_dnorm3(diff) * hr_sign(hr_ddot3(d, diff).v);
}
HR_FN hr_dual hr_dquadratic_t(const hr_dual& oo, const hr_dual& dd, const hr_dual& od, const hr_dual& radius)
{
    const hr_dual a = dd, b = od * 2.0f, cc = oo - radius * radius;
    const hr_dual disc = b * b - (a * cc) * 4.0f;
    if (disc.v <= 0.0f) return hr_dconst(0.0f);             // clamped, and t replaced by the constant 0
    const hr_dual sq = hr_dsqrt(disc + 1e-8f);
    const hr_dual t1 = (sq - b) / (a * 2.0f), t2 = (-b - sq) / (a * 2.0f);
    return ((t2.v < 0.0f) || (radius.v < 0.0f)) ? t1 : t2;
}
HR_FN hr_dual hr_dprocess_z(const hr_config& c, const hr_dual& z, float scale, float anchor)
{
    const hr_dual x = z * scale + anchor;
    if (!c.contract_samples) return x;
    return hr_dchain(x, hr_inverse_contract_distance(c, x.v), hr_inverse_contract_distance_grad(c, x.v));
}
// hr_sample_distance before the mask for the intersections above; zv[ch]: activated z_vals channel ch times (1 - sigma)
HR_FN hr_dual hr_sample_distance_dual(const hr_config& c, const hr_dual* zv, int k, const float* ro, const float* rd)
{
    if (c.isect_type == HR_ISECT_SPHERE || c.isect_type == HR_ISECT_CYLINDER) {        // primitive.py:410-431
        hr_dual o[3], d[3];
        for (int i = 0; i < 3; ++i) {
            const hr_dual s = zv[i] * c.origin_scale + c.origin_initial[i];
            o[i] = s * ro[i];
            d[i] = s * rd[i];
        }
        const hr_dual radius = hr_dprocess_z(c, zv[3], c.z_scale, c.samples[k]);
        if (c.isect_type == HR_ISECT_SPHERE) return hr_dquadratic_t(hr_ddot3(o, o), hr_ddot3(d, d), hr_ddot3(o, d), radius);
        return hr_dquadratic_t(o[0] * o[0] + o[2] * o[2], d[0] * d[0] + d[2] * d[2], o[0] * d[0] + o[2] * d[2], radius);
    }
    if (c.isect_type == HR_ISECT_DEFORMABLE_VOXEL_GRID) {                               // voxel.py:184-213
        const int axis = k % c.dvg_axes;
        hr_dual n0[3], n[3];
        for (int i = 0; i < 3; ++i) n0[i] = zv[i] * c.dvg_normal_scale + c.dvg_normals[3 * axis + i];
        hr_dnormalize3(n0, n);
        const hr_dual dplane = hr_dprocess_z(c, zv[3], c.z_scale, c.samples[k]);
        const hr_dual o_n = (n[0] * ro[0] + n[1] * ro[1]) + n[2] * ro[2];
        hr_dual d_n = (n[0] * rd[0] + n[1] * rd[1]) + n[2] * rd[2];
        if (fabsf(d_n.v) < 1e-5f) d_n = hr_dconst(1e12f);
        return (dplane - o_n) / d_n;
    }
    // sphere_new / cylinder_new, primitive.py:498-545, :313-363 (hr_isect_new)
    hr_dual o[3], d[3], dn[3];
    for (int i = 0; i < 3; ++i) {
        const hr_dual org = zv[i] * c.origin_scale;
        const hr_dual rs = zv[3 + i] * c.resize_scale + c.resize_initial[i];
        o[i] = (ro[i] - org) * rs;
        d[i] = rs * rd[i];
    }
    const hr_dual raw = hr_dprocess_z(c, zv[6], c.z_scale, c.samples[k]);
    const hr_dual radius = hr_dprocess_z(c, zv[7], c.z_scale, c.samples[k]);
    const hr_dual dnorm = hr_dnorm3(d);
    hr_dnormalize3(d, dn);
    hr_dual t, min_radius, base_distance;
    if (c.isect_type == HR_ISECT_SPHERE_NEW) {
        t = hr_dquadratic_t(hr_ddot3(o, o), hr_ddot3(dn, dn), hr_ddot3(o, dn), radius);
        hr_dual pos[3];
        hr_dpluecker_pos(o, dn, pos);
        min_radius = hr_dnorm3(pos);
        const hr_dual diff[3] = {pos[0] - o[0], pos[1] - o[1], pos[2] - o[2]};
        base_distance = hr_dsigned_base_distance(dn, diff);
    } else {
        t = hr_dquadratic_t(o[0] * o[0] + o[2] * o[2], dn[0] * dn[0] + dn[2] * dn[2], o[0] * dn[0] + o[2] * dn[2], radius);
        const hr_dual oc[3] = {o[0], hr_dconst(0.0f), o[2]}, dc[3] = {dn[0], hr_dconst(0.0f), dn[2]};
        hr_dual pos[3];
        hr_dpluecker_pos(oc, dc, pos);
        min_radius = hr_dnorm3(pos);
        const hr_dual diff[3] = {pos[0] - oc[0], pos[1] - oc[1], pos[2] - oc[2]};
        base_distance = hr_dsigned_base_distance(dc, diff) / hr_dnorm3(dc);
    }
    if (fabsf(radius.v) < min_radius.v + 4.0f * c.z_scale) t = raw + base_distance;
    return t / (dnorm + 1e-5f);
}

// Which models the training path differentiates.  Returns NULL when supported, else the reason.
#if defined(__HIPCC__)
__host__ __device__ inline
#else
static inline
#endif
const char* hr_train_unsupported(const hr_config& c)
{
    if (c.grid_dtype != HR_GRID_FP32) return "float16 grids";
    return nullptr;
}

// Backward of hr_sample_distance for the sample of ORIGINAL index k: dt = dL/d(distance after the mask).
// Adds to the z_vals channel and the intersect-sigma entry of dhk (the P gradients of sample k's head values).
HR_FN void hr_sample_distance_bwd(const hr_config& c, const float* hk, int k, const float* ro, const float* rd, float dt, float* dhk)
{
    if (dt == 0.0f) return;
    if (!c.isect_mask_off) {
        const float dist = hr_sample_distance(c, hk, k, ro, rd);     // 0 when masked (near > 0 in every shipped dataset)
        if (dist == 0.0f) return;
    }
    float sigma = 0.0f;
    if (c.f_isect_sigma.offset >= 0) sigma = hr_apply_act(c.f_isect_sigma.act, hk[c.f_isect_sigma.offset]);
    const float one_m = 1.0f - sigma;
    const bool origins = (c.isect_type == HR_ISECT_SPHERE || c.isect_type == HR_ISECT_CYLINDER) && c.origin_scale != 0.0f;
    if (origins || c.isect_type == HR_ISECT_SPHERE_NEW || c.isect_type == HR_ISECT_CYLINDER_NEW ||
        c.isect_type == HR_ISECT_DEFORMABLE_VOXEL_GRID) {
        // several head channels per sample: partials carried through the forward arithmetic (hr_sample_distance_dual)
        const int nch = (c.isect_type == HR_ISECT_SPHERE_NEW || c.isect_type == HR_ISECT_CYLINDER_NEW) ? 8 : 4;
        float zact[HR_DN];
        hr_dual zv[HR_DN];
        for (int i = 0; i < HR_DN; ++i) {
            zact[i] = (i < nch) ? hr_apply_act(c.z_act, hr_apply_act(c.f_z_vals.act, hk[c.f_z_vals.offset + i])) : 0.0f;
            zv[i] = hr_dvar(zact[i] * one_m, i);
        }
        const hr_dual dist = hr_sample_distance_dual(c, zv, k, ro, rd);
        float dsum = 0.0f;                       // dL / d (1 - sigma)
        for (int i = 0; i < nch; ++i) {
            const float g = dt * dist.d[i];
            if (g == 0.0f) continue;
            dhk[c.f_z_vals.offset + i] += g * one_m * hr_act2_grad(c.z_act, c.f_z_vals.act, hk[c.f_z_vals.offset + i]);
            dsum += g * zact[i];
        }
        if (c.f_isect_sigma.offset >= 0)
            dhk[c.f_isect_sigma.offset] += -dsum * hr_act_grad(c.f_isect_sigma.act, hk[c.f_isect_sigma.offset]);
        return;
    }
    int ch = 0;
    float scale = c.z_scale;
    float dzp;                                   // dL / d processed z (or radius)
    if (c.isect_type == HR_ISECT_Z_PLANE) {
        const float dd = (fabsf(rd[2]) < 1e-5f) ? 1e12f : rd[2];
        dzp = dt / dd;
    } else if (c.isect_type == HR_ISECT_SPHERE || c.isect_type == HR_ISECT_CYLINDER) {
        ch = 3;
        const float sx = c.origin_initial[0], sy = c.origin_initial[1], sz = c.origin_initial[2];
        const float radius = hr_process_z(c, hr_zval(c, hk, 3, one_m), c.z_scale, c.samples[k]);
        const float ox = ro[0] * sx, oy = ro[1] * sy, oz = ro[2] * sz;
        const float dx = rd[0] * sx, dy = rd[1] * sy, dz = rd[2] * sz;
        if (c.isect_type == HR_ISECT_SPHERE)
            dzp = dt * hr_quadratic_t_grad_radius(ox * ox + oy * oy + oz * oz, dx * dx + dy * dy + dz * dz, ox * dx + oy * dy + oz * dz, radius);
        else
            dzp = dt * hr_quadratic_t_grad_radius(ox * ox + oz * oz, dx * dx + dz * dz, ox * dx + oz * dz, radius);
    } else if (c.isect_type == HR_ISECT_VOXEL_GRID) {
        const int axis = k % 3;
        scale = c.voxel_scale[axis];
        const float d = (axis == 0) ? rd[0] : (axis == 1) ? rd[1] : rd[2];
        const float dd = (fabsf(d) < 1e-5f) ? 1e12f : d;
        dzp = dt / dd;
        if (c.isect_outward) dzp = dzp * hr_sign(d);
    } else {                                     // euclidean_distance_unified: dist = z + f(ray)
        dzp = dt;
    }
    // processed z = icd(zval * scale + anchor);  zval = act_z(act_f(h)) * (1 - sigma)
    const float h = hk[c.f_z_vals.offset + ch];
    const float zact = hr_apply_act(c.z_act, hr_apply_act(c.f_z_vals.act, h));
    const float x = zact * one_m * scale + c.samples[k];
    float dx_ = dzp;
    if (c.contract_samples) dx_ = dx_ * hr_inverse_contract_distance_grad(c, x);
    const float dzval = dx_ * scale;
    dhk[c.f_z_vals.offset + ch] += dzval * one_m * hr_act2_grad(c.z_act, c.f_z_vals.act, h);
    if (c.f_isect_sigma.offset >= 0)
        dhk[c.f_isect_sigma.offset] += -(dzval * zact) * hr_act_grad(c.f_isect_sigma.act, hk[c.f_isect_sigma.offset]);
}

// Backward of hr_sample_point for the sample of sorted rank k: dp = dL/d point, ddist = dL/d (final distance).
// Adds the flow / offset / offset-sigma gradients to dhk and returns dL/d (sorted pre-contraction distance).
HR_FN float hr_sample_point_bwd(const hr_config& c, const float* hk, float dist_sorted, const float* ro, const float* rd,
                                const float* oc, float time_offset, const float* dp, float ddist, float* dhk)
{
    const bool zero = (dist_sorted == 0.0f);
    const float px = ro[0] + rd[0] * dist_sorted, py = ro[1] + rd[1] * dist_sorted, pz = ro[2] + rd[2] * dist_sorted;
    float dP[3] = {dp[0], dp[1], dp[2]};
    float dt = 0.0f;
    if (c.contract_type != HR_CONTRACT_IDENTITY) {
        float q[3];
        hr_contract_point(c, px, py, pz, q);
        const float ex = q[0] - oc[0], ey = q[1] - oc[1], ez = q[2] - oc[2];
        const float dist = sqrtf(ex * ex + ey * ey + ez * ez);
        float dq[3] = {dp[0], dp[1], dp[2]};
        if (!zero && dist > 0.0f) {                                  // torch.norm backward; base.py:246 cuts it where dist == 0
            dq[0] += ddist * ex / dist; dq[1] += ddist * ey / dist; dq[2] += ddist * ez / dist;
        }
        hr_contract_point_bwd(c, px, py, pz, dq, dP);
    } else if (!zero) {
        dt = ddist;
    }
    dt += rd[0] * dP[0] + rd[1] * dP[1] + rd[2] * dP[2];
    if (c.advect && c.use_spatial_flow) {
        const hr_head_field& f = c.f_spatial_flow;
        for (int i = 0; i < 3; ++i) dhk[f.offset + i] += dp[i] * time_offset * hr_act2_grad(c.flow_act, f.act, hk[f.offset + i]);
    }
    if (c.point_offset) {
        float sig = 0.0f;
        if (c.f_offset_sigma.offset >= 0) sig = hr_apply_act(c.f_offset_sigma.act, hk[c.f_offset_sigma.offset]);
        const float om = 1.0f - sig;
        const hr_head_field& f = c.f_point_offset;
        float dsig = 0.0f;
        for (int i = 0; i < 3; ++i) {
            const float h = hk[f.offset + i];
            dhk[f.offset + i] += dp[i] * om * hr_act2_grad(c.offset_act, f.act, h);
            dsig -= dp[i] * hr_apply_act(c.offset_act, hr_apply_act(f.act, h));
        }
        if (c.f_offset_sigma.offset >= 0) dhk[c.f_offset_sigma.offset] += dsig * hr_act_grad(c.f_offset_sigma.act, hk[c.f_offset_sigma.offset]);
    }
    return dt;
}

// One axis of grid_sample with what its backward needs: d w0 / d ix and d w1 / d ix (taps outside the image contribute
// nothing, grid_sampler_2d_backward) and d ix / d g = (n - 1) / 2.
struct hr_axis_tap_g {
    hr_axis_tap t;
    float s0, s1;
    float mult;
};
HR_FN hr_axis_tap_g hr_make_tap_g(float g, int n)
{
    hr_axis_tap_g r;
    r.t = hr_make_tap(g, n);
    const float ix = ((g + 1.0f) / 2.0f) * (float)(n - 1);
    const int i0 = (int)floorf(ix), i1 = i0 + 1;
    r.s0 = (i0 >= 0 && i0 < n) ? -1.0f : 0.0f;
    r.s1 = (i1 >= 0 && i1 < n) ? 1.0f : 0.0f;
    r.mult = 0.5f * (float)(n - 1);
    return r;
}

// Texel indices and weights of one plane pair for a sample (the arithmetic of hr_gather_plane in sample_kernel.hip)
struct HrTrainTaps {
    int ia[4]; float wa[4];     // plane: nw, ne, sw, se
    int ib[4]; float wb[4];     // line: low, high / time plane: 4
    int nb;                     // 2 or 4
};
HR_FN HrTrainTaps hr_train_taps(const HrGridPlane& g, const hr_axis_tap& tx, const hr_axis_tap& ty, const hr_axis_tap& bx, const hr_axis_tap& by)
{
    HrTrainTaps t;
    t.wa[0] = tx.w0 * ty.w0; t.wa[1] = tx.w1 * ty.w0; t.wa[2] = tx.w0 * ty.w1; t.wa[3] = tx.w1 * ty.w1;
    t.ia[0] = ty.i0 * g.aw + tx.i0; t.ia[1] = ty.i0 * g.aw + tx.i1; t.ia[2] = ty.i1 * g.aw + tx.i0; t.ia[3] = ty.i1 * g.aw + tx.i1;
    if (g.bw == 1) {
        t.nb = 2;
        t.ib[0] = bx.i0; t.ib[1] = bx.i1; t.ib[2] = 0; t.ib[3] = 0;
        t.wb[0] = bx.w0; t.wb[1] = bx.w1; t.wb[2] = 0.0f; t.wb[3] = 0.0f;
    } else {
        t.nb = 4;
        t.ib[0] = by.i0 * g.bw + bx.i0; t.ib[1] = by.i0 * g.bw + bx.i1; t.ib[2] = by.i1 * g.bw + bx.i0; t.ib[3] = by.i1 * g.bw + bx.i1;
        t.wb[0] = bx.w0 * by.w0; t.wb[1] = bx.w1 * by.w0; t.wb[2] = bx.w0 * by.w1; t.wb[3] = bx.w1 * by.w1;
    }
    return t;
}

HR_FN int hr_plane_a0(int j) { return (j == 2) ? 1 : 0; }     // MAT_MODE[j][0]
HR_FN int hr_plane_a1(int j) { return (j == 0) ? 1 : 2; }     // MAT_MODE[j][1]
HR_FN int hr_plane_v(int j) { return 2 - j; }                 // VEC_MODE[j] == MAT_MODE_TIME[j][0]

// Forward gather of one sample over the three plane pairs: density feature and the three decoded pre-activations.
HR_FN void hr_train_gather(const HrTrainArgs& a, const hr_axis_tap_g* ax, const hr_axis_tap_g& at, const float* M, int CA,
                           float* sig_feat, float* pre)
{
    float s = 0.0f, p0 = 0.0f, p1 = 0.0f, p2 = 0.0f;
    for (int j = 0; j < 3; ++j) {
        const HrGridPlane& g = a.planes[j];
        const int ng = g.cd4 + g.ca4;
        if (ng == 0) continue;
        const HrTrainTaps t = hr_train_taps(g, ax[hr_plane_a0(j)].t, ax[hr_plane_a1(j)].t, ax[hr_plane_v(j)].t, at.t);
        const float* A = reinterpret_cast<const float*>(g.a);
        const float* B = reinterpret_cast<const float*>(g.b);
        for (int ch = 0; ch < 4 * ng; ++ch) {
            float pa = A[(size_t)t.ia[0] * g.tex + ch] * t.wa[0];
            pa = fmaf(A[(size_t)t.ia[1] * g.tex + ch], t.wa[1], pa);
            pa = fmaf(A[(size_t)t.ia[2] * g.tex + ch], t.wa[2], pa);
            pa = fmaf(A[(size_t)t.ia[3] * g.tex + ch], t.wa[3], pa);
            float pb = B[(size_t)t.ib[0] * g.tex + ch] * t.wb[0];
            pb = fmaf(B[(size_t)t.ib[1] * g.tex + ch], t.wb[1], pb);
            if (t.nb == 4) {
                pb = fmaf(B[(size_t)t.ib[2] * g.tex + ch], t.wb[2], pb);
                pb = fmaf(B[(size_t)t.ib[3] * g.tex + ch], t.wb[3], pb);
            }
            const float f = pa * pb;
            if (ch < 4 * g.cd4) {
                s = s + f;
            } else {
                const int slot = g.app_off + (ch - 4 * g.cd4);
                p0 = fmaf(M[slot], f, p0); p1 = fmaf(M[CA + slot], f, p1); p2 = fmaf(M[2 * CA + slot], f, p2);
            }
        }
    }
    *sig_feat = s; pre[0] = p0; pre[1] = p1; pre[2] = p2;
}

// Backward gather of ONE CHANNEL `ch` of plane pair j for a sample: dfeat = dL/d density feature, dpre = dL/d decoded
// pre-activations.  Scatter-adds the channel's texel gradients, adds the channel's share of the ray's decode-matrix
// gradient to dM and accumulates dL/d (unnormalised tap coordinates) of the pair's three axes in d3 = {x, y, line axis}.
// The device gives every channel of a texel its own lane: the 8..16 channels of a tap are then one contiguous 32..64-byte
// run per atomic instruction, which the memory system retires 17x faster than 64 lanes on 64 different cache lines
// (tools/atomic_ubench.hip: 331 vs 19.5 G atomics/s).
HR_FN void hr_train_gather_bwd_channel(const HrTrainArgs& a, int j, const HrTrainTaps& t, const hr_axis_tap_g& gx, const hr_axis_tap_g& gy,
                                       const hr_axis_tap_g& gv, const hr_axis_tap_g& at, int ch, const float* M, hr_acc_t* dM, int CA,
                                       float dfeat, const float* dpre, float* d3, const HrTrainWindow* win = nullptr)
{
    const HrGridPlane& g = a.planes[j];
    const float* A = reinterpret_cast<const float*>(g.a);
    const float* B = reinterpret_cast<const float*>(g.b);
    const float a00 = A[(size_t)t.ia[0] * g.tex + ch], a01 = A[(size_t)t.ia[1] * g.tex + ch];
    const float a10 = A[(size_t)t.ia[2] * g.tex + ch], a11 = A[(size_t)t.ia[3] * g.tex + ch];
    const float b0 = B[(size_t)t.ib[0] * g.tex + ch], b1 = B[(size_t)t.ib[1] * g.tex + ch];
    float b2 = 0.0f, b3 = 0.0f;
    if (t.nb == 4) { b2 = B[(size_t)t.ib[2] * g.tex + ch]; b3 = B[(size_t)t.ib[3] * g.tex + ch]; }
    const float pa = fmaf(a11, t.wa[3], fmaf(a10, t.wa[2], fmaf(a01, t.wa[1], a00 * t.wa[0])));
    const float pb = fmaf(b3, t.wb[3], fmaf(b2, t.wb[2], fmaf(b1, t.wb[1], b0 * t.wb[0])));
    const float f = pa * pb;
    float u;
    if (ch < 4 * g.cd4) {
        u = dfeat;
    } else {
        const int slot = g.app_off + (ch - 4 * g.cd4);
        u = dpre[0] * M[slot] + dpre[1] * M[CA + slot] + dpre[2] * M[2 * CA + slot];
        HR_ATOMIC_ADD_RAY(dM + slot, dpre[0] * f); HR_ATOMIC_ADD_RAY(dM + CA + slot, dpre[1] * f); HR_ATOMIC_ADD_RAY(dM + 2 * CA + slot, dpre[2] * f);
    }
    if (u == 0.0f) return;
    const float dpa = u * pb, dpb = u * pa;
    hr_acc_t* GA = a.g_a[j];
    hr_acc_t* GB = a.g_b[j];
    for (int i = 0; i < 4; ++i)
        if (t.wa[i] != 0.0f) HR_ATOMIC_ADD(GA + (size_t)t.ia[i] * g.tex + ch, dpa * t.wa[i]);
    // the line of a static net has a few hundred texels that EVERY sample of the batch hits, the time plane of a keyframe net
    // two rows that every sample of a ray at that time hits: on the device the caller can hand a workgroup-private accumulator
    // (LDS) for such a texel range, which it adds to the global gradient when it moves on
    for (int i = 0; i < t.nb; ++i) {
        if (t.wb[i] == 0.0f) continue;
        const int rel = win ? t.ib[i] - win->lo[j] : 0;
        if (win && win->acc[j] && (unsigned)rel < (unsigned)win->n[j]) HR_ATOMIC_ADD_RAY(win->acc[j] + (size_t)rel * g.tex + ch, dpb * t.wb[i]);
        else HR_ATOMIC_ADD(GB + (size_t)t.ib[i] * g.tex + ch, dpb * t.wb[i]);
    }
    // coordinates: d(weights)/d ix = (s0, s1) per axis
    d3[0] += dpa * ((a00 * gx.s0 + a01 * gx.s1) * gy.t.w0 + (a10 * gx.s0 + a11 * gx.s1) * gy.t.w1);
    d3[1] += dpa * ((a00 * gx.t.w0 + a01 * gx.t.w1) * gy.s0 + (a10 * gx.t.w0 + a11 * gx.t.w1) * gy.s1);
    if (t.nb == 2) d3[2] += dpb * (b0 * gv.s0 + b1 * gv.s1);
    else d3[2] += dpb * ((b0 * gv.s0 + b1 * gv.s1) * at.t.w0 + (b2 * gv.s0 + b3 * gv.s1) * at.t.w1);
}

// Backward gather of one sample, channels `ch0, ch0 + stride, ...` of every plane pair: returns this caller's share of
// dL/d normalised coordinates in dpn[3] (the host walks all channels with stride 1; on the device the 16 lanes of a
// sample take stride 16 and add their shares up).
HR_FN void hr_train_gather_bwd(const HrTrainArgs& a, const hr_axis_tap_g* ax, const hr_axis_tap_g& at, const float* M, hr_acc_t* dM, int CA,
                               float dfeat, const float* dpre, float* dpn, int ch0 = 0, int stride = 1, const HrTrainWindow* win = nullptr)
{
    dpn[0] = 0.0f; dpn[1] = 0.0f; dpn[2] = 0.0f;
    for (int j = 0; j < 3; ++j) {
        const HrGridPlane& g = a.planes[j];
        const int nch = 4 * (g.cd4 + g.ca4);
        if (nch == 0 || (win && !((win->pairs >> j) & 1u))) continue;
        const hr_axis_tap_g& gx = ax[hr_plane_a0(j)];
        const hr_axis_tap_g& gy = ax[hr_plane_a1(j)];
        const hr_axis_tap_g& gv = ax[hr_plane_v(j)];
        const HrTrainTaps t = hr_train_taps(g, gx.t, gy.t, gv.t, at.t);
        float d3[3] = {0.0f, 0.0f, 0.0f};
        for (int ch = ch0; ch < nch; ch += stride)
            hr_train_gather_bwd_channel(a, j, t, gx, gy, gv, at, ch, M, dM, CA, dfeat, dpre, d3, win);
        dpn[hr_plane_a0(j)] += d3[0] * gx.mult;
        dpn[hr_plane_a1(j)] += d3[1] * gy.mult;
        dpn[hr_plane_v(j)] += d3[2] * gv.mult;
    }
}

// Coefficient of appearance slot `pos` in colour channel c's decode (the M matrix of sample_kernel.hip) and the
// basis_mat column it comes from (-1 for a padding slot)
HR_FN int hr_train_slot_col(const HrTrainArgs& a, int pos)
{
    int col = -1;
    for (int j = 0; j < 3; ++j) {
        const int rel = pos - a.planes[j].app_off;
        if (rel >= 0 && rel < a.planes[j].app_real && a.planes[j].ca4 > 0) col = a.planes[j].app_real_off + rel;
    }
    return col;
}

// M[cc][pos] of the ray's decode matrix (RGB: basis_mat rows; SH: basis rows folded with the view direction's SH basis)
HR_FN float hr_train_decode_coef(const hr_config& c, const HrTrainArgs& a, const float* sh, int cc, int pos)
{
    const int col = hr_train_slot_col(a, pos);
    if (col < 0) return 0.0f;
    if (c.shading != HR_SHADING_SH) return a.basis[cc * a.n_basis_cols + col];
    float v = 0.0f;
    for (int j = 0; j < 9; ++j) v = fmaf(sh[j], a.basis[(cc * 9 + j) * a.n_basis_cols + col], v);
    return v;
}

// dM[cc][pos] of one ray -> basis_mat gradient.  `acc`: a workgroup-private copy of the whole gradient (LDS, device only) that
// the workgroup adds to the global one once at the end -- as global atomics these adds are 48 (RGB) or 432 (SH) per ray onto
// the same few hundred bytes from every ray of the batch: 0.5 of phase B's 0.9 ms on the DoNeRF scene
HR_FN void hr_train_fold_basis(const hr_config& c, const HrTrainArgs& a, const float* sh, int cc, int pos, float v, hr_acc_t* acc = nullptr)
{
    if (v == 0.0f) return;
    const int col = hr_train_slot_col(a, pos);
    if (col < 0) return;
    if (c.shading == HR_SHADING_SH) {
        for (int j = 0; j < 9; ++j) {
            if (acc) HR_ATOMIC_ADD_RAY(acc + (cc * 9 + j) * a.n_basis_cols + col, sh[j] * v);
            else HR_ATOMIC_ADD(a.d_basis + (cc * 9 + j) * a.n_basis_cols + col, sh[j] * v);
        }
    } else {
        if (acc) HR_ATOMIC_ADD_RAY(acc + cc * a.n_basis_cols + col, v);
        else HR_ATOMIC_ADD(a.d_basis + cc * a.n_basis_cols + col, v);
    }
}

// Per-ray quantities every phase recomputes from the ray itself
struct HrTrainRay {
    float ro[3], rd[3], oc[3];
    float time_off;
    hr_axis_tap_g tap_t;
};
HR_FN HrTrainRay hr_train_ray(const hr_config& c, const float* r)
{
    HrTrainRay q;
    for (int i = 0; i < 3; ++i) { q.ro[i] = r[i] - c.isect_origin[i]; q.rd[i] = r[3 + i]; q.oc[i] = 0.0f; }
    if (c.contract_type != HR_CONTRACT_IDENTITY) hr_contract_point(c, q.ro[0], q.ro[1], q.ro[2], q.oc);
    float base_t = 0.0f;
    q.time_off = 0.0f;
    if (c.advect) { base_t = hr_base_time(c, r[c.ray_dim - 1]); q.time_off = r[c.ray_dim - 1] - base_t; }
    q.tap_t = hr_make_tap_g(c.video ? hr_normalize_time(c, base_t) : 0.0f, c.video ? c.num_keyframes : 2);
    return q;
}

// The keyframe row a ray's time taps start at (the i0 of hr_train_ray's tap_t before clamping, held to [-1, K - 1]): rays with the
// same row blend between the same two rows of every time plane
HR_FN int hr_train_time_row(const hr_config& c, const float* r)
{
    float base_t = 0.0f;
    if (c.advect) base_t = hr_base_time(c, r[c.ray_dim - 1]);
    const int n = c.video ? c.num_keyframes : 2;
    const float g = c.video ? hr_normalize_time(c, base_t) : 0.0f;
    const float ix = ((g + 1.0f) / 2.0f) * (float)(n - 1);
    const float f0 = fminf(fmaxf(floorf(ix), -1.0f), (float)(n - 1));
    return (int)f0;
}

// Phase B: the sample of sorted rank k of `ray`.  M / dM: the ray's decode matrix and its gradient accumulator (3 * CA).
// The host calls it once per sample; on the device the `lanes` (a power of two <= 64, adjacent lanes of one wavefront)
// threads of a sample call it together with their `lane`, split the channels between them and combine their shares of
// the point gradient with HR_LANE_SUM.
HR_FN void hr_sample_train_bwd(const hr_config& c, const HrTrainArgs& a, int64_t ray, int k, const float* M, hr_acc_t* dM, int lane = 0,
                               int lanes = 1, const HrTrainWindow* win = nullptr)
{
    const int Z = c.z_channels, P = c.preds_per_z, CA = a.ca_total;
    const int64_t s = ray * Z + k, NS = a.n_rays * Z;
    const HrTrainRay q = hr_train_ray(c, a.rays + ray * c.ray_dim);
    const float* hk = a.head + s * P;
    const float ds = a.tape.ds[s], dfeat = a.tape.dfeat[s];
    const float dpre[3] = {a.tape.dpre[s], a.tape.dpre[NS + s], a.tape.dpre[2 * NS + s]};
    float dp[3] = {0.f, 0.f, 0.f};
    if (dfeat != 0.0f || dpre[0] != 0.0f || dpre[1] != 0.0f || dpre[2] != 0.0f) {     // only samples that were valid
        float p[3], dcc;
        hr_sample_point(c, hk, ds, q.ro, q.rd, q.oc, q.time_off, p, &dcc);
        hr_axis_tap_g ax[3];
        for (int i = 0; i < 3; ++i) ax[i] = hr_make_tap_g(hr_normalize_coord(c, p[i], i), c.grid[i]);
        float dpn[3];
        hr_train_gather_bwd(a, ax, q.tap_t, M, dM, CA, dfeat, dpre, dpn, lane, lanes, win);
        for (int i = 0; i < 3; ++i) dp[i] = HR_LANE_SUM(dpn[i], lanes) * c.inv_size[i];
    }
    if (lane != 0) return;
    const float dt = hr_sample_point_bwd(c, hk, ds, q.ro, q.rd, q.oc, q.time_off, dp, a.tape.ddc[s], a.d_head + s * P);
    a.tape.dts[ray * Z + a.tape.src[s]] = dt;
}

// Phase B on the device when phase A left the taps on the tape: the 16 lanes of a sample only do what is per channel -- no
// point, contraction or tap arithmetic repeated by every lane -- and the per-sample tail (hr_sample_point_bwd) is
// hr_sample_train_point_bwd's, one lane per sample.
HR_FN void hr_sample_train_bwd_taps(const hr_config& c, const HrTrainArgs& a, int64_t ray, int k, const float* M, hr_acc_t* dM, int lane, int lanes,
                                    const HrTrainWindow* win)
{
    const int Z = c.z_channels, CA = a.ca_total;
    const int64_t s = ray * Z + k, NS = a.n_rays * Z;
    const float dfeat = a.tape.dfeat[s];
    const float dpre[3] = {a.tape.dpre[s], a.tape.dpre[NS + s], a.tape.dpre[2 * NS + s]};
    float dp[3] = {0.f, 0.f, 0.f};
    if (dfeat != 0.0f || dpre[0] != 0.0f || dpre[1] != 0.0f || dpre[2] != 0.0f) {     // only samples that were valid
        hr_axis_tap_g ax[3];
        for (int i = 0; i < 3; ++i) {
            const float* t = a.tape.taps + (size_t)(6 * i) * NS + s;
            ax[i].t.i0 = __builtin_bit_cast(int, t[0]);
            ax[i].t.i1 = __builtin_bit_cast(int, t[NS]);
            ax[i].t.w0 = t[2 * NS]; ax[i].t.w1 = t[3 * NS];
            ax[i].s0 = t[4 * NS]; ax[i].s1 = t[5 * NS];
            ax[i].mult = 0.5f * (float)(c.grid[i] - 1);
        }
        float base_t = 0.0f;
        if (c.advect) base_t = hr_base_time(c, a.rays[ray * c.ray_dim + c.ray_dim - 1]);
        const hr_axis_tap_g tap_t = hr_make_tap_g(c.video ? hr_normalize_time(c, base_t) : 0.0f, c.video ? c.num_keyframes : 2);
        float dpn[3];
        hr_train_gather_bwd(a, ax, tap_t, M, dM, CA, dfeat, dpre, dpn, lane, lanes, win);
        for (int i = 0; i < 3; ++i) dp[i] = HR_LANE_SUM(dpn[i], lanes) * c.inv_size[i];
    }
    if (lane != 0) return;
    if (win && win->add_dp) { dp[0] += a.tape.dp[s]; dp[1] += a.tape.dp[NS + s]; dp[2] += a.tape.dp[2 * NS + s]; }
    a.tape.dp[s] = dp[0]; a.tape.dp[NS + s] = dp[1]; a.tape.dp[2 * NS + s] = dp[2];
}

// ... and its tail: the sample of sorted rank k of `ray`, one thread
HR_FN void hr_sample_train_point_bwd(const hr_config& c, const HrTrainArgs& a, int64_t ray, int k)
{
    const int Z = c.z_channels, P = c.preds_per_z;
    const int64_t s = ray * Z + k, NS = a.n_rays * Z;
    const HrTrainRay q = hr_train_ray(c, a.rays + ray * c.ray_dim);
    const float dp[3] = {a.tape.dp[s], a.tape.dp[NS + s], a.tape.dp[2 * NS + s]};
    const float dt = hr_sample_point_bwd(c, a.head + s * P, a.tape.ds[s], q.ro, q.rd, q.oc, q.time_off, dp, a.tape.ddc[s], a.d_head + s * P);
    a.tape.dts[ray * Z + a.tape.src[s]] = dt;
}

// Phase C: the sample of ORIGINAL index k of `ray`
HR_FN void hr_sample_train_dist_bwd(const hr_config& c, const HrTrainArgs& a, int64_t ray, int k)
{
    const int64_t s = ray * c.z_channels + k;
    const float* r = a.rays + ray * c.ray_dim;
    const float ro[3] = {r[0] - c.isect_origin[0], r[1] - c.isect_origin[1], r[2] - c.isect_origin[2]};
    const float rd[3] = {r[3], r[4], r[5]};
    hr_sample_distance_bwd(c, a.head + s * c.preds_per_z, k, ro, rd, a.tape.dts[s], a.d_head + s * c.preds_per_z);
}

// Phase A: one ray of a training step (forward, compositing backward).  ZP: compile-time bound on z_channels.
template <int ZP>
HR_FN void hr_ray_train(const hr_config& c, const HrTrainArgs& a, int64_t ray)
{
    const int Z = c.z_channels, P = c.preds_per_z, CA = a.ca_total;
    const float* r = a.rays + ray * c.ray_dim;
    const float* head = a.head + ray * (int64_t)Z * P;
    const float ro[3] = {r[0] - c.isect_origin[0], r[1] - c.isect_origin[1], r[2] - c.isect_origin[2]};
    const float rd[3] = {r[3], r[4], r[5]};
    const float t_ray = r[c.ray_dim - 1];

    // decode matrix of the ray (RGB: basis_mat rows; SH: basis rows folded with the view direction's SH basis)
    float sh[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c.shading == HR_SHADING_SH) hr_sh_deg2(rd[0], rd[1], rd[2], sh);
    float M[3 * HR_TRAIN_MAX_CA];
    for (int cc = 0; cc < 3; ++cc)
        for (int pos = 0; pos < CA; ++pos) M[cc * CA + pos] = hr_train_decode_coef(c, a, sh, cc, pos);

    // ---- forward
    float ds[ZP];                 // sorted pre-contraction distances
    int src[ZP];                  // original sample index of each rank
    for (int k = 0; k < Z; ++k) { ds[k] = hr_sample_distance(c, head + k * P, k, ro, rd); src[k] = k; }
    if (c.sort)
        for (int i = 1; i < Z; ++i) {             // stable insertion sort (sort_z, intersect_utils.py:12-16)
            const float v = ds[i];
            const int s = src[i];
            int j = i - 1;
            while (j >= 0 && ds[j] > v) { ds[j + 1] = ds[j]; src[j + 1] = src[j]; --j; }
            ds[j + 1] = v; src[j + 1] = s;
        }
    float oc[3] = {0.f, 0.f, 0.f};
    if (c.contract_type != HR_CONTRACT_IDENTITY) hr_contract_point(c, ro[0], ro[1], ro[2], oc);
    float base_t = 0.0f, time_off = 0.0f;
    if (c.advect) { base_t = hr_base_time(c, t_ray); time_off = t_ray - base_t; }
    const hr_axis_tap_g tap_t = hr_make_tap_g(c.video ? hr_normalize_time(c, base_t) : 0.0f, c.video ? c.num_keyframes : 2);

    float dc[ZP], feat[ZP], pre[ZP][3], trans[ZP], alpha[ZP], wgt[ZP];
    bool valid[ZP];
    for (int k = 0; k < Z; ++k) {
        float p[3];
        hr_sample_point(c, head + k * P, ds[k], ro, rd, oc, time_off, p, &dc[k]);
        valid[k] = hr_sample_valid(c, p, dc[k]);
        feat[k] = 0.0f; pre[k][0] = 0.0f; pre[k][1] = 0.0f; pre[k][2] = 0.0f;
        if (valid[k]) {
            hr_axis_tap_g ax[3];
            for (int i = 0; i < 3; ++i) ax[i] = hr_make_tap_g(hr_normalize_coord(c, p[i], i), c.grid[i]);
            hr_train_gather(a, ax, tap_t, M, CA, &feat[k], pre[k]);
        }
    }
    float T = 1.0f, acc_w = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    for (int k = 0; k < Z; ++k) {
        const float delta = (k == Z - 1) ? 1e10f : (dc[k + 1] - dc[k]);
        const float sigma = valid[k] ? hr_density(c, feat[k]) : 0.0f;
        alpha[k] = 1.0f - expf(-sigma * (delta * c.distance_scale));
        trans[k] = T;
        wgt[k] = alpha[k] * T;
        T = T * ((1.0f - alpha[k]) + 1e-10f);
        float rr[3] = {0.f, 0.f, 0.f};
        if (wgt[k] > c.weight_thresh)
            for (int i = 0; i < 3; ++i)
                rr[i] = (c.shading == HR_SHADING_SH) ? fmaxf(pre[k][i] + 0.5f, 0.0f) : 1.0f / (1.0f + expf(-pre[k][i]));
        if (c.f_color_scale.offset >= 0) {
            const float* hk = head + k * P;
            for (int i = 0; i < 3; ++i)
                rr[i] = rr[i] * (hr_apply_act(c.f_color_scale.act, hk[c.f_color_scale.offset + i]) + 1.0f) +
                        hr_apply_act(c.f_color_shift.act, hk[c.f_color_shift.offset + i]);
        }
        c0 += wgt[k] * rr[0]; c1 += wgt[k] * rr[1]; c2 += wgt[k] * rr[2];
        acc_w += wgt[k];
    }
    if (a.white_bg) { const float bg = 1.0f - acc_w; c0 += bg; c1 += bg; c2 += bg; }
    const float cpre[3] = {c0, c1, c2};          // the composited colour before the per-ray scale / shift
    float gscale[3] = {1.0f, 1.0f, 1.0f};
    const bool head_transform = c.f_color_scale_global.offset >= 0 && c.f_color_scale_global.channels == 9;
    float thead[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (head_transform) {                        // transform_color_one, the matrix from the head (`color_transform_global`)
        const hr_head_field& fs = c.f_color_scale_global;
        const hr_head_field& fh = c.f_color_shift_global;
        for (int i = 0; i < 9; ++i) thead[i] = hr_apply_act(fs.act, head[fs.offset + i]);
        const float n0 = c0 + ((c0 * thead[0] + c1 * thead[1]) + c2 * thead[2]);
        const float n1 = c1 + ((c0 * thead[3] + c1 * thead[4]) + c2 * thead[5]);
        const float n2 = c2 + ((c0 * thead[6] + c1 * thead[7]) + c2 * thead[8]);
        c0 = n0 + hr_apply_act(fh.act, head[fh.offset + 0]);
        c1 = n1 + hr_apply_act(fh.act, head[fh.offset + 1]);
        c2 = n2 + hr_apply_act(fh.act, head[fh.offset + 2]);
    }
    else if (c.f_color_scale_global.offset >= 0) {    // scale_shift_color_one (tensorf_utils.py:275-281): sample 0's head values
        const hr_head_field& fs = c.f_color_scale_global;
        const hr_head_field& fh = c.f_color_shift_global;
        for (int i = 0; i < 3; ++i) gscale[i] = hr_apply_act(fs.act, head[fs.offset + i]) + 1.0f;
        c0 = c0 * gscale[0] + hr_apply_act(fh.act, head[fh.offset + 0]);
        c1 = c1 * gscale[1] + hr_apply_act(fh.act, head[fh.offset + 1]);
        c2 = c2 * gscale[2] + hr_apply_act(fh.act, head[fh.offset + 2]);
    }
    else if (a.color_table) {                    // transform_color_one (tensorf_utils.py:308-320, point.py:588-594)
        int id = (int)rintf(r[c.ray_dim - 2]);
        id = id < 0 ? 0 : (id > c.color_table_views - 1 ? c.color_table_views - 1 : id);
        const float* e = a.color_table + 12 * id;
        float t[9];
        for (int i = 0; i < 9; ++i) t[i] = hr_apply_act(c.color_table_t_act, e[i]);
        const float n0 = c0 + ((c0 * t[0] + c1 * t[1]) + c2 * t[2]);
        const float n1 = c1 + ((c0 * t[3] + c1 * t[4]) + c2 * t[5]);
        const float n2 = c2 + ((c0 * t[6] + c1 * t[7]) + c2 * t[8]);
        c0 = n0 + hr_apply_act(c.color_table_s_act, e[9]);
        c1 = n1 + hr_apply_act(c.color_table_s_act, e[10]);
        c2 = n2 + hr_apply_act(c.color_table_s_act, e[11]);
    }
    if (a.rgb) { a.rgb[ray * 3 + 0] = c0; a.rgb[ray * 3 + 1] = c1; a.rgb[ray * 3 + 2] = c2; }
    if (!a.d_rgb) return;

    // ---- backward
    float g[3] = {a.d_rgb[ray * 3 + 0], a.d_rgb[ray * 3 + 1], a.d_rgb[ray * 3 + 2]};
    float* dhead = a.d_head + ray * (int64_t)Z * P;
    for (int i = 0; i < Z * P; ++i) dhead[i] = 0.0f;
    if (head_transform) {
        const hr_head_field& fs = c.f_color_scale_global;
        const hr_head_field& fh = c.f_color_shift_global;
        float gn[3] = {g[0], g[1], g[2]};
        for (int i = 0; i < 3; ++i) {
            dhead[fh.offset + i] += g[i] * hr_act_grad(fh.act, head[fh.offset + i]);
            for (int j = 0; j < 3; ++j) {
                dhead[fs.offset + 3 * i + j] += g[i] * cpre[j] * hr_act_grad(fs.act, head[fs.offset + 3 * i + j]);
                gn[j] += g[i] * thead[3 * i + j];
            }
        }
        g[0] = gn[0]; g[1] = gn[1]; g[2] = gn[2];
    }
    else if (c.f_color_scale_global.offset >= 0) {
        const hr_head_field& fs = c.f_color_scale_global;
        const hr_head_field& fh = c.f_color_shift_global;
        for (int i = 0; i < 3; ++i) {
            dhead[fs.offset + i] += g[i] * cpre[i] * hr_act_grad(fs.act, head[fs.offset + i]);
            dhead[fh.offset + i] += g[i] * hr_act_grad(fh.act, head[fh.offset + i]);
            g[i] = g[i] * gscale[i];             // everything below sees the gradient of the un-scaled colour
        }
    }
    else if (a.color_table) {
        int id = (int)rintf(r[c.ray_dim - 2]);
        id = id < 0 ? 0 : (id > c.color_table_views - 1 ? c.color_table_views - 1 : id);
        const float* e = a.color_table + 12 * id;
        hr_acc_t* de = a.d_color_table + 12 * id;
        float gn[3] = {g[0], g[1], g[2]};
        for (int i = 0; i < 3; ++i) {
            HR_ATOMIC_ADD(de + 9 + i, g[i] * hr_act_grad(c.color_table_s_act, e[9 + i]));
            for (int j = 0; j < 3; ++j) {
                HR_ATOMIC_ADD(de + 3 * i + j, g[i] * cpre[j] * hr_act_grad(c.color_table_t_act, e[3 * i + j]));
                gn[j] += g[i] * hr_apply_act(c.color_table_t_act, e[3 * i + j]);
            }
        }
        g[0] = gn[0]; g[1] = gn[1]; g[2] = gn[2];
    }
    const float gsum = a.white_bg ? (g[0] + g[1] + g[2]) : 0.0f;
    float ddc[ZP];                // dL / d final distance
    float dfeat[ZP];
    for (int k = 0; k < Z; ++k) ddc[k] = 0.0f;
    float S = 0.0f;               // sum over later samples of dw_j * w_j
    for (int k = Z - 1; k >= 0; --k) {
        // colour of the sample again (cheap) for dw = g . rgb_k - [white] sum(g)
        float rr[3] = {0.f, 0.f, 0.f}, raw[3] = {0.f, 0.f, 0.f};
        const bool app = wgt[k] > c.weight_thresh;
        if (app)
            for (int i = 0; i < 3; ++i)
                raw[i] = (c.shading == HR_SHADING_SH) ? fmaxf(pre[k][i] + 0.5f, 0.0f) : 1.0f / (1.0f + expf(-pre[k][i]));
        const float* hk = head + k * P;
        float* dhk = dhead + k * P;
        float dpre[3];
        for (int i = 0; i < 3; ++i) {
            float sc = 1.0f;
            rr[i] = raw[i];
            const float dr = wgt[k] * g[i];
            if (c.f_color_scale.offset >= 0) {
                const float hs = hk[c.f_color_scale.offset + i], hh = hk[c.f_color_shift.offset + i];
                sc = hr_apply_act(c.f_color_scale.act, hs) + 1.0f;
                rr[i] = raw[i] * sc + hr_apply_act(c.f_color_shift.act, hh);
                dhk[c.f_color_scale.offset + i] += dr * raw[i] * hr_act_grad(c.f_color_scale.act, hs);
                dhk[c.f_color_shift.offset + i] += dr * hr_act_grad(c.f_color_shift.act, hh);
            }
            const float draw = dr * sc;
            if (!app) dpre[i] = 0.0f;
            else if (c.shading == HR_SHADING_SH) dpre[i] = (pre[k][i] + 0.5f > 0.0f) ? draw : 0.0f;
            else dpre[i] = draw * raw[i] * (1.0f - raw[i]);
        }
        pre[k][0] = dpre[0]; pre[k][1] = dpre[1]; pre[k][2] = dpre[2];          // reuse the storage for dL/d pre
        const float dw = (g[0] * rr[0] + g[1] * rr[1] + g[2] * rr[2]) - gsum;
        const float inc = (1.0f - alpha[k]) + 1e-10f;
        const float dalpha = dw * trans[k] - S / inc;
        S += dw * wgt[k];
        // alpha = 1 - exp(-sigma * delta * scale)
        const float delta = (k == Z - 1) ? 1e10f : (dc[k + 1] - dc[k]);
        const float sigma = valid[k] ? hr_density(c, feat[k]) : 0.0f;
        const float e = 1.0f - alpha[k];
        const float dsigma = dalpha * e * (delta * c.distance_scale);
        dfeat[k] = valid[k] ? dsigma * hr_density_grad(c, feat[k]) : 0.0f;
        if (k < Z - 1) {
            const float ddelta = dalpha * e * sigma * c.distance_scale;
            ddc[k + 1] += ddelta;
            ddc[k] -= ddelta;
        }
    }
    // hand the per-sample upstream gradients to phases B and C
    const int64_t NS = a.n_rays * Z;
    for (int k = 0; k < Z; ++k) {
        const int64_t s = ray * Z + k;
        a.tape.ds[s] = ds[k];
        a.tape.src[s] = src[k];
        a.tape.dfeat[s] = dfeat[k];
        a.tape.dpre[s] = pre[k][0]; a.tape.dpre[NS + s] = pre[k][1]; a.tape.dpre[2 * NS + s] = pre[k][2];
        a.tape.ddc[s] = ddc[k];
    }
}

// ---------------------------------------------------------------- coarse level of a point_prediction cascade
// PointPredictionEmbedding (nlf/embedding/point.py:137-203) feeds its MLP one row per (ray, coarse sample): the sample's
// point after the first intersect (sorted, contracted, advected / offset) next to ray constants.  Forward: those rows;
// backward: dL/d rows -> dL/d (raw head of the ray MLP), through hr_sample_point_bwd and hr_sample_distance_bwd -- no
// gather and no compositing at this level.  The fine level is the ordinary sample stage: its head, one row of M samples
// per coarse point, is laid out exactly like (n_rays, Z * P).
struct HrRowsArgs {
    const hr_config* cfg_dev;   // the COARSE level's configuration, caller's column order
    const float* rays;
    const float* head;          // (n, Zc * Pc) raw output of the ray MLP
    int64_t n_rays;
    float* rows;                // (n * Zc, row_dim), written by the forward
    const float* d_rows;        // NULL: forward only
    float* d_head;              // (n, Zc * Pc), written by the backward
    int row_dim, n_inputs;
    int kind[4], len[4];        // HR_PIN_* and columns of each input of the row
    HrTrainTape tape;           // ds, src, dts (n * Zc each)
};

template <int ZP>
HR_FN void hr_ray_rows(const hr_config& c, const HrRowsArgs& a, int64_t ray)
{
    const int Z = c.z_channels, P = c.preds_per_z;
    const float* r = a.rays + ray * c.ray_dim;
    const float* head = a.head + ray * (int64_t)Z * P;
    const HrTrainRay q = hr_train_ray(c, r);
    float ds[ZP];
    int src[ZP];
    for (int k = 0; k < Z; ++k) { ds[k] = hr_sample_distance(c, head + k * P, k, q.ro, q.rd); src[k] = k; }
    if (c.sort)
        for (int i = 1; i < Z; ++i) {
            const float v = ds[i];
            const int s = src[i];
            int j = i - 1;
            while (j >= 0 && ds[j] > v) { ds[j + 1] = ds[j]; src[j + 1] = src[j]; --j; }
            ds[j + 1] = v; src[j + 1] = s;
        }
    for (int k = 0; k < Z; ++k) {
        float p[3], dc;
        hr_sample_point(c, head + k * P, ds[k], q.ro, q.rd, q.oc, q.time_off, p, &dc);
        float* row = a.rows + (ray * Z + k) * a.row_dim;
        int col = 0;
        for (int i = 0; i < a.n_inputs; ++i)
            for (int j = 0; j < a.len[i]; ++j)
                row[col++] = (a.kind[i] == HR_PIN_POINTS) ? p[j] : (a.kind[i] == HR_PIN_VIEWDIRS) ? r[3 + j]
                             : (a.kind[i] == HR_PIN_ORIGINS) ? r[j] : r[c.ray_dim - 1];
    }
    if (!a.d_rows) return;
    float* dhead = a.d_head + ray * (int64_t)Z * P;
    for (int i = 0; i < Z * P; ++i) dhead[i] = 0.0f;
    for (int k = 0; k < Z; ++k) { a.tape.ds[ray * Z + k] = ds[k]; a.tape.src[ray * Z + k] = src[k]; }
}

// backward of one coarse sample (sorted rank k): the point columns of its row
HR_FN void hr_sample_rows_bwd(const hr_config& c, const HrRowsArgs& a, int64_t ray, int k)
{
    const int Z = c.z_channels, P = c.preds_per_z;
    const int64_t s = ray * Z + k;
    const HrTrainRay q = hr_train_ray(c, a.rays + ray * c.ray_dim);
    const float* drow = a.d_rows + s * a.row_dim;
    float dp[3] = {0.f, 0.f, 0.f};
    int col = 0;
    for (int i = 0; i < a.n_inputs; ++i)
        for (int j = 0; j < a.len[i]; ++j, ++col)
            if (a.kind[i] == HR_PIN_POINTS) dp[j] += drow[col];
    const float dt = hr_sample_point_bwd(c, a.head + s * P, a.tape.ds[s], q.ro, q.rd, q.oc, q.time_off, dp, 0.0f, a.d_head + s * P);
    a.tape.dts[ray * Z + a.tape.src[s]] = dt;
}

// backward of one coarse sample (ORIGINAL index k): intersection and head activations
HR_FN void hr_sample_rows_dist_bwd(const hr_config& c, const HrRowsArgs& a, int64_t ray, int k)
{
    const int64_t s = ray * c.z_channels + k;
    const float* r = a.rays + ray * c.ray_dim;
    const float ro[3] = {r[0] - c.isect_origin[0], r[1] - c.isect_origin[1], r[2] - c.isect_origin[2]};
    const float rd[3] = {r[3], r[4], r[5]};
    hr_sample_distance_bwd(c, a.head + s * c.preds_per_z, k, ro, rd, a.tape.dts[s], a.d_head + s * c.preds_per_z);
}

#endif  // HR_TRAIN_H
