// Occupancy ("alpha mask") of the feature grids, the grid-management step of the reference's training loop
// (SURVEY 8f-3): TensorBase.getDenseAlpha / compute_alpha (nlf/nets/tensorf_base.py:381-401, 489-507),
// TensorVMKeyframeTime.getDenseAlpha / compute_alpha (nlf/nets/tensorf_dynamic.py:499-536, 618-643) and
// AlphaGridMask.sample_alpha (utils/tensorf_utils.py:459-484).  One function of one grid point, compiled for the device
// (one thread per point) and for the host by the CPU test-suite.
#ifndef HR_MASK_H
#define HR_MASK_H

#include "hr_grid.h"
#include "hr_math.h"

struct HrMaskArgs {
    const hr_config* cfg_dev;
    HrGridPlane planes[3];       // packed parameter values (fp32 texels)
    int n[3];                    // points per axis: dense_xyz = aabb0 * (1 - s) + aabb1 * s, s = linspace(0, 1, n)
    float length;                // alpha = 1 - exp(-sigma * length)  (0.01 in updateAlphaMask)
    int num_frames;              // keyframe nets: frames of the sequence (train_dataset.num_frames)
    const float* prev_volume;    // previous mask (D = pn[2], H = pn[1], W = pn[0]) or NULL: points it rejects get sigma = 0
    int pn[3];
    float prev_aabb[6];
    float* alpha;                // (n[0], n[1], n[2]), x slowest -- the layout getDenseAlpha returns
};

// torch.linspace(0, 1, n)[i] in float32 (two-sided evaluation around the midpoint, as ATen does)
HR_FN float hr_linspace01(int i, int n)
{
    if (n == 1) return 0.0f;
    const float step = 1.0f / (float)(n - 1);
    return (i < n / 2) ? step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

// F.grid_sample of a (1, 1, D, H, W) volume at one normalised point (x -> W, y -> H, z -> D), bilinear,
// align_corners=True, zeros padding
HR_FN float hr_volume_sample(const float* vol, int W, int H, int D, float gx, float gy, float gz)
{
    const hr_axis_tap tx = hr_make_tap(gx, W), ty = hr_make_tap(gy, H), tz = hr_make_tap(gz, D);
    float acc = 0.0f;
    for (int c = 0; c < 8; ++c) {
        const int ix = (c & 1) ? tx.i1 : tx.i0, iy = (c & 2) ? ty.i1 : ty.i0, iz = (c & 4) ? tz.i1 : tz.i0;
        const float w = ((c & 1) ? tx.w1 : tx.w0) * ((c & 2) ? ty.w1 : ty.w0) * ((c & 4) ? tz.w1 : tz.w0);
        acc += vol[((size_t)iz * H + iy) * W + ix] * w;
    }
    return acc;
}

// density feature of one point: sum over the plane pairs and density channels of plane x (line | time plane)
HR_FN float hr_density_feature(const HrGridPlane* planes, const hr_axis_tap* ax, const hr_axis_tap& at)
{
    float s = 0.0f;
    for (int j = 0; j < 3; ++j) {
        const HrGridPlane& g = planes[j];
        if (g.cd4 == 0) continue;
        const hr_axis_tap tx = ax[(j == 2) ? 1 : 0], ty = ax[(j == 0) ? 1 : 2], bx = ax[2 - j];
        const float* A = reinterpret_cast<const float*>(g.a);
        const float* B = reinterpret_cast<const float*>(g.b);
        const size_t a00 = (size_t)(ty.i0 * g.aw + tx.i0) * g.tex, a01 = (size_t)(ty.i0 * g.aw + tx.i1) * g.tex;
        const size_t a10 = (size_t)(ty.i1 * g.aw + tx.i0) * g.tex, a11 = (size_t)(ty.i1 * g.aw + tx.i1) * g.tex;
        const bool line = (g.bw == 1);
        for (int ch = 0; ch < 4 * g.cd4; ++ch) {
            float pa = A[a00 + ch] * (tx.w0 * ty.w0);
            pa = fmaf(A[a01 + ch], tx.w1 * ty.w0, pa);
            pa = fmaf(A[a10 + ch], tx.w0 * ty.w1, pa);
            pa = fmaf(A[a11 + ch], tx.w1 * ty.w1, pa);
            float pb;
            if (line) {
                pb = fmaf(B[(size_t)bx.i1 * g.tex + ch], bx.w1, B[(size_t)bx.i0 * g.tex + ch] * bx.w0);
            } else {
                pb = B[(size_t)(at.i0 * g.bw + bx.i0) * g.tex + ch] * (bx.w0 * at.w0);
                pb = fmaf(B[(size_t)(at.i0 * g.bw + bx.i1) * g.tex + ch], bx.w1 * at.w0, pb);
                pb = fmaf(B[(size_t)(at.i1 * g.bw + bx.i0) * g.tex + ch], bx.w0 * at.w1, pb);
                pb = fmaf(B[(size_t)(at.i1 * g.bw + bx.i1) * g.tex + ch], bx.w1 * at.w1, pb);
            }
            s = s + pa * pb;
        }
    }
    return s;
}

// alpha of grid point (ix, iy, iz); keyframe nets: the maximum over the frames of the sequence (tensorf_dynamic.py:515-534)
HR_FN float hr_point_alpha(const hr_config& c, const HrMaskArgs& a, int ix, int iy, int iz)
{
    const int idx[3] = {ix, iy, iz};
    float p[3];
    for (int i = 0; i < 3; ++i) {
        const float s = hr_linspace01(idx[i], a.n[i]);
        p[i] = c.aabb[i] * (1.0f - s) + c.aabb[3 + i] * s;
    }
    // static nets only: TensorBase.compute_alpha asks the previous mask first (tensorf_base.py:491-503);
    // TensorVMKeyframeTime.compute_alpha (tensorf_dynamic.py:618-643) never reads alphaMask
    if (a.prev_volume && !c.video) {
        float g[3];
        for (int i = 0; i < 3; ++i)
            g[i] = (p[i] - a.prev_aabb[i]) * (1.0f / (a.prev_aabb[3 + i] - a.prev_aabb[i]) * 2.0f) - 1.0f;   // AlphaGridMask.normalize_coord
        if (!(hr_volume_sample(a.prev_volume, a.pn[0], a.pn[1], a.pn[2], g[0], g[1], g[2]) > 0.0f)) return 0.0f;
    }
    hr_axis_tap ax[3];
    for (int i = 0; i < 3; ++i) ax[i] = hr_make_tap(hr_normalize_coord(c, p[i], i), c.grid[i]);
    if (!c.video) {
        const hr_axis_tap at = hr_make_tap(0.0f, 2);
        return 1.0f - expf(-hr_density(c, hr_density_feature(a.planes, ax, at)) * a.length);
    }
    float best = 0.0f;
    const int F = a.num_frames;
    // time_scale_factor (tensorf_dynamic.py:513) and its reciprocal are python floats that meet float32 tensors
    const double tsf_d = (double)(F - 1) / (double)F;
    const float tsf = (float)tsf_d, inv_tsf = (float)(1.0 / tsf_d);
    for (int f = 0; f < F; ++f) {
        // np.linspace(0, 1, F)[f] in float64, then `ones * t` in float32
        const float t = (F == 1) ? 0.0f : ((f == F - 1) ? 1.0f : (float)((double)f * (1.0 / (double)(F - 1))));
        const float base = rintf(fminf(fmaxf(t * tsf, 0.0f), (float)(c.num_keyframes - 1))) * inv_tsf;
        const hr_axis_tap at = hr_make_tap(hr_normalize_time(c, base), c.num_keyframes);
        best = fmaxf(best, 1.0f - expf(-hr_density(c, hr_density_feature(a.planes, ax, at)) * a.length));
    }
    return best;
}

#endif  // HR_MASK_H
