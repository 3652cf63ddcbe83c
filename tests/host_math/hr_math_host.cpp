// TEST-ONLY host build of hyperreel_amd/csrc/hr_math.h (the per-ray / per-sample formulas
// the HIP kernels call), so the CPU suite can compare them with the oracle without a GPU.
// Nothing in the product links or loads this file.
#include <algorithm>
#include <vector>

#include "../../hyperreel_amd/csrc/hr_math.h"

extern "C" {

int hm_sizeof_config() { return (int)sizeof(hr_config); }

// rays (n, ray_dim) -> out (n, mlp_in)
void hm_features(const hr_config* c, const float* rays, int n, float* out)
{
    for (int i = 0; i < n; ++i) hr_ray_features(*c, rays + (size_t)i * c->ray_dim, out + (size_t)i * c->mlp_in);
}

// Emulates the order of operations of hr_sample_kernel up to the colour net, with the
// cross-lane steps (sort, neighbour delta) done serially:
// rays (n, ray_dim), head (n, Z*P) -> dist (n,Z), points (n,Z,3), valid (n,Z), base_t (n)
void hm_embed(const hr_config* c, const float* rays, const float* head, int n, float* dist_out, float* points_out,
              int* valid_out, float* base_t_out)
{
    const int Z = c->z_channels, P = c->preds_per_z;
    std::vector<float> d(Z);
    for (int i = 0; i < n; ++i) {
        const float* r = rays + (size_t)i * c->ray_dim;
        float ro[3] = {r[0] - c->isect_origin[0], r[1] - c->isect_origin[1], r[2] - c->isect_origin[2]};
        float rd[3] = {r[3], r[4], r[5]};
        const float* h = head + (size_t)i * Z * P;
        for (int k = 0; k < Z; ++k) d[k] = hr_sample_distance(*c, h + k * P, k, ro, rd);
        if (c->sort) std::sort(d.begin(), d.end());
        float oc[3] = {0, 0, 0};
        if (c->contract_type != HR_CONTRACT_IDENTITY) hr_contract_point(*c, ro[0], ro[1], ro[2], oc);
        float t = r[c->ray_dim - 1], base_t = 0.f, toff = 0.f;
        if (c->advect) { base_t = hr_base_time(*c, t); toff = t - base_t; }
        base_t_out[i] = base_t;
        for (int k = 0; k < Z; ++k) {
            float p[3], dc;
            hr_sample_point(*c, h + k * P, d[k], ro, rd, oc, toff, p, &dc);
            dist_out[(size_t)i * Z + k] = dc;
            for (int a = 0; a < 3; ++a) points_out[((size_t)i * Z + k) * 3 + a] = p[a];
            valid_out[(size_t)i * Z + k] = hr_sample_valid(*c, p, dc) ? 1 : 0;
        }
    }
}

// hr_sample_distance with the quadratic's ray terms handed in (what the sample kernel does once per ray: sample_core.inc, hr_ray_constants)
// and without: dist_pre / dist_own (n, Z)
void hm_distance_both(const hr_config* c, const float* rays, const float* head, int n, float* dist_pre, float* dist_own)
{
    const int Z = c->z_channels, P = c->preds_per_z;
    for (int i = 0; i < n; ++i) {
        const float* r = rays + (size_t)i * c->ray_dim;
        float ro[3] = {r[0] - c->isect_origin[0], r[1] - c->isect_origin[1], r[2] - c->isect_origin[2]};
        float rd[3] = {r[3], r[4], r[5]};
        float quad[3] = {0.0f, 0.0f, 0.0f};
        if ((c->isect_type == HR_ISECT_SPHERE || c->isect_type == HR_ISECT_CYLINDER) && c->origin_scale == 0.0f)
            hr_quadratic_ray_terms(*c, ro, rd, c->origin_initial[0], c->origin_initial[1], c->origin_initial[2], quad);
        const float* h = head + (size_t)i * Z * P;
        for (int k = 0; k < Z; ++k) {
            dist_pre[(size_t)i * Z + k] = hr_sample_distance(*c, h + k * P, k, ro, rd, nullptr, quad);
            dist_own[(size_t)i * Z + k] = hr_sample_distance(*c, h + k * P, k, ro, rd);
        }
    }
}

// one axis of grid_sample: g (n) on an axis of `size` texels -> i0,i1,w0,w1
void hm_taps(const float* g, int n, int size, int* i0, int* i1, float* w0, float* w1)
{
    for (int i = 0; i < n; ++i) {
        hr_axis_tap t = hr_make_tap(g[i], size);
        i0[i] = t.i0; i1[i] = t.i1; w0[i] = t.w0; w1[i] = t.w1;
    }
}

// the clamped form of the same taps (base, base + 1)
void hm_taps_c(const float* g, int n, int size, int* i0, float* w0, float* w1)
{
    for (int i = 0; i < n; ++i) {
        hr_axis_tap_c t = hr_make_tap_c(g[i], size);
        i0[i] = t.i0; w0[i] = t.w0; w1[i] = t.w1;
    }
}

// the in-range form the render kernels use for coordinates of valid samples
void hm_taps_in(const float* g, int n, int size, int* i0, float* w0, float* w1)
{
    for (int i = 0; i < n; ++i) {
        hr_axis_tap_c t = hr_make_tap_in(g[i], size);
        i0[i] = t.i0; w0[i] = t.w0; w1[i] = t.w1;
    }
}

void hm_sh(const float* d, int n, float* out)
{
    for (int i = 0; i < n; ++i) hr_sh_deg2(d[3 * i], d[3 * i + 1], d[3 * i + 2], out + 9 * i);
}

void hm_density(const hr_config* c, const float* f, int n, float* out)
{
    for (int i = 0; i < n; ++i) out[i] = hr_density(*c, f[i]);
}

void hm_normalize(const hr_config* c, const float* p, int n, float* out)
{
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) out[3 * i + a] = hr_normalize_coord(*c, p[3 * i + a], a);
}

float hm_normalize_time(const hr_config* c, float t) { return hr_normalize_time(*c, t); }

// contraction of points / inverse contraction of distances (nlf/contract.py)
void hm_contract_points(const hr_config* c, const float* p, int n, float* out)
{
    for (int i = 0; i < n; ++i) hr_contract_point(*c, p[3 * i], p[3 * i + 1], p[3 * i + 2], out + 3 * i);
}

void hm_inverse_contract_distance(const hr_config* c, const float* d, int n, float* out)
{
    for (int i = 0; i < n; ++i) out[i] = hr_inverse_contract_distance(*c, d[i]);
}

// display pack: to8b and the viewer's transpose / flip index map
void hm_to8b(const float* x, int n, unsigned char* out)
{
    for (int i = 0; i < n; ++i) out[i] = hr_to8b(x[i]);
}

void hm_display_map(int h, int w, int transpose, int flip, long long* src)
{
    const int oh = transpose ? w : h, ow = transpose ? h : w;
    for (int y = 0; y < oh; ++y)
        for (int x = 0; x < ow; ++x) src[(long long)y * ow + x] = hr_display_src_pixel(y, x, h, w, transpose, flip);
}

}  // extern "C"
