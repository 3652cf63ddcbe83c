// TEST-ONLY host build of hyperreel_amd/csrc/hr_train.h (the per-ray forward + backward of the training path), so the
// CPU suite can compare it with torch.autograd on the CPU restatement of the reference without a GPU.
// Nothing in the product links or loads this file.
#include <vector>

#include "../../hyperreel_amd/csrc/hr_mask.h"
#include "../../hyperreel_amd/csrc/hr_train.h"

extern "C" {

int ht_sizeof_plane() { return (int)sizeof(HrGridPlane); }

const char* ht_unsupported(const hr_config* c) { return hr_train_unsupported(*c); }

// planes: 3 descriptors with a / b pointing at packed fp32 texels; g_a / g_b: packed gradient accumulators (zeroed by
// the caller); d_rgb == NULL runs the forward only.  Returns 0, or -1 when z_channels / ca_total exceed the bounds.
int ht_train(const hr_config* c, const float* rays, const float* head, long long n, const float* d_rgb, float* rgb, float* d_head,
             const HrGridPlane* planes, float** g_a, float** g_b, const float* basis, float* d_basis, int n_basis_cols, int ca_total,
             int white_bg, const float* color_table, float* d_color_table)
{
    if (c->z_channels > 256 || ca_total > HR_TRAIN_MAX_CA) return -1;
    HrTrainArgs a = {};
    a.cfg_dev = c;
    a.rays = rays; a.head = head; a.n_rays = n; a.rgb = rgb; a.d_rgb = d_rgb; a.d_head = d_head;
    for (int j = 0; j < 3; ++j) { a.planes[j] = planes[j]; a.g_a[j] = g_a[j]; a.g_b[j] = g_b[j]; }
    a.basis = basis; a.d_basis = d_basis; a.n_basis_cols = n_basis_cols; a.ca_total = ca_total; a.white_bg = white_bg;
    a.color_table = color_table; a.d_color_table = d_color_table;
    // the tape between the phases (the device keeps it in a workspace of the model)
    const size_t NS = (size_t)n * c->z_channels;
    std::vector<float> ds(NS), dfeat(NS), dpre(3 * NS), ddc(NS), dts(NS);
    std::vector<int> src(NS);
    a.tape.ds = ds.data(); a.tape.src = src.data(); a.tape.dfeat = dfeat.data(); a.tape.dpre = dpre.data();
    a.tape.ddc = ddc.data(); a.tape.dts = dts.data();
    int ZP = 8;
    while (ZP < c->z_channels) ZP <<= 1;
    for (long long i = 0; i < n; ++i) {
        switch (ZP) {
            case 8: hr_ray_train<8>(*c, a, i); break;
            case 16: hr_ray_train<16>(*c, a, i); break;
            case 32: hr_ray_train<32>(*c, a, i); break;
            case 64: hr_ray_train<64>(*c, a, i); break;
            case 128: hr_ray_train<128>(*c, a, i); break;
            default: hr_ray_train<256>(*c, a, i); break;
        }
    }
    if (!d_rgb) return 0;
    for (long long i = 0; i < n; ++i) {                     // phase B, the way a workgroup of the device does it per ray
        const float* r = rays + (size_t)i * c->ray_dim;
        float sh[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (c->shading == HR_SHADING_SH) hr_sh_deg2(r[3], r[4], r[5], sh);
        float M[3 * HR_TRAIN_MAX_CA], dM[3 * HR_TRAIN_MAX_CA];
        for (int cc = 0; cc < 3; ++cc)
            for (int pos = 0; pos < ca_total; ++pos) { M[cc * ca_total + pos] = hr_train_decode_coef(*c, a, sh, cc, pos); dM[cc * ca_total + pos] = 0.0f; }
        for (int k = 0; k < c->z_channels; ++k) hr_sample_train_bwd(*c, a, i, k, M, dM);
        for (int cc = 0; cc < 3; ++cc)
            for (int pos = 0; pos < ca_total; ++pos) hr_train_fold_basis(*c, a, sh, cc, pos, dM[cc * ca_total + pos]);
    }
    for (long long i = 0; i < n; ++i)                       // phase C
        for (int k = 0; k < c->z_channels; ++k) hr_sample_train_dist_bwd(*c, a, i, k);
    return 0;
}

// dense alpha of the grid (hr_mask.h) the way the device does it, one point after the other
void ht_dense_alpha(const hr_config* c, const HrGridPlane* planes, const int* n, float length, int num_frames, const float* prev_volume,
                    const int* pn, const float* prev_aabb, float* alpha)
{
    HrMaskArgs a = {};
    a.cfg_dev = c;
    for (int j = 0; j < 3; ++j) { a.planes[j] = planes[j]; a.n[j] = n[j]; a.pn[j] = pn ? pn[j] : 0; }
    for (int j = 0; j < 6; ++j) a.prev_aabb[j] = prev_aabb ? prev_aabb[j] : 0.0f;
    a.length = length; a.num_frames = num_frames; a.prev_volume = prev_volume; a.alpha = alpha;
    for (int x = 0; x < n[0]; ++x)
        for (int y = 0; y < n[1]; ++y)
            for (int z = 0; z < n[2]; ++z) alpha[((size_t)x * n[1] + y) * n[2] + z] = hr_point_alpha(*c, a, x, y, z);
}

// coarse level of a cascade: rows forward and (d_rows != NULL) backward
int ht_rows(const hr_config* c, const float* rays, const float* head, long long n, const float* d_rows, float* rows, float* d_head, int row_dim,
            int n_inputs, const int* kind, const int* len)
{
    if (c->z_channels > 256 || n_inputs > 4) return -1;
    HrRowsArgs a = {};
    a.cfg_dev = c; a.rays = rays; a.head = head; a.n_rays = n; a.rows = rows; a.d_rows = d_rows; a.d_head = d_head;
    a.row_dim = row_dim; a.n_inputs = n_inputs;
    for (int i = 0; i < n_inputs; ++i) { a.kind[i] = kind[i]; a.len[i] = len[i]; }
    const size_t NS = (size_t)n * c->z_channels;
    std::vector<float> ds(NS), dts(NS);
    std::vector<int> src(NS);
    a.tape.ds = ds.data(); a.tape.src = src.data(); a.tape.dts = dts.data();
    int ZP = 8;
    while (ZP < c->z_channels) ZP <<= 1;
    for (long long i = 0; i < n; ++i) {
        switch (ZP) {
            case 8: hr_ray_rows<8>(*c, a, i); break;
            case 16: hr_ray_rows<16>(*c, a, i); break;
            case 32: hr_ray_rows<32>(*c, a, i); break;
            case 64: hr_ray_rows<64>(*c, a, i); break;
            case 128: hr_ray_rows<128>(*c, a, i); break;
            default: hr_ray_rows<256>(*c, a, i); break;
        }
    }
    if (!d_rows) return 0;
    for (long long i = 0; i < n; ++i)
        for (int k = 0; k < c->z_channels; ++k) hr_sample_rows_bwd(*c, a, i, k);
    for (long long i = 0; i < n; ++i)
        for (int k = 0; k < c->z_channels; ++k) hr_sample_rows_dist_bwd(*c, a, i, k);
    return 0;
}

}  // extern "C"
