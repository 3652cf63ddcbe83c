// TEST-ONLY host build of hyperreel_amd/csrc/hr_train.h (the per-ray forward + backward of the training path), so the
// CPU suite can compare it with torch.autograd on the CPU restatement of the reference without a GPU.
// Nothing in the product links or loads this file.
#include "../../hyperreel_amd/csrc/hr_train.h"

extern "C" {

int ht_sizeof_plane() { return (int)sizeof(HrGridPlane); }

const char* ht_unsupported(const hr_config* c) { return hr_train_unsupported(*c); }

// planes: 3 descriptors with a / b pointing at packed fp32 texels; g_a / g_b: packed gradient accumulators (zeroed by
// the caller); d_rgb == NULL runs the forward only.  Returns 0, or -1 when z_channels / ca_total exceed the bounds.
int ht_train(const hr_config* c, const float* rays, const float* head, long long n, const float* d_rgb, float* rgb, float* d_head,
             const HrGridPlane* planes, float** g_a, float** g_b, const float* basis, float* d_basis, int n_basis_cols, int ca_total,
             int white_bg)
{
    if (c->z_channels > 256 || ca_total > HR_TRAIN_MAX_CA) return -1;
    HrTrainArgs a = {};
    a.cfg_dev = c;
    a.rays = rays; a.head = head; a.n_rays = n; a.rgb = rgb; a.d_rgb = d_rgb; a.d_head = d_head;
    for (int j = 0; j < 3; ++j) { a.planes[j] = planes[j]; a.g_a[j] = g_a[j]; a.g_b[j] = g_b[j]; }
    a.basis = basis; a.d_basis = d_basis; a.n_basis_cols = n_basis_cols; a.ca_total = ca_total; a.white_bg = white_bg;
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
    return 0;
}

}  // extern "C"
