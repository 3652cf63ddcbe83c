// C ABI of libhyperreel_hip.so (declared in include/hyperreel_hip.h).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <string>
#include <vector>

#include "hr_kernels.h"
#include "hr_mask.h"
#include "hr_train.h"

// sample wavefronts per workgroup of the frame kernel when the caller does not say (measured: DESIGN.md section 3)
#ifndef HR_DEFAULT_SAMPLE_WAVES
#define HR_DEFAULT_SAMPLE_WAVES 0      // the plan's own choice (8)
#endif

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HR_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess) return fail(HR_E_HIP, "%s failed: %s", #call, hipGetErrorString(e__));  \
    } while (0)

const int MAT_MODE[3][2] = {{0, 1}, {0, 2}, {1, 2}};   // tensorf_base.py:231
const int VEC_MODE[3] = {2, 1, 0};                     // tensorf_base.py:232
const int MAT_MODE_TIME0[3] = {2, 1, 0};               // tensorf_dynamic.py:48 (first index of each pair)

struct DevBuf {
    float* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct hr_model {
    hr_config cfg;        // as handed over by the caller
    hr_config kcfg;       // what the kernels see: dead head columns removed (preds_per_z, field offsets)
    HrColMap col_map;     // user column -> live column (-1: never read by the path, not computed)
    int p_live = 0;
    bool finalized = false;
    std::map<std::string, DevBuf> raw;     // uploaded tensors, reference layout, device memory
    std::map<std::string, size_t> expect;  // name -> expected byte size
    // packed MLP
    float4* wpack[HR_MAX_LAYERS] = {};
    void* wsplit[HR_MAX_LAYERS] = {};
    float* bias[HR_MAX_LAYERS] = {};
    float winv[HR_MAX_LAYERS] = {};       // 2^-s of the packed split weights (HrMlpArgs::winv)
    int xexp[HR_MAX_LAYERS] = {};         // f16 + fp8 split: exponent of the fp8 images of hidden Linear l's output (HrMlpArgs::xexp), from act_max
    int n_tiles[HR_MAX_LAYERS] = {};
    int k0p = 0;
    int n_out = 0;
    int active_precision = HR_MLP_FP32;   // the arithmetic the MLP kernels run: cfg.mlp_precision, with HR_MLP_AUTO resolved (resolve_precision)
    float act_max[HR_MAX_LAYERS] = {};    // calibration: max |input feature|, max |pre-activation| of hidden Linear l - 1
    int calibrated = 0;                   // 0: not calibrated (cascade rows / unsupported width), 1: synthetic rays (finalize), 2: the caller's rays
    unsigned* flags = nullptr;            // sticky device status word (HrMlpArgs::flags)
    // verified fast path: the MLP runs f16f8, rays with a comparison at risk (or a range bit) are listed on the device and rendered
    // again with the f16x3 tiles below by a second, list-driven pass at the end of hr_render
    int verified = 0;
    // tier 1: the f16x3 tiles of the second pass; tier 2: bf16x3 tiles (fp32 exponent range) for the third pass -- the tiles of the second pass in
    // which an activation left the IEEE-half range (what the reference's fp32 BaseMLP, nlf/nets/mlp.py:159-172, cannot do)
    void* wsplit_safe[2][HR_MAX_LAYERS] = {};
    float* bias_safe[2][HR_MAX_LAYERS] = {};
    float winv_safe[2][HR_MAX_LAYERS] = {};
    int n_tiles_safe[2][HR_MAX_LAYERS] = {};
    int64_t mlp_bytes_safe[2] = {};
    int* redo_list = nullptr;            // the second pass's rays
    int* wide_list = nullptr;            // the third pass's rays
    unsigned* redo_count = nullptr;      // [0] second-pass counter, [1] its copy, [2] third-pass counter, [3] its copy
    int redo_cap = 0;                    // entries the list holds (hr_model_reserve); a call uses max(32 768, n_rays / 16) of them
    int wide_cap = 0;
    float redo_band = 0.0f;              // the margins of THIS model (calibrate_band; hr_math.h HrRisk): of zc,
    float redo_band_q = 0.0f;            //   of a point coordinate per unit of amplification,
    float redo_band_off = 0.0f;          //   of the point-offset / flow heads
    hr_verify_info vinfo = {};
    float* calib_rays = nullptr;         // the rays the arithmetic was decided on (synthetic, or a strided sample of the caller's): kept for the band
    int64_t calib_n = 0;
    bool band_stale = false;             // hr_model_update_config changed the activations' constants: the band is measured again before the next render
    int64_t mlp_bytes = 0;
    // packed grids
    float* grid_a[3] = {};   // texel storage (floats, or halfs when cfg.grid_dtype == HR_GRID_FP16)
    float* grid_b[3] = {};
    HrGridPlane planes[3] = {};
    float* basis = nullptr;
    float* basis_t = nullptr;            // column-major copy for the decode-matrix fold (HrSampleArgs::basis_t)
    int* slot_col = nullptr;
    int basis_ld = 0;
    int n_basis_cols = 0;
    int ca_total = 0;
    // workspace
    float* head = nullptr;
    int64_t chunk = 0;
    int64_t packed_bytes = 0;
    // point_prediction cascade (hr_model_create_cascade): `this` is the fine level (point MLP, second intersect,
    // colour); `coarse` holds the ray MLP and the first intersect and owns no grids
    hr_config* kcfg_dev = nullptr;       // device copy of kcfg for the sample kernel (the MLP kernels take it by value)
    hr_model* coarse = nullptr;
    bool is_coarse = false;
    float* rows = nullptr;   // input rows of the point MLP for one chunk: (chunk * casc_in_z, casc_row_dim)
    // training path (hr_train_*): the caller's configuration on the device and packed gradient accumulators
    hr_config* ucfg_dev = nullptr;
    float* grad_a[3] = {};               // training: packed texel-gradient accumulators of the plane pairs -- slices of grad_pool
    float* grad_b[3] = {};
    float* grad_pool = nullptr;          // ONE allocation (cleared by one memset per step)
    size_t grad_pool_bytes = 0;
    void* wsplit_t[HR_MAX_LAYERS] = {};  // training forward (hr_mlp_train_forward): bf16 split tiles of the CURRENT parameter values, re-packed on the device every step
    float* bias_t[HR_MAX_LAYERS] = {};
    int n_tiles_t[HR_MAX_LAYERS] = {};
    long long* grad_fx = nullptr;        // deterministic training (HR_OPT_TRAIN_DETERMINISTIC): ONE 64-bit fixed-point buffer for every accumulator of a step
    size_t grad_fx_elems = 0;
    HrFxUnit* fx_unit = nullptr;         // ... and THIS model's fixed-point unit of the step (hr_train.h)
    int opt_train_det = 0;
    float* tape = nullptr;               // per-sample values between the backward's phases: 8 words x tape_samples
    int64_t tape_samples = 0;
    // occupancy early-reject (hr_model_set_occupancy)
    float* occ = nullptr;
    unsigned* occ_cells = nullptr;        // one bit per lattice cell, built from a 0/1 volume (HrSampleArgs::occ_cells)
    int occ_n[3] = {};
    float occ_lo[3] = {}, occ_inv[3] = {};
    // execution plan of hr_render (hr_model_set_option)
    int frame_row = -1;                  // hr_render_frame: >= 0 while a call renders from frame_line[] (-1: general path)
    float* frame_line[3] = {nullptr, nullptr, nullptr};   // the frame's blended keyframe rows, one line per time plane (float32 texels)
    int opt_frame_kernel = 0;              // two kernels per chunk: level with the frame kernel since K1 took buffer loads (1.96 vs 1.99 ms per DoNeRF frame, interleaved
                                           // events, profiles/r05_headline_diag_*.json) and with the tighter tail (hardware block dispatch instead of a static tile deal)
    int opt_sample_waves = HR_DEFAULT_SAMPLE_WAVES;
    int n_cus = 0;
};

namespace {

int layer_in(const hr_config& c, int l)
{
    if (l == 0) return c.mlp_in;
    return c.mlp_hidden + (((c.mlp_skip_mask >> l) & 1) ? c.mlp_in : 0);
}

// samples whose head values one MLP row produces: all Z of a ray, or Z / casc_in_z per coarse point
int samples_per_row(const hr_config& c) { return c.casc_in_z > 0 ? c.z_channels / c.casc_in_z : c.z_channels; }
int rows_per_ray(const hr_config& c) { return c.casc_in_z > 0 ? c.casc_in_z : 1; }

int layer_out(const hr_config& c, int l) { return (l == c.mlp_layers - 1) ? samples_per_row(c) * c.preds_per_z : c.mlp_hidden; }

// z_vals channels read per sample: z (z_plane, euclidean_distance_unified, voxel_grid), origin xyz + radius
// (sphere/cylinder), origin xyz + resize xyz + raw offset + radius (sphere_new/cylinder_new)
int isect_z_channels(int t)
{
    if (t == HR_ISECT_SPHERE || t == HR_ISECT_CYLINDER || t == HR_ISECT_DEFORMABLE_VOXEL_GRID) return 4;
    if (t == HR_ISECT_SPHERE_NEW || t == HR_ISECT_CYLINDER_NEW) return 8;
    return 1;
}

int validate(const hr_config& c, bool coarse = false)
{
    if (c.ray_dim != 6 && c.ray_dim != 8) return fail(HR_E_INVALID, "ray_dim must be 6 or 8 (got %d)", c.ray_dim);
    if (c.n_groups < 1 || c.n_groups > HR_MAX_GROUPS) return fail(HR_E_INVALID, "n_groups out of range");
    for (int g = 0; g < c.n_groups; ++g)
        if (c.groups[g].pe_type == HR_PE_WINDOWED && c.groups[g].pe_n_freqs > HR_MAX_FREQS)
            return fail(HR_E_INVALID, "windowed positional encoding with more than %d frequencies", HR_MAX_FREQS);
    if (c.mlp_layers != 0) {   // 0: ZeroMLP (nlf/nets/mlp.py:14-33), the head is all zeros and samples sit on their anchors
        if (c.mlp_hidden != 64 && c.mlp_hidden != 128 && c.mlp_hidden != 256)
            return fail(HR_E_INVALID, "mlp_hidden must be 64, 128 or 256 (got %d)", c.mlp_hidden);
        // nn.LeakyReLU(0.01) (nlf/nets/mlp.py:149-154).  The split kernels evaluate it as max(v, slope v), which is the same function for a slope in [0, 1]
        if (!(c.leaky_slope >= 0.0f && c.leaky_slope <= 1.0f))
            return fail(HR_E_INVALID, "leaky_slope must be in [0, 1] (got %g)", (double)c.leaky_slope);
        if (c.mlp_layers < 2 || c.mlp_layers > HR_MAX_LAYERS) return fail(HR_E_INVALID, "mlp_layers must be 0 or in [2,%d]", HR_MAX_LAYERS);
        if (c.mlp_in < 1 || c.mlp_in > HR_MAX_MLP_IN) return fail(HR_E_INVALID, "mlp_in must be in [1,%d]", HR_MAX_MLP_IN);
        if (c.mlp_skip_mask & 1) return fail(HR_E_INVALID, "layer 0 cannot be a skip layer");
    }
    if (c.z_channels < 1 || c.z_channels > HR_KERNEL_MAX_Z) return fail(HR_E_INVALID, "z_channels must be in [1,%d]", HR_KERNEL_MAX_Z);
    if (c.preds_per_z < 1 || c.preds_per_z > 64) return fail(HR_E_INVALID, "preds_per_z out of range");
    const hr_head_field* fs[9] = {&c.f_z_vals, &c.f_isect_sigma, &c.f_offset_sigma, &c.f_point_offset, &c.f_color_scale,
                                  &c.f_color_shift, &c.f_spatial_flow, &c.f_color_scale_global, &c.f_color_shift_global};
    for (const hr_head_field* f : fs)
        if (f->offset >= 0 && f->offset + f->channels > c.preds_per_z) return fail(HR_E_INVALID, "head field exceeds preds_per_z");
    if (c.f_z_vals.offset < 0) return fail(HR_E_INVALID, "z_vals head is required");
    if (c.isect_type < HR_ISECT_Z_PLANE || c.isect_type > HR_ISECT_DEFORMABLE_VOXEL_GRID) return fail(HR_E_INVALID, "unknown isect_type %d", c.isect_type);
    if (c.f_z_vals.channels != isect_z_channels(c.isect_type))
        return fail(HR_E_INVALID, "z_vals needs %d channel(s) for intersect type %d (got %d)", isect_z_channels(c.isect_type),
                    c.isect_type, c.f_z_vals.channels);
    if (c.isect_type == HR_ISECT_VOXEL_GRID && c.z_channels % 3) return fail(HR_E_INVALID, "voxel_grid needs z_channels divisible by 3");
    if (c.isect_type == HR_ISECT_DEFORMABLE_VOXEL_GRID && (c.dvg_axes < 1 || c.dvg_axes > 3 || c.z_channels % c.dvg_axes))
        return fail(HR_E_INVALID, "deformable_voxel_grid needs 1..3 start normals dividing z_channels");
    if (c.contract_type < HR_CONTRACT_IDENTITY || c.contract_type > HR_CONTRACT_DONERF) return fail(HR_E_INVALID, "unknown contract_type");
    if (c.contract_type == HR_CONTRACT_DONERF && !(c.c_pow_fac > 0.0f && c.c_pow_power > 0.0f && c.c_pow_inv_power > 0.0f))
        return fail(HR_E_INVALID, "donerf contraction needs positive c_pow_fac / c_pow_power / c_pow_inv_power");
    if (c.contract_type == HR_CONTRACT_AFFINE)
        for (int i = 0; i < 3; ++i)
            if (c.c_aff_size[i] == 0.0f) return fail(HR_E_INVALID, "affine contraction with an empty box");
    if ((c.f_color_scale.offset >= 0) != (c.f_color_shift.offset >= 0)) return fail(HR_E_INVALID, "color_scale and color_shift come together");
    if ((c.f_color_scale_global.offset >= 0) != (c.f_color_shift_global.offset >= 0))
        return fail(HR_E_INVALID, "color_scale_global and color_shift_global come together");
    if (c.point_offset && (c.f_point_offset.offset < 0 || c.f_point_offset.channels != 3)) return fail(HR_E_INVALID, "point_offset head missing");
    if (c.advect && c.use_spatial_flow && (c.f_spatial_flow.offset < 0 || c.f_spatial_flow.channels != 3))
        return fail(HR_E_INVALID, "spatial_flow head missing");
    if (c.video && c.ray_dim != 8) return fail(HR_E_INVALID, "video net needs 8-column rays");
    if (c.video && !coarse && (!c.advect || c.num_keyframes < 1)) return fail(HR_E_INVALID, "video net needs the advect stage and num_keyframes >= 1");
    if (c.casc_in_z < 0 || (coarse && c.casc_in_z != 0)) return fail(HR_E_INVALID, "casc_in_z is set on the fine config of a cascade only");
    if (c.casc_in_z > 0) {
        if (c.z_channels % c.casc_in_z) return fail(HR_E_INVALID, "z_channels must be a multiple of casc_in_z");
        if (c.casc_n_inputs < 1 || c.casc_n_inputs > 4 || c.casc_row_dim < 1 || c.casc_row_dim > 8)
            return fail(HR_E_INVALID, "point_prediction rows: 1..4 inputs, 1..8 columns");
        int sum = 0;
        for (int i = 0; i < c.casc_n_inputs; ++i) {
            if (c.casc_input_kind[i] < HR_PIN_POINTS || c.casc_input_kind[i] > HR_PIN_TIMES || c.casc_input_dim[i] < 1 || c.casc_input_dim[i] > 3)
                return fail(HR_E_INVALID, "bad point_prediction input %d", i);
            sum += c.casc_input_dim[i];
        }
        if (sum != c.casc_row_dim) return fail(HR_E_INVALID, "casc_row_dim does not match the inputs");
        for (int g = 0; g < c.n_groups; ++g)
            if (c.groups[g].fn != HR_PARAM_IDENTITY || c.groups[g].end > c.casc_row_dim)
                return fail(HR_E_INVALID, "point_prediction params must be identity groups over the row's columns");
    }
    for (int i = 0; i < 3; ++i)
        if (c.grid[i] < 2) return fail(HR_E_INVALID, "grid size must be >= 2 on every axis");
    if (c.shading == HR_SHADING_RGB ? c.app_dim != 3 : c.app_dim != 27) return fail(HR_E_INVALID, "app_dim must be 3 (RGB) or 27 (SH)");
    if (c.mlp_precision < HR_MLP_FP32 || c.mlp_precision > HR_MLP_F16F8V) return fail(HR_E_INVALID, "unknown mlp_precision");
    if (c.mlp_layers != 0 && c.mlp_precision != HR_MLP_FP32 && c.mlp_precision != HR_MLP_AUTO && c.mlp_hidden != 256)      // (AUTO resolves to fp32 there)
        return fail(HR_E_INVALID, "the split (bf16x3 / f16x3) MLP needs mlp_hidden == 256");
    if (c.grid_dtype != HR_GRID_FP32 && c.grid_dtype != HR_GRID_FP16) return fail(HR_E_INVALID, "unknown grid_dtype");
    if (c.color_table_views < 0) return fail(HR_E_INVALID, "negative color_table_views");
    if (c.color_table_views > 0 && c.ray_dim != 8) return fail(HR_E_INVALID, "the colour table is indexed by rays[..., -2]: needs 8-column rays");
    return HR_OK;
}

// Which of the P per-sample head columns does the path read?  Columns that are not read are
// dropped from the last Linear (fewer MFMAs, smaller head).  Shipped cases: the three sphere /
// cylinder origin channels when origin_scale_factor == 0 (primitive.py:410-412 multiplies them
// by zero) and `point_sigma` in models whose point_offset stage reads `sigma` instead.
void analyse_live_columns(hr_model* m)
{
    const hr_config& c = m->cfg;
    bool live[64] = {};
    auto mark = [&](const hr_head_field& f, int first, int count) {
        if (f.offset < 0) return;
        for (int i = first; i < first + count && f.offset + i < 64; ++i) live[f.offset + i] = true;
    };
    int z_anchor = 0;                 // a z_vals channel that is always read
    if (c.isect_type == HR_ISECT_SPHERE || c.isect_type == HR_ISECT_CYLINDER) {
        z_anchor = 3;
        mark(c.f_z_vals, 3, 1);
        if (c.origin_scale != 0.0f) mark(c.f_z_vals, 0, 3);
    } else if (c.isect_type == HR_ISECT_DEFORMABLE_VOXEL_GRID) {
        z_anchor = 3;
        mark(c.f_z_vals, 3, 1);
        if (c.dvg_normal_scale != 0.0f) mark(c.f_z_vals, 0, 3);
    } else if (c.isect_type == HR_ISECT_SPHERE_NEW || c.isect_type == HR_ISECT_CYLINDER_NEW) {
        z_anchor = 7;
        mark(c.f_z_vals, 6, 2);
        // kept contiguous up to the anchor so that offset + channel stays valid after compaction
        if (c.resize_scale != 0.0f || c.origin_scale != 0.0f) mark(c.f_z_vals, 3, 3);
        if (c.origin_scale != 0.0f) mark(c.f_z_vals, 0, 3);
    } else {
        mark(c.f_z_vals, 0, 1);
    }
    mark(c.f_isect_sigma, 0, 1);
    if (c.point_offset) {
        mark(c.f_point_offset, 0, 3);
        mark(c.f_offset_sigma, 0, 1);
    }
    mark(c.f_color_scale, 0, 3);
    mark(c.f_color_shift, 0, 3);
    mark(c.f_color_scale_global, 0, c.f_color_scale_global.channels == 9 ? 9 : 3);    // 9: the head is a 3x3 `color_transform_global`
    mark(c.f_color_shift_global, 0, 3);
    if (c.advect && c.use_spatial_flow) mark(c.f_spatial_flow, 0, 3);
    const char* e = getenv("HR_PRUNE");
    const bool prune = !(e && e[0] == '0');
    int n = 0;
    for (int i = 0; i < 64; ++i) {
        const bool keep = (i < c.preds_per_z) && (live[i] || !prune);
        m->col_map.col[i] = keep ? n++ : -1;
    }
    m->p_live = n;
    m->kcfg = c;
    m->kcfg.preds_per_z = n;
    auto remap = [&](hr_head_field& f, int anchor) {   // anchor: a channel of the field that is always live
        if (f.offset < 0) return;
        f.offset = m->col_map.col[f.offset + anchor] - anchor;
    };
    remap(m->kcfg.f_z_vals, z_anchor);   // may become negative: only the live channels are read then
    remap(m->kcfg.f_isect_sigma, 0);
    if (c.point_offset) { remap(m->kcfg.f_point_offset, 0); remap(m->kcfg.f_offset_sigma, 0); }
    else { m->kcfg.f_point_offset.offset = -1; m->kcfg.f_offset_sigma.offset = -1; }
    remap(m->kcfg.f_color_scale, 0);
    remap(m->kcfg.f_color_shift, 0);
    remap(m->kcfg.f_color_scale_global, 0);
    remap(m->kcfg.f_color_shift_global, 0);
    if (c.advect && c.use_spatial_flow) remap(m->kcfg.f_spatial_flow, 0); else m->kcfg.f_spatial_flow.offset = -1;
}

void free_dev(float*& p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

// float -> bf16 bits, round to nearest even (finite inputs)
uint16_t bf16_rne(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// float -> IEEE half bits and back (round to nearest even; overflow -> inf like the hardware conversion)
uint16_t f16_rne(float f)
{
    const _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

float f16_to_float(uint16_t u)
{
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}

// OCP e4m3 (what v_mfma_scale_f32_32x32x64_f8f6f4 reads with cbsz / blgp = 0): 1-4-3, bias 7, no infinities, 0x7f = NaN, largest 448;
// round-to-nearest-even, subnormals down to 2^-9.  The packed weights stay below 2^8 by construction, so nothing saturates here.
uint8_t e4m3_rne(float f)
{
    const uint8_t sign = std::signbit(f) ? 0x80 : 0;
    float a = fabsf(f);
    if (!(a == a)) return 0x7f;
    if (a > 448.0f) a = 448.0f;
    if (a < ldexpf(1.0f, -10)) return sign;                    // below half the smallest subnormal (a tie at 2^-10 goes to even = 0)
    int e = 0;
    (void)frexpf(a, &e);                                       // a = m * 2^e, m in [0.5, 1)
    int ex = e - 1;                                            // a = 1.xxx * 2^ex
    if (ex < -6) ex = -6;                                      // subnormal range: fixed quantum 2^-9
    const float q = ldexpf(1.0f, ex - 3);                      // spacing
    const float r = nearbyintf(a / q);                         // ties to even (default rounding mode)
    float v = r * q;
    if (v > 448.0f) v = 448.0f;
    if (v < ldexpf(1.0f, -6)) return (uint8_t)(sign | (int)(v / ldexpf(1.0f, -9)));
    (void)frexpf(v, &e);
    const int E = e - 1 + 7;
    const int M = (int)(v / ldexpf(1.0f, e - 1 - 3)) - 8;
    return (uint8_t)(sign | (E << 3) | M);
}
float bf16_to_float(uint16_t h)
{
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

}  // namespace

extern "C" {

int hr_abi_version(void) { return HR_ABI_VERSION; }

int hr_sizeof_config(void) { return (int)sizeof(hr_config); }

const char* hr_last_error(void) { return g_err; }

static int create_level(const hr_config* cfg, bool coarse, hr_model** out)
{
    if (!cfg || !out) return fail(HR_E_INVALID, "null argument");
    *out = nullptr;
    int rc = validate(*cfg, coarse);
    if (rc != HR_OK) return rc;
    int ndev = 0;
    HR_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(HR_E_HIP, "no HIP device");
    hr_model* m = new hr_model();
    m->cfg = *cfg;
    m->is_coarse = coarse;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) m->n_cus = prop.multiProcessorCount;
        if (m->n_cus < 1) m->n_cus = 256;
    }
    analyse_live_columns(m);
    {   // the kernels read the configuration from device memory
        if (hipMalloc((void**)&m->kcfg_dev, sizeof(hr_config)) != hipSuccess) {
            delete m;
            return fail(HR_E_HIP, "hipMalloc of the device configuration failed");
        }
        (void)hipMemcpy(m->kcfg_dev, &m->kcfg, sizeof(hr_config), hipMemcpyHostToDevice);
    }
    const hr_config& c = m->cfg;
    {   // LDS of the sample kernel: 256/ZP rays x head rows x (live head columns + 4) + the decode matrices
        int ZP = 8;
        while (ZP < c.z_channels) ZP <<= 1;
        const size_t rpb = 256 / ZP, nq = ((size_t)samples_per_row(c) * m->p_live + 3) / 4;
        size_t ca = 0;
        for (int j = 0; j < 3; ++j) ca += 4 * (size_t)((c.n_app[j] + 3) / 4);
        const size_t lds = 4 * (rpb * rows_per_ray(c) * (nq * 4 + 4) + rpb * 3 * ca + 256);
        if (lds > 160 * 1024 - 4096) {             // (- the static words of the sample kernel: the ray records, hr_gather_ones)
            const int z = c.z_channels, pl = m->p_live;
            hr_model_destroy(m);                      // also releases the device configuration
            return fail(HR_E_INVALID, "z_channels %d x %d head columns need %zu bytes of LDS per workgroup (160 KiB available)", z, pl, lds);
        }
    }
    char name[64];
    for (int l = 0; l < c.mlp_layers; ++l) {
        snprintf(name, sizeof(name), "mlp.%d.weight", l);
        m->expect[name] = sizeof(float) * (size_t)layer_out(c, l) * layer_in(c, l);
        snprintf(name, sizeof(name), "mlp.%d.bias", l);
        m->expect[name] = sizeof(float) * (size_t)layer_out(c, l);
    }
    if (coarse) {          // ray MLP + first intersect only: no grids
        *out = m;
        return HR_OK;
    }
    int n_app_sum = 0;
    for (int j = 0; j < 3; ++j) {
        const size_t hw = (size_t)c.grid[MAT_MODE[j][1]] * c.grid[MAT_MODE[j][0]];
        const char* kinds[2] = {"density", "app"};
        const int nch[2] = {c.n_den[j], c.n_app[j]};
        for (int t = 0; t < 2; ++t) {
            if (c.video) {
                snprintf(name, sizeof(name), "%s_plane_space.%d", kinds[t], j);
                m->expect[name] = sizeof(float) * nch[t] * hw;
                snprintf(name, sizeof(name), "%s_plane_time.%d", kinds[t], j);
                m->expect[name] = sizeof(float) * (size_t)nch[t] * c.num_keyframes * c.grid[MAT_MODE_TIME0[j]];
            } else {
                snprintf(name, sizeof(name), "%s_plane.%d", kinds[t], j);
                m->expect[name] = sizeof(float) * nch[t] * hw;
                snprintf(name, sizeof(name), "%s_line.%d", kinds[t], j);
                m->expect[name] = sizeof(float) * (size_t)nch[t] * c.grid[VEC_MODE[j]];
            }
        }
        n_app_sum += c.n_app[j];
    }
    m->expect["basis_mat.weight"] = sizeof(float) * (size_t)c.app_dim * n_app_sum;
    if (c.color_table_views > 0) m->expect["color_embedding"] = sizeof(float) * (size_t)c.color_table_views * 12;
    *out = m;
    return HR_OK;
}

int hr_model_create(const hr_config* cfg, hr_model** out)
{
    if (cfg && cfg->casc_in_z != 0) return fail(HR_E_INVALID, "a cascade's fine config goes through hr_model_create_cascade");
    return create_level(cfg, false, out);
}

int hr_model_create_cascade(const hr_config* coarse, const hr_config* fine, hr_model** out)
{
    if (!coarse || !fine || !out) return fail(HR_E_INVALID, "null argument");
    *out = nullptr;
    if (fine->casc_in_z <= 0) return fail(HR_E_INVALID, "the fine config needs casc_in_z (samples of the coarse level)");
    if (fine->casc_in_z != coarse->z_channels) return fail(HR_E_INVALID, "casc_in_z %d != coarse z_channels %d", fine->casc_in_z, coarse->z_channels);
    if (fine->ray_dim != coarse->ray_dim) return fail(HR_E_INVALID, "both levels read the same rays: ray_dim must agree");
    hr_model* c = nullptr;
    int rc = create_level(coarse, true, &c);
    if (rc != HR_OK) return rc;
    hr_model* m = nullptr;
    rc = create_level(fine, false, &m);
    if (rc != HR_OK) {
        hr_model_destroy(c);
        return rc;
    }
    m->coarse = c;
    *out = m;
    return HR_OK;
}

int hr_model_upload(hr_model* m, const char* name, const void* ptr, size_t bytes)
{
    if (!m || !name) return fail(HR_E_INVALID, "null argument");
    std::string key = name;
    if (m->coarse) {                      // cascade: mlp.* is the coarse ray MLP, mlp1.* the point MLP of this level
        if (key.compare(0, 4, "mlp.") == 0) {
            m->finalized = false;
            return hr_model_upload(m->coarse, name, ptr, bytes);
        }
        if (key.compare(0, 5, "mlp1.") == 0) key = "mlp." + key.substr(5);
    }
    name = key.c_str();
    auto it = m->expect.find(name);
    if (it == m->expect.end()) return fail(HR_E_INVALID, "unknown tensor name '%s'", name);
    if (bytes != it->second) return fail(HR_E_INVALID, "tensor '%s': expected %zu bytes, got %zu", name, it->second, bytes);
    if (bytes > 0 && !ptr) return fail(HR_E_INVALID, "tensor '%s': null data", name);
    DevBuf& b = m->raw[name];
    if (b.bytes != bytes || (bytes > 0 && !b.p)) {
        free_dev(b.p);
        b.bytes = bytes;
        if (bytes > 0) HR_HIP(hipMalloc((void**)&b.p, bytes));
    }
    if (bytes > 0) HR_HIP(hipMemcpy(b.p, ptr, bytes, hipMemcpyDefault));
    m->finalized = false;
    return HR_OK;
}

// The MLP's weights re-laid out for the active arithmetic (m->active_precision).  Called by hr_model_finalize and again by
// hr_model_calibrate when the calibration changes that choice.
// f16 + fp8 split: how far above the calibration's largest activation of a layer the fp8 image of that layer's output still is finite
// (e4m3 keeps 4 significant bits over the 15 octaves below that; the correction products it feeds are 2^-11 of the result)
static const float HR_F8_HEADROOM = 16.0f;

// fp8 image of hidden Linear l's output (f16 + fp8 split only): e4m3(x * 2^-Ea) with the calibration's largest |pre-activation| of that layer
// (act_max[l + 1]) times HR_F8_HEADROOM at or below 448 -- beyond 448 * 2^Ea the image saturates and the kernels say so (HR_OPT_MLP_F8_SATURATED).  Depends on the
// calibration only, not on the packed weights: hr_model_calibrate refreshes it without re-packing.
static void f8_exponents(hr_model* m)
{
    for (int l = 0; l < HR_MAX_LAYERS; ++l) {
        m->xexp[l] = 0;
        if (m->active_precision != HR_MLP_F16F8 || l + 1 >= m->cfg.mlp_layers) continue;
        const float mx = m->act_max[l + 1] * HR_F8_HEADROOM;
        int e = 0;
        if (mx > 0.0f && std::isfinite(mx)) {
            (void)frexpf(mx / 448.0f, &e);              // mx / 448 = f * 2^e, f in [0.5, 1): mx <= 448 * 2^e
            e = e < -30 ? -30 : (e > 30 ? 30 : e);
        }
        m->xexp[l] = e;
    }
}

struct HrPackOut {                       // where one packing of the MLP goes (the model's primary tiles, or the verified path's f16x3 ones)
    float4** wpack;
    void** wsplit;
    float** bias;
    float* winv;
    int* n_tiles;
    int64_t* bytes;
};

static int pack_mlp_as(hr_model* m, const int precision, const HrPackOut o)
{
    const hr_config& c = m->cfg;
    char name[64];
    *o.bytes = 0;
    // ---- MLP: MFMA B-operand tiles (layout documented in hr_kernels.h)
    const int W = c.mlp_hidden;
    m->k0p = (c.mlp_in + 15) & ~15;
    m->n_out = samples_per_row(c) * m->p_live;     // head columns of one MLP row
    const int P_user = c.preds_per_z, P_live = m->p_live;
    int live_cols[64];
    for (int i = 0, j = 0; i < P_user; ++i)
        if (m->col_map.col[i] >= 0) live_cols[j++] = i;
    for (int l = 0; l < c.mlp_layers; ++l) {
        const bool last = (l == c.mlp_layers - 1);
        const int N_user = layer_out(c, l), Kt = layer_in(c, l);
        const int N = last ? m->n_out : N_user;   // rows the kernels compute
        const bool first = (l == 0);
        const bool skip = (c.mlp_skip_mask >> l) & 1;
        const int Kp = first ? m->k0p : (skip ? m->k0p + W : W);
        const bool split = (precision != HR_MLP_FP32);
        const bool f8lo = (precision == HR_MLP_F16F8);
        const bool half = (precision == HR_MLP_F16X3 || precision == HR_MLP_F16X2 || f8lo);
        const int tile_n = split ? 32 : 16;
        const int nt = (N + tile_n - 1) / tile_n;
        std::vector<float> w((size_t)N_user * Kt), b(N_user);
        snprintf(name, sizeof(name), "mlp.%d.weight", l);
        HR_HIP(hipMemcpy(w.data(), m->raw[name].p, w.size() * sizeof(float), hipMemcpyDeviceToHost));
        snprintf(name, sizeof(name), "mlp.%d.bias", l);
        HR_HIP(hipMemcpy(b.data(), m->raw[name].p, b.size() * sizeof(float), hipMemcpyDeviceToHost));
        // torch weight element for (output feature n, kernel K index kk); 0 outside the matrix
        auto wk = [&](int n, int kk) -> float {
            int col = -1;                                                 // torch in-feature index
            if (first) {
                if (kk < c.mlp_in) col = kk;
            } else if (skip) {
                if (kk < m->k0p) { if (kk < c.mlp_in) col = kk; }
                else col = c.mlp_in + (kk - m->k0p);                      // cat([input, x]), mlp.py:166-168
            } else {
                col = kk;
            }
            if (!(n < N && col >= 0 && col < Kt)) return 0.0f;
            // last layer: kernel row n = k*P_live + c' is the user's row k*P + live_cols[c']
            const int row = last ? (n / P_live) * P_user + live_cols[n % P_live] : n;
            return w[(size_t)row * Kt + col];
        };
        if (o.wpack) free_dev(reinterpret_cast<float*&>(o.wpack[l]));
        free_dev(reinterpret_cast<float*&>(o.wsplit[l]));
        free_dev(o.bias[l]);
        // fp16 modes: the weights of these MLPs are ~1/sqrt(fan_in), so the low half w - half(w) (~2^-12 w) would be a
        // subnormal half with an ABSOLUTE rounding error of 2^-25.  Packing w * 2^s (exact), with s putting the largest
        // weight of the layer into [2^13, 2^14), keeps every low half of a weight above max|w| * 2^-16 normal; the
        // epilogue multiplies the accumulator by 2^-s (exact again).  bf16 halves have the fp32 exponent range: s = 0.
        float wmul = 1.0f;
        o.winv[l] = 1.0f;
        if (half) {
            float mx = 0.0f;
            for (float v : w) mx = fmaxf(mx, fabsf(v));
            if (mx > 0.0f && std::isfinite(mx)) {
                int e = 0;
                (void)frexpf(mx, &e);                       // mx = f * 2^e, f in [0.5, 1)
                int sft = 14 - e;
                sft = sft < -14 ? -14 : (sft > 40 ? 40 : sft);
                wmul = ldexpf(1.0f, sft);
                o.winv[l] = ldexpf(1.0f, -sft);
            }
        }
        if (!split) {
            std::vector<float> pk((size_t)(Kp / 16) * nt * 64 * 4, 0.0f);
            for (int kt = 0; kt < Kp / 16; ++kt)
                for (int t = 0; t < nt; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int s = 0; s < 4; ++s)
                            pk[(((size_t)kt * nt + t) * 64 + lane) * 4 + s] = wk(16 * t + (lane & 15), 16 * kt + 4 * (lane >> 4) + s);
            HR_HIP(hipMalloc((void**)&o.wpack[l], pk.size() * sizeof(float)));
            HR_HIP(hipMemcpy(o.wpack[l], pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
            *o.bytes += (int64_t)pk.size() * sizeof(float);
        } else {
            // hi = bf16(w), lo = bf16(w - hi), both round-to-nearest-even (layout: hr_kernels.h)
            std::vector<uint16_t> pk((size_t)(Kp / 16) * nt * 2 * 64 * 8, 0);
            for (int kt = 0; kt < Kp / 16; ++kt)
                for (int t = 0; t < nt; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const float v = wk(32 * t + (lane & 31), 16 * kt + 8 * (lane >> 5) + j) * wmul;
                            const uint16_t hi = half ? f16_rne(v) : bf16_rne(v);
                            const uint16_t lo = half ? f16_rne(v - f16_to_float(hi)) : bf16_rne(v - bf16_to_float(hi));
                            const size_t base = ((((size_t)kt * nt + t) * 2) * 64 + lane) * 8 + j;
                            pk[base] = hi;
                            pk[base + 64 * 8] = lo;
                        }
            if (f8lo) {
                // f16 + fp8 split (mlp_split_core.inc, hr_accumulate_f8): over the HIDDEN k-steps (those past the input segment of the first / skip
                // layer, which keeps three f16 products) the 16 bytes of a lane's "lo" half become the fp8 images of the SAME 8 weights its f16 half
                // holds: e4m3((w' - half(w')) * 2^6) x 8, then e4m3(w' * 2^-6) x 8
                const int kseg = first ? Kp / 16 : (skip ? m->k0p / 16 : 0);
                uint8_t* bytes = reinterpret_cast<uint8_t*>(pk.data());
                for (int kt = kseg; kt < Kp / 16; ++kt)
                    for (int t = 0; t < nt; ++t)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const float v = wk(32 * t + (lane & 31), 16 * kt + 8 * (lane >> 5) + j) * wmul;
                                const size_t at = (((((size_t)kt * nt + t) * 2 + 1) * 64 + lane) * 8) * 2;
                                bytes[at + j] = e4m3_rne(ldexpf(v - f16_to_float(f16_rne(v)), 6));
                                bytes[at + 8 + j] = e4m3_rne(ldexpf(v, -6));
                            }
            }
            HR_HIP(hipMalloc((void**)&o.wsplit[l], pk.size() * sizeof(uint16_t)));
            HR_HIP(hipMemcpy(o.wsplit[l], pk.data(), pk.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            *o.bytes += (int64_t)pk.size() * sizeof(uint16_t);
        }
        const int nb = nt * tile_n;
        std::vector<float> bp(nb, 0.0f);
        // split kernels: the accumulators START from the bias (mlp_split_core.inc, hr_acc_init_bias), in the accumulator's unit: b * 2^s (exact;
        // 1 for bf16 halves and for the exact-fp32 kernel, which adds its bias in the epilogue)
        for (int i = 0; i < N; ++i) bp[i] = b[last ? (i / P_live) * P_user + live_cols[i % P_live] : i] * wmul;
        HR_HIP(hipMalloc((void**)&o.bias[l], nb * sizeof(float)));
        HR_HIP(hipMemcpy(o.bias[l], bp.data(), nb * sizeof(float), hipMemcpyHostToDevice));
        o.n_tiles[l] = nt;
        *o.bytes += (int64_t)nb * sizeof(float);
    }
    return HR_OK;
}

static void free_safe_pack(hr_model* m)
{
    for (int t = 0; t < 2; ++t) {
        for (int l = 0; l < HR_MAX_LAYERS; ++l) {
            free_dev(reinterpret_cast<float*&>(m->wsplit_safe[t][l]));
            free_dev(m->bias_safe[t][l]);
        }
        m->mlp_bytes_safe[t] = 0;
    }
}

static const float HR_BAND_FLOOR = 1e-6f;
static const float HR_VERIFY_LISTED_LIMIT = 0.05f; // fraction of the calibration rays the first pass may list (a call's list holds a sixteenth of its rays)
static const float HR_VERIFY_RGB_LIMIT = 6e-5f;   // on <= 65 536 calibration rays; the shipped families measure 1.5e-5 - 5e-5 here and 2.3e-5 - 5.4e-5 on their 640 000-ray frames
static const float HR_VERIFY_AMP_CUT = 2.0f;      // a ray with a live sample beyond it (60 degrees off a plane's normal; a sphere nearly tangent) is not what the margins
                                                  // are measured on -- its errors are the geometry's, the MLP's two-plane / Pluecker inputs included -- and is always listed

static float scene_extent(const hr_config& c)
{
    float ext = 1.0f;
    for (int i = 0; i < 6; ++i) if (std::isfinite(c.aabb[i])) ext = fmaxf(ext, fabsf(c.aabb[i]));
    return fmaxf(ext, std::isfinite(c.near) ? fabsf(c.near) : 0.0f);
}

// the primary tiles in the active arithmetic, and -- verified fast path -- the f16x3 tiles of the second pass
static int pack_mlp(hr_model* m)
{
    f8_exponents(m);
    m->k0p = (m->cfg.mlp_in + 15) & ~15;
    m->n_out = samples_per_row(m->cfg) * m->p_live;
    int64_t b1 = 0;
    int rc = pack_mlp_as(m, m->active_precision, HrPackOut{m->wpack, m->wsplit, m->bias, m->winv, m->n_tiles, &b1});
    if (rc != HR_OK) return rc;
    free_safe_pack(m);
    if (m->verified) {
        rc = pack_mlp_as(m, HR_MLP_F16X3, HrPackOut{nullptr, m->wsplit_safe[0], m->bias_safe[0], m->winv_safe[0], m->n_tiles_safe[0], &m->mlp_bytes_safe[0]});
        if (rc != HR_OK) return rc;
        rc = pack_mlp_as(m, HR_MLP_BF16X3, HrPackOut{nullptr, m->wsplit_safe[1], m->bias_safe[1], m->winv_safe[1], m->n_tiles_safe[1], &m->mlp_bytes_safe[1]});
        if (rc != HR_OK) return rc;
        if (!m->redo_count) HR_HIP(hipMalloc((void**)&m->redo_count, 4 * sizeof(unsigned)));
        HR_HIP(hipMemset(m->redo_count, 0, 4 * sizeof(unsigned)));
        // the margins of the decisions at risk (HrRisk): here their floor -- four float32 ulps of the largest |zc|; the model's own are measured
        // by calibrate_band once the workspace exists
        m->redo_band = m->redo_band_q = HR_BAND_FLOOR;
        m->redo_band_off = 0.0f;
        m->band_stale = true;
    }
    m->mlp_bytes = b1 + m->mlp_bytes_safe[0] + m->mlp_bytes_safe[1];
    return HR_OK;
}

// largest activation a model may show in calibration for the fp16 split arithmetic to be used: a factor 8 below the IEEE-half maximum,
// because calibration sees 4096 rays and a frame has 640 000
static const float HR_F16_CALIBRATION_LIMIT = 65504.0f / 8.0f;

// Activation range of the MLP on `rays_dev` (NULL: 4096 synthetic rays -- origins uniform in the scene box, unit directions,
// times in [0, 1)) -> m->act_max, then the arithmetic: HR_MLP_AUTO becomes f16x3 when every input feature and hidden activation
// stays below HR_F16_CALIBRATION_LIMIT and bf16x3 (fp32 exponent range) otherwise; a FORCED fp16 mode that fails the test is an error.
static int resolve_precision(hr_model* m, const float* rays_dev, int64_t n, hipStream_t st)
{
    const hr_config& c = m->cfg;
    const int want = c.mlp_precision;
    for (int l = 0; l < HR_MAX_LAYERS; ++l) m->act_max[l] = 0.0f;
    m->calibrated = 0;
    m->verified = 0;
    if (c.mlp_layers == 0 || want == HR_MLP_FP32 || want == HR_MLP_BF16X3) {
        m->active_precision = (c.mlp_layers == 0 && want == HR_MLP_AUTO) ? HR_MLP_F16X3 : want;
        return HR_OK;
    }
    if (want == HR_MLP_AUTO && c.mlp_hidden != 256) {        // the split kernels are written for 256-wide layers
        m->active_precision = HR_MLP_FP32;
        return HR_OK;
    }
    if (!hr_mlp_range_supported(c))
        return fail(HR_E_INVALID, "activation-range calibration does not cover mlp_in %d / mlp_hidden %d", c.mlp_in, c.mlp_hidden);
    float* synth = nullptr;
    float* d_max = nullptr;
    hipError_t e = hipSuccess;
    const int rd = m->coarse ? c.casc_row_dim : c.ray_dim;
    if (rays_dev) {
        // the caller's rays: a strided sample of at most 65 536 of them stays with the model (the band of the verified fast path is measured
        // on it, again after hr_model_update_config)
        const int64_t stride = (n + 65535) / 65536, keep = (n + stride - 1) / stride;
        free_dev(m->calib_rays);
        m->calib_n = 0;
        e = hipMalloc((void**)&m->calib_rays, sizeof(float) * keep * rd);
        if (e == hipSuccess) e = hipMemcpy2DAsync(m->calib_rays, sizeof(float) * rd, rays_dev, sizeof(float) * rd * stride, sizeof(float) * rd, keep,
                                                  hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) m->calib_n = keep;
    }
    if (!rays_dev) {
        n = 4096;
        e = hipMalloc((void**)&synth, sizeof(float) * n * rd);
        if (e == hipSuccess) {
            // where rays start: the model's own box, or (cascade rows, whose first columns are points) the same box.  Real cameras may stand
            // outside it, and a parameterisation such as the Pluecker moment o x d grows with |o|: a caller who has the real rays passes
            // them (hr_model_calibrate), and the host checks the kernels' sticky overflow bit on the first rendered batches and
            // falls back to bf16x3 (models.py, _overflow_guard).  (Origins in a box three times as large were tried for this default: with
            // uniformly random directions the two-plane families then show activations of 1.8e4 that no camera of theirs produces,
            // and AUTO would give up f16x3 -- and its zero flipped rays, DESIGN 3a -- on every one of them.)
            hr_launch_synthetic_rays(synth, n, rd, c.aabb, c.aabb + 3, 0x5eedu, st);
            rays_dev = synth;
        }
    }
    if (e == hipSuccess) e = hipMalloc((void**)&d_max, sizeof(float) * HR_MAX_LAYERS);
    if (e == hipSuccess) e = hipMemsetAsync(d_max, 0, sizeof(float) * HR_MAX_LAYERS, st);
    if (e == hipSuccess) {
        HrRangeArgs ra;
        ra.rays = rays_dev;
        ra.n_rays = n;
        ra.act_max = d_max;
        char name[64];
        for (int l = 0; l < HR_MAX_LAYERS; ++l) {
            ra.w[l] = ra.b[l] = nullptr;
            if (l < c.mlp_layers) {
                snprintf(name, sizeof(name), "mlp.%d.weight", l);
                ra.w[l] = m->raw[name].p;
                snprintf(name, sizeof(name), "mlp.%d.bias", l);
                ra.b[l] = m->raw[name].p;
            }
        }
        hr_config kc = m->kcfg;
        if (m->coarse) kc.ray_dim = c.casc_row_dim;              // the point MLP's "rays" are the rows (launch_cascade_front)
        hr_launch_mlp_range(kc, ra, st);
        e = hipMemcpyAsync(m->act_max, d_max, sizeof(float) * HR_MAX_LAYERS, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (d_max) (void)hipFree(d_max);
    if (synth && e == hipSuccess) {            // kept (see above)
        free_dev(m->calib_rays);
        m->calib_rays = synth;
        m->calib_n = n;
    } else if (synth) (void)hipFree(synth);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(HR_E_HIP, "activation-range calibration: %s", hipGetErrorString(e)); }
    m->calibrated = synth ? 1 : 2;
    float mx = 0.0f;
    bool finite = true;
    for (int l = 0; l < c.mlp_layers; ++l) {
        if (!std::isfinite(m->act_max[l])) finite = false;
        mx = fmaxf(mx, m->act_max[l]);
    }
    const bool fits = finite && mx < HR_F16_CALIBRATION_LIMIT;
    // the verified fast path: f16f8 + a list-driven second pass in f16x3.  What it needs: a ray's samples inside one wavefront
    // (the list entry is written from a wave-level vote), no cascade (the point MLP's rows are internal).  (An occupancy volume adds a
    // head-dependent decision the band does not cover: hr_render then takes the f16x3 tiles throughout, see hr_render_fields.)
    // The per-sample margins (hr_math.h, HrRisk) are derived for: axis planes, sphere / cylinder with fixed origins, the euclidean distance;
    // the identity, affine and MIP-NeRF contractions.
    const bool isect_ok = c.isect_type == HR_ISECT_Z_PLANE || c.isect_type == HR_ISECT_VOXEL_GRID || c.isect_type == HR_ISECT_EUCLIDEAN_UNIFIED ||
                          ((c.isect_type == HR_ISECT_SPHERE || c.isect_type == HR_ISECT_CYLINDER) && c.origin_scale == 0.0f);
    const bool can_verify = !m->coarse && !m->is_coarse && c.z_channels <= 64 && c.mlp_layers >= 2 && c.mlp_hidden == 256 && isect_ok &&
                            c.contract_type != HR_CONTRACT_DONERF;
    m->verified = 0;
    if (want == HR_MLP_AUTO) {
        m->active_precision = fits ? (can_verify ? HR_MLP_F16F8 : HR_MLP_F16X3) : HR_MLP_BF16X3;
        m->verified = (fits && can_verify) ? 1 : 0;
        return HR_OK;
    }
    if (want == HR_MLP_F16F8V && !can_verify)
        return fail(HR_E_INVALID, "mlp_precision f16f8v (verified) needs a plain ray MLP of width 256, at most 64 samples per ray and an intersection "
                    "the margins are derived for (axis planes, sphere / cylinder with fixed origins, euclidean; not DoNeRFContract)");
    if (!fits)
        return fail(HR_E_RANGE, "mlp_precision %s was requested, but the MLP's activations reach %.4g on the calibration rays (limit %.4g = "
                    "65504 / 8): IEEE-half operands would overflow.  Use HR_MLP_AUTO (falls back to bf16x3) or HR_MLP_BF16X3",
                    want == HR_MLP_F16X3 ? "f16x3" : (want == HR_MLP_F16X2 ? "f16x2" : "f16f8"), (double)mx, (double)HR_F16_CALIBRATION_LIMIT);
    m->active_precision = (want == HR_MLP_F16F8V) ? HR_MLP_F16F8 : want;
    m->verified = (want == HR_MLP_F16F8V) ? 1 : 0;
    return HR_OK;
}

static int calibrate_band(hr_model* m, hipStream_t st);

int hr_model_finalize(hr_model* m)
{
    if (!m) return fail(HR_E_INVALID, "null argument");
    if (m->coarse) {
        int rc = hr_model_finalize(m->coarse);
        if (rc != HR_OK) return rc;
    }
    const hr_config& c = m->cfg;
    for (auto& kv : m->expect)
        if (m->raw.find(kv.first) == m->raw.end())
            return fail(HR_E_MISSING, "tensor '%s%s' was never uploaded", (m->coarse && kv.first.compare(0, 4, "mlp.") == 0) ? "mlp1." : "",
                        (m->coarse && kv.first.compare(0, 4, "mlp.") == 0) ? kv.first.c_str() + 4 : kv.first.c_str());
    m->packed_bytes = 0;
    char name[64];

    // ---- MLP: which arithmetic (the fp16 split needs every activation below 65504), then MFMA operand tiles
    if (!m->flags) HR_HIP(hipMalloc((void**)&m->flags, sizeof(unsigned)));
    HR_HIP(hipMemset(m->flags, 0, sizeof(unsigned)));
    {
        int rc = resolve_precision(m, nullptr, 0, nullptr);
        if (rc != HR_OK) return rc;
        rc = pack_mlp(m);
        if (rc != HR_OK) return rc;
        m->packed_bytes += m->mlp_bytes;
    }

    if (m->is_coarse) {      // coarse level of a cascade: no grids
        HR_HIP(hipDeviceSynchronize());
        m->finalized = true;
        return HR_OK;
    }

    // ---- grids: channel-last texels, density | appearance interleaved per plane pair
    int app_off = 0, real_off = 0;
    for (int j = 0; j < 3; ++j) {
        HrGridPlane& g = m->planes[j];
        g = HrGridPlane();
        free_dev(m->grid_a[j]);
        free_dev(m->grid_b[j]);
        int nd = c.n_den[j], na = c.n_app[j];
        // tensorf_dynamic.py:310-311,355-356: a plane pair whose DENSITY plane has no
        // components is skipped for density and appearance alike
        if (c.video && nd == 0) na = 0;
        g.cd4 = (nd + 3) / 4;
        g.ca4 = (na + 3) / 4;
        g.aw = c.grid[MAT_MODE[j][0]];
        g.ah = c.grid[MAT_MODE[j][1]];
        g.ax = MAT_MODE[j][0];
        g.ay = MAT_MODE[j][1];
        if (c.video) {
            g.bw = c.grid[MAT_MODE_TIME0[j]];
            g.bh = c.num_keyframes;
            g.bx = MAT_MODE_TIME0[j];
        } else {
            g.bw = 1;
            g.bh = c.grid[VEC_MODE[j]];
            g.bx = VEC_MODE[j];
        }
        g.app_off = app_off;
        g.app_real = na;
        g.app_real_off = real_off;
        app_off += 4 * g.ca4;
        real_off += na;
        const int half = (c.grid_dtype == HR_GRID_FP16);
        int tex = 4 * (g.cd4 + g.ca4);
        if (tex == 0) continue;
        if (half) tex = (tex + 7) & ~7;               // whole 16-byte loads of 8 halfs
        g.tex = tex;
        const size_t esz = half ? 2 : sizeof(float);
        const size_t a_bytes = esz * (size_t)g.aw * g.ah * tex;
        const size_t b_bytes = esz * (size_t)g.bw * g.bh * tex;
        // the gathers address texels by 32-bit BYTE offsets (and the class-specialised one marks a masked sample by the offset 0xffffffff)
        if (a_bytes >= ((size_t)1 << 32) || b_bytes >= ((size_t)1 << 32))
            return fail(HR_E_INVALID, "plane pair %d: %zu / %zu bytes -- a feature plane must stay below 4 GiB (32-bit texel offsets)", j, a_bytes, b_bytes);
        HR_HIP(hipMalloc((void**)&m->grid_a[j], a_bytes));
        HR_HIP(hipMalloc((void**)&m->grid_b[j], b_bytes));
        HR_HIP(hipMemset(m->grid_a[j], 0, a_bytes));
        HR_HIP(hipMemset(m->grid_b[j], 0, b_bytes));
        const char* an = c.video ? "plane_space" : "plane";
        const char* bn = c.video ? "plane_time" : "line";
        snprintf(name, sizeof(name), "density_%s.%d", an, j);
        hr_launch_interleave(m->raw[name].p, m->grid_a[j], half, nd, g.ah, g.aw, tex, 0, nullptr);
        snprintf(name, sizeof(name), "app_%s.%d", an, j);
        hr_launch_interleave(m->raw[name].p, m->grid_a[j], half, na, g.ah, g.aw, tex, 4 * g.cd4, nullptr);
        snprintf(name, sizeof(name), "density_%s.%d", bn, j);
        hr_launch_interleave(m->raw[name].p, m->grid_b[j], half, nd, g.bh, g.bw, tex, 0, nullptr);
        snprintf(name, sizeof(name), "app_%s.%d", bn, j);
        hr_launch_interleave(m->raw[name].p, m->grid_b[j], half, na, g.bh, g.bw, tex, 4 * g.cd4, nullptr);
        g.a = m->grid_a[j];
        g.b = m->grid_b[j];
        m->packed_bytes += (int64_t)(a_bytes + b_bytes);
    }
    m->ca_total = app_off;
    // basis_mat columns follow the reference's torch.cat over the sampled planes.  For the
    // video net a skipped plane pair contributes no columns; its n_app must then be 0 too
    // (otherwise the reference itself fails with a shape error in basis_mat).
    int n_app_sum = 0;
    for (int j = 0; j < 3; ++j) n_app_sum += c.n_app[j];
    if (real_off != n_app_sum) return fail(HR_E_INVALID, "video net: n_lamb_sh must be 0 wherever n_lamb_sigma is 0");
    m->n_basis_cols = n_app_sum;
    free_dev(m->basis);
    {
        const size_t bytes = m->raw["basis_mat.weight"].bytes;
        HR_HIP(hipMalloc((void**)&m->basis, bytes > 0 ? bytes : 16));
        if (bytes > 0) HR_HIP(hipMemcpy(m->basis, m->raw["basis_mat.weight"].p, bytes, hipMemcpyDeviceToDevice));
        m->packed_bytes += (int64_t)bytes;
        // column-major copy + the slot -> column map (what hr_fill_decode used to recompute per ray and slot)
        const int AD = c.app_dim, ld = (AD + 3) & ~3;
        std::vector<float> bm((size_t)AD * n_app_sum), bt((size_t)(n_app_sum > 0 ? n_app_sum : 1) * ld, 0.0f);
        if (bytes > 0) HR_HIP(hipMemcpy(bm.data(), m->raw["basis_mat.weight"].p, bytes, hipMemcpyDeviceToHost));
        for (int col = 0; col < n_app_sum; ++col)
            for (int r = 0; r < AD; ++r) bt[(size_t)col * ld + r] = bm[(size_t)r * n_app_sum + col];
        std::vector<int> sc(m->ca_total > 0 ? m->ca_total : 1, -1);
        for (int j = 0; j < 3; ++j)
            if (m->planes[j].ca4 > 0)
                for (int rel = 0; rel < m->planes[j].app_real; ++rel) sc[m->planes[j].app_off + rel] = m->planes[j].app_real_off + rel;
        free_dev(m->basis_t);
        free_dev(reinterpret_cast<float*&>(m->slot_col));
        HR_HIP(hipMalloc((void**)&m->basis_t, bt.size() * sizeof(float)));
        HR_HIP(hipMemcpy(m->basis_t, bt.data(), bt.size() * sizeof(float), hipMemcpyHostToDevice));
        HR_HIP(hipMalloc((void**)&m->slot_col, sc.size() * sizeof(int)));
        HR_HIP(hipMemcpy(m->slot_col, sc.data(), sc.size() * sizeof(int), hipMemcpyHostToDevice));
        m->basis_ld = ld;
    }
    // hr_render_frame: one line per time plane for the frame's blended keyframe rows (float32 texels)
    for (int j = 0; j < 3; ++j) {
        free_dev(m->frame_line[j]);
        const HrGridPlane& p = m->planes[j];
        if (c.video && c.grid_dtype != HR_GRID_FP16 && p.bw > 1 && p.cd4 + p.ca4 > 0)
            HR_HIP(hipMalloc((void**)&m->frame_line[j], sizeof(float) * (size_t)p.bw * p.tex));
    }
    HR_HIP(hipDeviceSynchronize());
    HR_HIP(hipGetLastError());
    m->finalized = true;
    if (m->chunk == 0) {
        // 131072 rays per launch measured best among 16k..640k (DoNeRF: a 185 MB head).  The head of a chunk should still be in the
        // 256 MB Infinity Cache when the sample kernel reads it: wide heads (Neural-3D: 64 samples x 15 columns = 3840 bytes per ray) get
        // fewer rays per launch -- measured on the 800x800 frames (profiles/r04_z_chunk_sweep.txt): neural_3d 4.44 ms at 131 072 rays
        // (503 MB), 4.18 at 65 536 (252 MB), 4.24 at 49 152; the 1920-byte heads (technicolor, immersive: 252 MB at 131 072) are best there
        const int64_t nq = ((int64_t)m->n_out + 3) / 4;
        int64_t rays = (256ll << 20) / (nq * 16 * rows_per_ray(m->cfg));
        if (rays >= 16384) rays &= ~(int64_t)16383;
        // (163 840 = 231 MB of DoNeRF head: the largest that still sits in the cache next to the grids' hot lines -- and with hr_render's even split
        //  an 800x800 frame is 4 launches of 160 000 rays instead of 4 x 131 072 + 115 712: 1.717 vs 1.729 ms, profiles/r06_chunk_sweep.txt; 213 376: 1.824)
        rays = rays > 163840 ? 163840 : (rays < 4096 ? 4096 : rays);
        const int rc = hr_model_reserve(m, rays);
        if (rc != HR_OK) return rc;
    }
    return calibrate_band(m, nullptr);
}

// the configuration with every schedule-dependent constant blanked: what hr_model_update_config may not change
int hr_model_calibrate(hr_model* m, const float* rays_dev, int64_t n_rays, float* act_max, void* stream)
{
    if (!m) return fail(HR_E_INVALID, "null model");
    if (!m->finalized) return fail(HR_E_STATE, "hr_model_calibrate before hr_model_finalize");
    if (m->coarse || m->is_coarse) return fail(HR_E_INVALID, "hr_model_calibrate: cascades are calibrated by hr_model_finalize (the point MLP's rows are internal)");
    if (!rays_dev || n_rays <= 0) return fail(HR_E_INVALID, "hr_model_calibrate needs rays");
    const int before = m->active_precision, calibrated_before = m->calibrated, verified_before = m->verified;
    float act_before[HR_MAX_LAYERS];
    for (int l = 0; l < HR_MAX_LAYERS; ++l) act_before[l] = m->act_max[l];
    int rc = resolve_precision(m, rays_dev, n_rays, (hipStream_t)stream);
    if (rc != HR_OK) {                         // the model stays exactly as it was
        m->active_precision = before;
        m->verified = verified_before;
        m->calibrated = calibrated_before;
        for (int l = 0; l < HR_MAX_LAYERS; ++l) m->act_max[l] = act_before[l];
        return rc;
    }
    HR_HIP(hipMemset(m->flags, 0, sizeof(unsigned)));
    f8_exponents(m);
    if (m->active_precision != before || m->verified != verified_before) {
        m->packed_bytes -= m->mlp_bytes;
        rc = pack_mlp(m);
        if (rc != HR_OK) return rc;
        m->packed_bytes += m->mlp_bytes;
        HR_HIP(hipDeviceSynchronize());
    }
    if (act_max)
        for (int l = 0; l < m->cfg.mlp_layers; ++l) act_max[l] = m->act_max[l];
    m->band_stale = true;
    return calibrate_band(m, (hipStream_t)stream);
}

static hr_config structure_of(const hr_config& in)
{
    hr_config c = in;
    hr_act* acts[] = {&c.f_z_vals.act, &c.f_isect_sigma.act, &c.f_offset_sigma.act, &c.f_point_offset.act, &c.f_color_scale.act,
                      &c.f_color_shift.act, &c.f_spatial_flow.act, &c.f_color_scale_global.act, &c.f_color_shift_global.act,
                      &c.z_act, &c.flow_act, &c.offset_act, &c.color_table_t_act, &c.color_table_s_act};
    for (hr_act* a : acts) { a->outer = 0.0f; a->add = 0.0f; }
    c.isect_mask_off = 0;                          // the near/far mask is dropped after mask.stop_iters (intersect/base.py:104-108)
    for (int g = 0; g < HR_MAX_GROUPS; ++g)
        for (int j = 0; j < HR_MAX_FREQS; ++j) c.groups[g].pe_weight[j] = 0.0f;
    return c;
}

int hr_model_update_config(hr_model* m, const hr_config* cfg, void* stream)
{
    if (!m || !cfg) return fail(HR_E_INVALID, "null argument");
    if (!m->finalized) return fail(HR_E_STATE, "hr_model_update_config before hr_model_finalize");
    if (m->coarse || m->is_coarse) return fail(HR_E_INVALID, "hr_model_update_config: cascades are re-created instead");
    const hr_config a = structure_of(m->cfg), b = structure_of(*cfg);
    if (memcmp(&a, &b, sizeof(hr_config)) != 0)
        return fail(HR_E_INVALID, "hr_model_update_config: the configurations differ in more than activation / PE schedule constants");
    HR_HIP(hipStreamSynchronize((hipStream_t)stream));      // launches in flight still read the device copies
    m->cfg = *cfg;
    analyse_live_columns(m);                                  // same live columns (structure unchanged): rebuilds kcfg
    if (m->kcfg_dev) HR_HIP(hipMemcpy(m->kcfg_dev, &m->kcfg, sizeof(hr_config), hipMemcpyHostToDevice));
    if (m->ucfg_dev) HR_HIP(hipMemcpy(m->ucfg_dev, &m->cfg, sizeof(hr_config), hipMemcpyHostToDevice));
    m->band_stale = true;                                     // the activations' constants feed the distances: measured again before the next render
    return HR_OK;
}

int hr_model_reserve(hr_model* m, int64_t rays_per_chunk)
{
    if (!m) return fail(HR_E_INVALID, "null argument");
    if (rays_per_chunk < 64) rays_per_chunk = 64;
    rays_per_chunk = (rays_per_chunk + 63) & ~(int64_t)63;
    if (rays_per_chunk == m->chunk && m->head) return HR_OK;
    free_dev(m->head);
    free_dev(m->rows);
    m->chunk = 0;
    const size_t n_rows = (size_t)rays_per_chunk * rows_per_ray(m->cfg);      // a multiple of 64
    const size_t nq = ((size_t)samples_per_row(m->cfg) * m->p_live + 3) / 4;
    const size_t bytes = sizeof(float) * n_rows * nq * 4;                        // HQ layout over rows
    HR_HIP(hipMalloc((void**)&m->head, bytes));
    if (m->cfg.mlp_layers == 0) HR_HIP(hipMemset(m->head, 0, bytes));   // ZeroMLP: written once, only ever read
    if (m->coarse) {
        int rc = hr_model_reserve(m->coarse, rays_per_chunk);
        if (rc != HR_OK) return rc;
        HR_HIP(hipMalloc((void**)&m->rows, sizeof(float) * n_rows * m->cfg.casc_row_dim));
    }
    m->chunk = rays_per_chunk;
    // verified fast path: the list of rays the second pass renders again.  The buffer holds 4 M entries (16 MB); a call uses
    // max(32 768, n_rays / 16) of them (measured: 0.01 - 2.5 % of a frame's rays are listed; the calibration gives the fast path up above 5 %)
    // and walks them in slices of the chunk's head workspace.  Beyond that the kernels raise bit 2 of the status word (HR_OPT_REDO_OVERFLOW)
    free_dev(reinterpret_cast<float*&>(m->redo_list));
    free_dev(reinterpret_cast<float*&>(m->wide_list));
    m->redo_cap = 1 << 22;
    m->wide_cap = (int)(rays_per_chunk < 8192 ? rays_per_chunk : 8192);       // third pass: 128 tiles (rays outside the calibrated range are the exception)
    HR_HIP(hipMalloc((void**)&m->redo_list, sizeof(int) * (size_t)m->redo_cap));
    HR_HIP(hipMalloc((void**)&m->wide_list, sizeof(int) * (size_t)m->wide_cap));
    if (!m->redo_count) {
        HR_HIP(hipMalloc((void**)&m->redo_count, 4 * sizeof(unsigned)));
        HR_HIP(hipMemset(m->redo_count, 0, 4 * sizeof(unsigned)));
    }
    return HR_OK;
}

// tier: 0 = the model's primary arithmetic; the verified fast path's later passes: 1 = its f16x3 tiles, 2 = its bf16x3 tiles (fill_mlp_args(..., tier))
static void launch_mlp(const hr_model* m, const hr_config& c, const HrMlpArgs& a, hipStream_t st, int tier = 0)
{
    if (c.mlp_layers == 0) return;               // ZeroMLP: the workspace already holds the (all-zero) head
    const int prec = tier == 1 ? HR_MLP_F16X3 : (tier == 2 ? HR_MLP_BF16X3 : m->active_precision);
    if (prec == HR_MLP_BF16X3) hr_launch_mlp_bf16x3(c, a, st);
    else if (prec == HR_MLP_F16X3) hr_launch_mlp_f16x3(c, a, st);
    else if (prec == HR_MLP_F16X2) hr_launch_mlp_f16x2(c, a, st);
    else if (prec == HR_MLP_F16F8) hr_launch_mlp_f16f8(c, a, st);
    else hr_launch_mlp(c, a, st);
}

static void fill_mlp_args(const hr_model* m, HrMlpArgs& a, const float* rays, int64_t n, int tier = 0)
{
    const bool safe = tier > 0;
    const int ti = tier > 0 ? tier - 1 : 0;
    a.rays = rays;
    a.n_rays = n;
    a.head = m->head;
    for (int l = 0; l < HR_MAX_LAYERS; ++l) {
        a.wpack[l] = m->wpack[l];
        a.wsplit[l] = safe ? m->wsplit_safe[ti][l] : m->wsplit[l];
        a.bias[l] = safe ? m->bias_safe[ti][l] : m->bias[l];
        a.winv[l] = safe ? m->winv_safe[ti][l] : m->winv[l];
        a.xexp[l] = safe ? 0 : m->xexp[l];
        a.n_tiles[l] = safe ? m->n_tiles_safe[ti][l] : m->n_tiles[l];
    }
    a.ray0 = 0;
    a.ray_index = nullptr;
    a.n_rays_dev = nullptr;
    a.list_off = 0;
    a.n_rays_copy = nullptr;
    a.redo_list = nullptr;
    a.redo_count = nullptr;
    a.redo_cap = 0;
    a.n_out = m->n_out;
    a.nq = (m->n_out + 3) / 4;
    a.k0p = m->k0p;
    a.trace = nullptr;
    a.flags = m->flags;
}

// Plane pair j as the render kernels get it.  Inside hr_render_frame on a keyframe net all rays of the call share one time, and that time
// sits on a keyframe row (advect_points quantises it, utils/flow_utils.py:10-35): the time plane is then handed over as the LINE that
// row is -- the gather's line form, 2 taps instead of 4 (the other row's weight is the 1e-7 left by rounding, see hr_render_frame).
static HrGridPlane render_plane(const hr_model* m, int j)
{
    HrGridPlane g = m->planes[j];
    if (m->frame_row >= 0 && m->frame_line[j]) {
        g.b = m->frame_line[j];
        g.bh = g.bw;
        g.bw = 1;
    }
    return g;
}

#ifdef HR_DEBUG_HSUM
static unsigned* g_dbg_hsum = nullptr;
extern "C" void hr_debug_set_hsum(void* p) { g_dbg_hsum = (unsigned*)p; }
#endif
static void fill_sample_args(const hr_model* m, HrSampleArgs& a, const float* rays, int64_t n, float* rgb)
{
    a.cfg_dev = m->kcfg_dev;
    a.rays = rays;
    a.head = m->head;
    a.nq = (m->n_out + 3) / 4;
    a.n_rays = n;
    a.rgb = rgb;
    a.fields = hr_fields();
    for (int j = 0; j < 3; ++j) a.planes[j] = render_plane(m, j);
    a.basis = m->basis;
    a.basis_t = m->basis_t;
    a.slot_col = m->slot_col;
    a.basis_ld = m->basis_ld;
    a.n_basis_cols = m->n_basis_cols;
    a.ca_total = m->ca_total;
    // the table is read in place from the uploaded copy (12 floats per camera, no re-layout)
    a.color_table = nullptr;
    if (m->cfg.color_table_views > 0) {
        auto it = m->raw.find("color_embedding");
        if (it != m->raw.end()) a.color_table = it->second.p;
    }
    a.dbg_mode = 0;
    a.ray0 = 0;
    a.ray_index = nullptr;
    a.n_rays_dev = nullptr;
    a.list_off = 0;
    a.zero_word = nullptr;
    a.redo_list = nullptr;
    a.redo_count = nullptr;
    a.redo_cap = 0;
    a.redo_band = a.redo_band_q = a.redo_band_off = a.redo_amp_cut = 0.0f;
    a.flags = m->flags;
#ifdef HR_DEBUG_HSUM
    a.dbg_hsum = g_dbg_hsum;
#endif
    a.occ = m->occ;
    a.occ_cells = m->occ_cells;
    a.occ_w = m->occ_n[0]; a.occ_h = m->occ_n[1]; a.occ_d = m->occ_n[2];
    for (int i = 0; i < 3; ++i) { a.occ_lo[i] = m->occ_lo[i]; a.occ_inv[i] = m->occ_inv[i]; }
    a.rows_per_ray = rows_per_ray(m->cfg);
    a.rows_out = nullptr;
    a.row_dim = a.n_row_inputs = 0;
    for (int i = 0; i < 4; ++i) a.row_kind[i] = a.row_len[i] = 0;
}

// Cascade, everything before the final sample kernel: coarse MLP -> coarse intersect (emits the point MLP's input
// rows, one per coarse sample) -> point MLP over n * casc_in_z rows.  Leaves the fine head in m->head.
static void launch_cascade_front(hr_model* m, const float* rays, int64_t n, hipStream_t st)
{
    hr_model* c0 = m->coarse;
    HrMlpArgs ma;
    fill_mlp_args(c0, ma, rays, n);
    launch_mlp(c0, c0->kcfg, ma, st);
    HrSampleArgs sa;
    fill_sample_args(c0, sa, rays, n, nullptr);
    sa.rows_out = m->rows;
    sa.row_dim = m->cfg.casc_row_dim;
    sa.n_row_inputs = m->cfg.casc_n_inputs;
    for (int i = 0; i < 4; ++i) { sa.row_kind[i] = m->cfg.casc_input_kind[i]; sa.row_len[i] = m->cfg.casc_input_dim[i]; }
    hr_launch_samples(c0->kcfg, sa, st);
    hr_config kc = m->kcfg;
    kc.ray_dim = m->cfg.casc_row_dim;            // the point MLP's "rays" are the rows
    HrMlpArgs mb;
    fill_mlp_args(m, mb, m->rows, n * m->cfg.casc_in_z);
    launch_mlp(m, kc, mb, st);
}

// redo0 >= 0: first pass of the verified fast path -- tiles that raise a range bit list their rays (indices start at redo0);
// safe: the whole launch with the f16x3 tiles (hr_render_fields with diagnostics: one arithmetic for every output)
static void launch_front(hr_model* m, const float* rays, int64_t n, hipStream_t st, int64_t redo0 = -1, int tier = 0)
{
    if (m->coarse) {
        launch_cascade_front(m, rays, n, st);
        return;
    }
    HrMlpArgs ma;
    fill_mlp_args(m, ma, rays, n, tier);
    if (redo0 >= 0) {
        ma.ray0 = redo0;
        ma.redo_list = m->redo_list;
        ma.redo_count = m->redo_count;
        ma.redo_cap = m->redo_cap;
    }
    launch_mlp(m, m->kcfg, ma, st, tier);
}

// The frame kernel (fused_impl.inc) for the whole ray list; false: the model does not fit it (nothing launched)
static bool launch_frame(hr_model* m, const float* rays, int64_t n, float* rgb, bool probe, hipStream_t st)
{
    if (!m->opt_frame_kernel || m->coarse || m->is_coarse || m->cfg.mlp_layers == 0) return false;
    if (m->verified) return false;               // the verified fast path is a two-pass plan over the HBM workspace
    if (n > ((int64_t)1 << 36)) return false;
    HrMlpArgs ma;
    fill_mlp_args(m, ma, rays, n);
    ma.head = nullptr;
    HrSampleArgs sa;
    fill_sample_args(m, sa, rays, n, rgb);
    sa.head = nullptr;
    switch (m->active_precision) {
        case HR_MLP_BF16X3: return hr_launch_frame_bf16x3(m->kcfg, ma, sa, m->opt_sample_waves, m->opt_frame_kernel, m->n_cus, probe, st);
        case HR_MLP_F16X3: return hr_launch_frame_f16x3(m->kcfg, ma, sa, m->opt_sample_waves, m->opt_frame_kernel, m->n_cus, probe, st);
        case HR_MLP_F16X2: return hr_launch_frame_f16x2(m->kcfg, ma, sa, m->opt_sample_waves, m->opt_frame_kernel, m->n_cus, probe, st);
        case HR_MLP_F16F8: return hr_launch_frame_f16f8(m->kcfg, ma, sa, m->opt_sample_waves, m->opt_frame_kernel, m->n_cus, probe, st);
        default: return false;          // the exact-fp32 MLP (v_mfma_f32_16x16x4_f32) keeps its own kernel
    }
}

static int check_render(const hr_model* m, const float* rays, int64_t n, const float* rgb)
{
    if (!m) return fail(HR_E_INVALID, "null model");
    if (!m->finalized) return fail(HR_E_STATE, "hr_model_finalize has not been called");
    if (n < 0) return fail(HR_E_INVALID, "negative ray count");
    if (n > 0 && (!rays || !rgb)) return fail(HR_E_INVALID, "null ray / rgb buffer");
    return HR_OK;
}

// rays per launch of a call of n rays: as many launches as the workspace demands, of equal size (a short last launch leaves the chip half empty
// for a whole kernel)
static int64_t even_chunk(const hr_model* m, int64_t n)
{
    if (n <= m->chunk) return m->chunk;
    const int64_t k = (n + m->chunk - 1) / m->chunk;
    const int64_t per = (((n + k - 1) / k) + 63) & ~(int64_t)63;
    return per < m->chunk ? per : m->chunk;
}

// The verified fast path over one call's rays: first pass in f16f8 with the rays at risk listed on the device, then the list
// again with the f16x3 tiles (in slices of the chunk's head workspace), then whatever left the half range there with the bf16x3 tiles.
// list_cap: entries of the list this call may use.
static void render_verified(hr_model* m, const float* rays_dev, int64_t n_rays, float* rgb_dev, int list_cap, hipStream_t st)
{
    const hr_config& c = m->cfg;
    const int64_t per = even_chunk(m, n_rays);
    for (int64_t r0 = 0; r0 < n_rays; r0 += per) {
        const int64_t n = (n_rays - r0 < per) ? (n_rays - r0) : per;
        const float* rays = rays_dev + r0 * c.ray_dim;
        HrMlpArgs ma;
        fill_mlp_args(m, ma, rays, n, 0);
        ma.ray0 = r0;                                  // tiles that raise a range bit list their rays (indices start at r0)
        ma.redo_list = m->redo_list;
        ma.redo_count = m->redo_count;
        ma.redo_cap = list_cap;
        launch_mlp(m, m->kcfg, ma, st, 0);
        HrSampleArgs sa;
        fill_sample_args(m, sa, rays, n, rgb_dev + r0 * 3);
        sa.ray0 = r0;
        sa.redo_list = m->redo_list;
        sa.redo_count = m->redo_count;
        sa.redo_cap = list_cap;
        sa.redo_band = m->redo_band;
        sa.redo_band_q = m->redo_band_q;
        sa.redo_band_off = m->redo_band_off;
        sa.redo_amp_cut = HR_VERIFY_AMP_CUT;
        hr_launch_samples(m->kcfg, sa, st);
    }
    // second pass: the listed rays (count on the device: the launches are sized for the capacity, blocks past the count leave at once) through
    // the f16x3 tiles, gathered from / scattered to the caller's buffers by index.  The head workspace is free again; a list longer than it is
    // walked in slices.  The counter is cleared for the next call by the FIRST slice's sample kernel, which like every later launch of the
    // pass reads the copy the first slice's MLP kernel made (a memset node between calls does not survive hipGraph replay)
    for (int64_t off = 0; off < list_cap; off += m->chunk) {
        const int64_t cap = (list_cap - off < m->chunk) ? (list_cap - off) : m->chunk;
        HrMlpArgs ma;
        fill_mlp_args(m, ma, rays_dev, cap, 1);
        ma.ray_index = m->redo_list + off;
        ma.list_off = off;
        ma.n_rays_dev = off == 0 ? m->redo_count : m->redo_count + 1;
        ma.n_rays_copy = off == 0 ? m->redo_count + 1 : nullptr;
        ma.redo_list = m->wide_list;                   // a tile of THIS pass in which an activation leaves the half range goes on to the third
        ma.redo_count = m->redo_count + 2;
        ma.redo_cap = m->wide_cap;
        launch_mlp(m, m->kcfg, ma, st, 1);
        HrSampleArgs sa;
        fill_sample_args(m, sa, rays_dev, cap, rgb_dev);
        sa.ray_index = m->redo_list + off;
        sa.list_off = off;
        sa.n_rays_dev = m->redo_count + 1;
        sa.zero_word = off == 0 ? m->redo_count : nullptr;
        hr_launch_samples(m->kcfg, sa, st);
    }
    // third pass: those tiles' rays with the bf16x3 tiles -- halves with the fp32 exponent range, nothing to overflow.  What a captured
    // viewer loop gets where the host's guard (models.py: a sticky bit read between calls) cannot reach
    HrMlpArgs ma;
    fill_mlp_args(m, ma, rays_dev, m->wide_cap, 2);
    ma.ray_index = m->wide_list;
    ma.n_rays_dev = m->redo_count + 2;
    ma.n_rays_copy = m->redo_count + 3;
    launch_mlp(m, m->kcfg, ma, st, 2);
    HrSampleArgs sa;
    fill_sample_args(m, sa, rays_dev, m->wide_cap, rgb_dev);
    sa.ray_index = m->wide_list;
    sa.n_rays_dev = m->redo_count + 3;
    sa.zero_word = m->redo_count + 2;
    hr_launch_samples(m->kcfg, sa, st);
}

// entries of the ray list one hr_render call may fill: a sixteenth of its rays, at least 32 768 (never more than the rays there are, or the buffer).
// The second pass's launches are sized for it -- ~1.7 ns per workgroup that finds nothing to do -- and the calibration gives the fast path up
// above a twentieth (HR_VERIFY_LISTED_LIMIT)
static int redo_list_cap(const hr_model* m, int64_t n_rays)
{
    int64_t cap = n_rays / 16 > 32768 ? n_rays / 16 : 32768;
    cap = (cap + 63) & ~(int64_t)63;
    if (cap > n_rays) cap = (n_rays + 63) & ~(int64_t)63;
    return (int)(cap < m->redo_cap ? cap : m->redo_cap);
}

// The margins of the verified fast path for THIS model: both arithmetics' heads on the calibration rays, pushed through the
// model's own activations, anchors, contraction and intersection by the probe kernel (band_kernel.hip) in the normalisation the sample
// stage's per-sample margins use (hr_math.h, HrRisk); margin = 4 x the largest difference, never below HR_BAND_FLOOR.  Then the
// well-conditioned calibration rays once through the verified path and once through the f16x3 tiles: the fraction listed, and how far the
// two images are apart.  HR_MLP_AUTO gives the fast path up (f16x3 throughout) above 5 % / 6e-5.  Synchronises `st`.
static int calibrate_band(hr_model* m, hipStream_t st)
{
    m->band_stale = false;
    hr_verify_info& vi = m->vinfo;
    const int fallback_before = vi.fallback;
    vi = hr_verify_info();
    vi.verified = m->verified;
    vi.band_floor = HR_BAND_FLOOR;
    vi.fallback = (fallback_before && !m->verified) ? fallback_before : 0;
    if (!m->verified || m->coarse || m->is_coarse) return HR_OK;
    if (!m->calib_rays || m->calib_n <= 0 || !m->head) return fail(HR_E_STATE, "verified fast path without calibration rays / workspace");
    const hr_config& c = m->cfg;
    const int Z = c.z_channels, P = c.preds_per_z;
    const int64_t N = m->calib_n, nc_max = N < m->chunk ? N : m->chunk;
    float *ha = nullptr, *hb = nullptr, *rgb = nullptr, *rgb2 = nullptr, *sel = nullptr;
    unsigned* stats = nullptr;
    unsigned char* ok_dev = nullptr;
    hr_config* pcfg = nullptr;
    auto cleanup = [&]() {
        if (ha) (void)hipFree(ha);
        if (hb) (void)hipFree(hb);
        if (rgb) (void)hipFree(rgb);
        if (rgb2) (void)hipFree(rgb2);
        if (sel) (void)hipFree(sel);
        if (stats) (void)hipFree(stats);
        if (ok_dev) (void)hipFree(ok_dev);
        if (pcfg) (void)hipFree(pcfg);
    };
#define HR_BAND_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); return fail(HR_E_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); } } while (0)
    HR_BAND_HIP(hipMalloc((void**)&ha, sizeof(float) * nc_max * Z * P));
    HR_BAND_HIP(hipMalloc((void**)&hb, sizeof(float) * nc_max * Z * P));
    HR_BAND_HIP(hipMalloc((void**)&stats, sizeof(unsigned) * HR_BAND_WORDS));
    HR_BAND_HIP(hipMalloc((void**)&ok_dev, (size_t)N));
    HR_BAND_HIP(hipMalloc((void**)&pcfg, sizeof(hr_config)));
    HR_BAND_HIP(hipMemsetAsync(stats, 0, sizeof(unsigned) * HR_BAND_WORDS, st));
    HR_BAND_HIP(hipMemsetAsync(ok_dev, 1, (size_t)N, st));
    hr_config probe_cfg = c;                       // user column order; the probe applies the near / far mask itself
    probe_cfg.isect_mask_off = 1;
    HR_BAND_HIP(hipMemcpyAsync(pcfg, &probe_cfg, sizeof(hr_config), hipMemcpyHostToDevice, st));
    HR_BAND_HIP(hipStreamSynchronize(st));          // (probe_cfg is a local)
    for (int64_t r0 = 0; r0 < N; r0 += nc_max) {
        const int64_t n = (N - r0 < nc_max) ? (N - r0) : nc_max;
        const float* rays = m->calib_rays + r0 * c.ray_dim;
        for (int tier = 0; tier < 2; ++tier) {
            HrMlpArgs ma;
            fill_mlp_args(m, ma, rays, n, tier);
            launch_mlp(m, m->kcfg, ma, st, tier);
            hr_launch_head_export(m->head, tier == 0 ? ha : hb, n, Z, P, m->p_live, (m->n_out + 3) / 4, rows_per_ray(c), m->col_map, st);
        }
        HrBandArgs ba;
        ba.cfg_dev = pcfg;
        ba.rays = rays;
        ba.head_a = ha;
        ba.head_b = hb;
        ba.n_rays = n;
        ba.mask_on = c.isect_mask_off ? 0 : 1;
        ba.flip_cut = 1e-3f;
        ba.stats = stats;
        ba.ray_ok = ok_dev + r0;
        ba.amp_cut = HR_VERIFY_AMP_CUT;
        ba.phase = 0;                              // which rays are well conditioned ...
        hr_launch_band_probe(ba, Z, st);
        ba.phase = 1;                              // ... and the statistics over those
        hr_launch_band_probe(ba, Z, st);
    }
    unsigned hs[HR_BAND_WORDS];
    std::vector<unsigned char> ok((size_t)N);
    std::vector<float> rays_h((size_t)N * c.ray_dim);
    HR_BAND_HIP(hipMemcpyAsync(hs, stats, sizeof(hs), hipMemcpyDeviceToHost, st));
    HR_BAND_HIP(hipMemcpyAsync(ok.data(), ok_dev, (size_t)N, hipMemcpyDeviceToHost, st));
    HR_BAND_HIP(hipMemcpyAsync(rays_h.data(), m->calib_rays, sizeof(float) * rays_h.size(), hipMemcpyDeviceToHost, st));
    HR_BAND_HIP(hipStreamSynchronize(st));
    auto f = [&](int i) { float v; memcpy(&v, &hs[i], sizeof(v)); return v; };
    vi.max_d_zc = f(HR_BAND_ZC);
    vi.max_d_dist_n = f(HR_BAND_DIST_N);
    vi.max_d_geo_n = f(HR_BAND_GEO_N);
    vi.max_d_off = f(HR_BAND_OFF);
    vi.max_d_dist = f(HR_BAND_DIST);
    for (int i = 0; i < 64; ++i) vi.max_d_head = fmaxf(vi.max_d_head, f(HR_BAND_HEAD0 + i));
    vi.n_rays = N;
    vi.n_samples = hs[HR_BAND_COUNTED];
    vi.n_flipped = hs[HR_BAND_FLIPPED];
    vi.n_shaky = hs[HR_BAND_SHAKY];
    m->redo_band = fmaxf(HR_BAND_FLOOR, 4.0f * fmaxf(vi.max_d_zc, vi.max_d_dist_n));
    m->redo_band_q = fmaxf(HR_BAND_FLOOR, 4.0f * vi.max_d_geo_n);
    m->redo_band_off = 4.0f * vi.max_d_off;
    vi.band = m->redo_band;
    vi.band_q = m->redo_band_q;
    vi.band_off = m->redo_band_off;
    // the well-conditioned rays, compacted on the host (<= 65 536 rays: not worth a kernel)
    int64_t nu = 0;
    for (int64_t i = 0; i < N; ++i)
        if (ok[(size_t)i]) {
            if (nu != i) memcpy(&rays_h[(size_t)nu * c.ray_dim], &rays_h[(size_t)i * c.ray_dim], sizeof(float) * c.ray_dim);
            ++nu;
        }
    vi.n_rays_used = nu;
    // (fewer than 64 well-conditioned rays -- a caller calibrating on a handful: the margins stand, there is no image to judge by)
    bool too_many = false, too_far = false;
    if (nu >= 64) {
        // through the verified path as a render call takes it, and through the f16x3 tiles throughout: what fraction the first pass lists with
        // these margins, and how far the two IMAGES are apart -- f16f8's continuous error (every pixel that is not listed keeps it) grows with
        // the weights like the margins do, and no margin repairs it
        HR_BAND_HIP(hipMalloc((void**)&sel, sizeof(float) * nu * c.ray_dim));
        HR_BAND_HIP(hipMalloc((void**)&rgb, sizeof(float) * nu * 3));
        HR_BAND_HIP(hipMalloc((void**)&rgb2, sizeof(float) * nu * 3));
        HR_BAND_HIP(hipMemcpyAsync(sel, rays_h.data(), sizeof(float) * nu * c.ray_dim, hipMemcpyHostToDevice, st));
        HR_BAND_HIP(hipMemsetAsync(m->redo_count, 0, 4 * sizeof(unsigned), st));
        render_verified(m, sel, nu, rgb, redo_list_cap(m, nu), st);
        unsigned listed = 0;
        HR_BAND_HIP(hipMemcpyAsync(&listed, m->redo_count + 1, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        for (int64_t r0 = 0; r0 < nu; r0 += m->chunk) {
            const int64_t n = (nu - r0 < m->chunk) ? (nu - r0) : m->chunk;
            const float* rays = sel + r0 * c.ray_dim;
            launch_front(m, rays, n, st, -1, 1);
            HrSampleArgs sa;
            fill_sample_args(m, sa, rays, n, rgb2 + r0 * 3);
            hr_launch_samples(m->kcfg, sa, st);
        }
        std::vector<float> ia((size_t)nu * 3), ib((size_t)nu * 3);
        HR_BAND_HIP(hipMemcpyAsync(ia.data(), rgb, sizeof(float) * ia.size(), hipMemcpyDeviceToHost, st));
        HR_BAND_HIP(hipMemcpyAsync(ib.data(), rgb2, sizeof(float) * ib.size(), hipMemcpyDeviceToHost, st));
        HR_BAND_HIP(hipStreamSynchronize(st));
        for (size_t i = 0; i < ia.size(); ++i) {
            const float d = fabsf(ia[i] - ib[i]);
            vi.max_d_rgb = (d > vi.max_d_rgb || d != d) ? d : vi.max_d_rgb;
        }
        vi.listed_frac = (float)((double)listed / (double)nu);
        too_far = !(vi.max_d_rgb <= HR_VERIFY_RGB_LIMIT);
    }
    // the CALLER's rays are what will be rendered: the ill-conditioned ones among them are listed too (synthetic rays point anywhere;
    // half of them graze a z-plane net's planes, which says nothing about its cameras)
    if (m->calibrated == 2) vi.listed_frac = (float)(((double)(N - nu) + (double)vi.listed_frac * (double)nu) / (double)N);
    too_many = vi.listed_frac > HR_VERIFY_LISTED_LIMIT;
    HR_BAND_HIP(hipMemsetAsync(m->redo_count, 0, 4 * sizeof(unsigned), st));
    HR_BAND_HIP(hipMemsetAsync(m->flags, 0, sizeof(unsigned), st));        // range bits the calibration rays raised are not the caller's
    HR_BAND_HIP(hipStreamSynchronize(st));
#undef HR_BAND_HIP
    cleanup();
    if ((too_many || too_far) && m->cfg.mlp_precision == HR_MLP_AUTO) {
        // more than a twentieth of the rays would be rendered twice (a call's list holds a sixteenth), or the cheap arithmetic's own error is
        // too large a share of the 1e-4 budget on this model: plain f16x3 lists nothing and has neither problem
        m->verified = 0;
        m->active_precision = HR_MLP_F16X3;
        m->packed_bytes -= m->mlp_bytes;
        const int rc = pack_mlp(m);
        if (rc != HR_OK) return rc;
        m->packed_bytes += m->mlp_bytes;
        HR_HIP(hipDeviceSynchronize());
        m->band_stale = false;
        vi.verified = 0;
        vi.fallback = too_far ? 2 : 1;
    }
    return HR_OK;
}

int hr_model_verify_info(hr_model* m, hr_verify_info* out)
{
    if (!m || !out) return fail(HR_E_INVALID, "null argument");
    if (!m->finalized) return fail(HR_E_STATE, "hr_model_finalize has not been called");
    *out = m->vinfo;
    out->verified = m->verified;
    return HR_OK;
}

int hr_render_fields(hr_model* m, const float* rays_dev, int64_t n_rays, float* rgb_dev, const hr_fields* fields, void* stream)
{
    int rc = check_render(m, rays_dev, n_rays, rgb_dev);
    if (rc != HR_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const hr_config& c = m->cfg;
    const int Z = c.z_channels;
    if (!fields && launch_frame(m, rays_dev, n_rays, rgb_dev, false, st)) {
        HR_HIP(hipGetLastError());
        return HR_OK;
    }
    // (Running the sample stage of chunk i on a second stream under the MLP of chunk i+1 was
    //  measured twice -- plain, and with the MLP limited to one workgroup per CU so that sample
    //  blocks could co-reside -- and is slower than back-to-back launches: 3.0-3.9 vs 2.79 ms per
    //  800x800 frame; the two kernels do not interleave on the CUs.)
    // verified fast path.  With diagnostics requested every output comes from ONE arithmetic: the f16x3 tiles throughout.
    // So does a model with an occupancy volume (hr_occupancy_test decides per cell from a head-dependent point; the band does not cover it),
    // and a render inside a stream capture whose band is out of date (hr_model_update_config since the last measurement: measuring synchronises).
    bool verify = m->verified && !fields && !m->occ && n_rays < ((int64_t)1 << 31);
    if (verify && m->band_stale) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs == hipStreamCaptureStatusNone) {
            rc = calibrate_band(m, st);
            if (rc != HR_OK) return rc;
            verify = verify && m->verified;            // HR_MLP_AUTO may just have given the fast path up
        } else {
            verify = false;
        }
    }
    if (verify && n_rays > 0) {
        render_verified(m, rays_dev, n_rays, rgb_dev, redo_list_cap(m, n_rays), st);
        HR_HIP(hipGetLastError());
        return HR_OK;
    }
    const bool safe_all = m->verified != 0;
    const int64_t per = even_chunk(m, n_rays);
    for (int64_t r0 = 0; r0 < n_rays; r0 += per) {
        const int64_t n = (n_rays - r0 < per) ? (n_rays - r0) : per;
        const float* rays = rays_dev + r0 * c.ray_dim;
        launch_front(m, rays, n, st, -1, safe_all ? 1 : 0);
        HrSampleArgs sa;
        fill_sample_args(m, sa, rays, n, rgb_dev + r0 * 3);
        if (fields) {
            if (fields->distances_dev) sa.fields.distances_dev = fields->distances_dev + r0 * Z;
            if (fields->points_dev) sa.fields.points_dev = fields->points_dev + r0 * Z * 3;
            if (fields->sigma_dev) sa.fields.sigma_dev = fields->sigma_dev + r0 * Z;
            if (fields->weights_dev) sa.fields.weights_dev = fields->weights_dev + r0 * Z;
            if (fields->head_dev)
                hr_launch_head_export(m->head, fields->head_dev + r0 * (int64_t)Z * c.preds_per_z, n, Z, c.preds_per_z, m->p_live,
                                      (m->n_out + 3) / 4, rows_per_ray(c), m->col_map, st);
        }
        hr_launch_samples(m->kcfg, sa, st);
    }
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int hr_render(hr_model* m, const float* rays_dev, int64_t n_rays, float* rgb_dev, void* stream)
{
    return hr_render_fields(m, rays_dev, n_rays, rgb_dev, nullptr, stream);
}

int hr_render_frame(hr_model* m, const float* rays_dev, int64_t n_rays, float time, float* rgb_dev, void* stream)
{
    if (!m) return fail(HR_E_INVALID, "null model");
    const hr_config& c = m->cfg;
    hipStream_t st = (hipStream_t)stream;
    m->frame_row = -1;
    if (c.video && c.num_keyframes >= 2 && !m->coarse && !m->is_coarse && c.grid_dtype != HR_GRID_FP16 && m->finalized) {
        // the time tap of every ray of the frame, as hr_sample_body computes it from the ray's last column (host restatement of
        // hr_base_time, hr_normalize_time and hr_make_tap, csrc/hr_math.h; float32 throughout)
        float base_t = 0.0f;
        if (c.advect) {
            float tt = time * c.flow_fac;
            tt = fminf(fmaxf(tt, 0.0f), c.flow_kmax);
            base_t = rintf(tt - 1e-5f) * c.flow_inv_fac;
        }
        const float g = (base_t * c.time_scale + c.time_offset) * 2.0f - 1.0f;
        const int n = c.num_keyframes;
        const float ix = ((g + 1.0f) / 2.0f) * (float)(n - 1);
        const float f0 = floorf(ix), f1 = f0 + 1.0f;
        const int i0 = (int)f0, i1 = i0 + 1;
        const bool ok0 = i0 >= 0 && i0 < n, ok1 = i1 >= 0 && i1 < n;
        const float w0 = ok0 ? f1 - ix : 0.0f, w1 = ok1 ? ix - f0 : 0.0f;
        for (int j = 0; j < 3; ++j) {
            const HrGridPlane& p = m->planes[j];
            if (p.bw <= 1 || p.cd4 + p.ca4 == 0) continue;
            const int row_floats = p.bw * p.tex;
            if (!m->frame_line[j]) continue;
            hr_launch_blend_rows(reinterpret_cast<const float*>(p.b), m->frame_line[j], row_floats, ok0 ? i0 : 0, ok1 ? i1 : 0, w0, w1, st);
            m->frame_row = 0;
        }
    }
    const int rc = hr_render_fields(m, rays_dev, n_rays, rgb_dev, nullptr, stream);
    m->frame_row = -1;
    return rc;
}

int hr_model_set_occupancy(hr_model* m, const float* volume_dev, const int32_t n[3], const float aabb[6], void* stream)
{
    if (!m) return fail(HR_E_INVALID, "null model");
    if (m->is_coarse) return fail(HR_E_INVALID, "the coarse level of a cascade has no colour net");
    HR_HIP(hipStreamSynchronize((hipStream_t)stream));          // launches in flight may still read the old volume
    free_dev(m->occ);
    free_dev(reinterpret_cast<float*&>(m->occ_cells));
    if (!volume_dev) return HR_OK;
    if (!n || !aabb || n[0] < 1 || n[1] < 1 || n[2] < 1) return fail(HR_E_INVALID, "occupancy volume without a size / box");
    for (int i = 0; i < 3; ++i)
        if (!(aabb[3 + i] > aabb[i])) return fail(HR_E_INVALID, "empty occupancy box");
    const size_t bytes = sizeof(float) * (size_t)n[0] * n[1] * n[2];
    HR_HIP(hipMalloc((void**)&m->occ, bytes));
    HR_HIP(hipMemcpy(m->occ, volume_dev, bytes, hipMemcpyDefault));
    for (int i = 0; i < 3; ++i) {
        m->occ_n[i] = n[i];
        m->occ_lo[i] = aabb[i];
        m->occ_inv[i] = (1.0f / (aabb[3 + i] - aabb[i])) * 2.0f;        // AlphaGridMask: invgridSize = 1.0 / aabbSize * 2
    }
    // cell table: a 0/1 volume (what updateAlphaMask stores) sampled strictly inside a lattice cell is > 0 exactly when one of
    // the cell's 8 corners is set
    if (n[0] > 1 && n[1] > 1 && n[2] > 1) {
        const size_t W = n[0], H = n[1], D = n[2];
        std::vector<float> v(W * H * D);
        HR_HIP(hipMemcpy(v.data(), m->occ, bytes, hipMemcpyDeviceToHost));
        bool binary = true;
        for (float x : v) if (x != 0.0f && x != 1.0f) { binary = false; break; }
        if (binary) {
            const size_t cells = (W - 1) * (H - 1) * (D - 1);
            std::vector<unsigned> bits((cells + 31) / 32, 0u);
            for (size_t z = 0; z + 1 < D; ++z)
                for (size_t y = 0; y + 1 < H; ++y)
                    for (size_t x = 0; x + 1 < W; ++x) {
                        bool any = false;
                        for (int c = 0; c < 8 && !any; ++c) any = v[((z + (c >> 2)) * H + y + ((c >> 1) & 1)) * W + x + (c & 1)] != 0.0f;
                        if (any) {
                            const size_t cell = (z * (H - 1) + y) * (W - 1) + x;
                            bits[cell >> 5] |= 1u << (cell & 31);
                        }
                    }
            HR_HIP(hipMalloc((void**)&m->occ_cells, bits.size() * sizeof(unsigned)));
            HR_HIP(hipMemcpy(m->occ_cells, bits.data(), bits.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        }
    }
    return HR_OK;
}

int hr_model_set_option(hr_model* m, int32_t option, int32_t value)
{
    if (!m) return fail(HR_E_INVALID, "null model");
    if (option == HR_OPT_FRAME_KERNEL) {
        if (value < 0 || value > 2) return fail(HR_E_INVALID, "HR_OPT_FRAME_KERNEL takes 0, 1 or 2");
        m->opt_frame_kernel = value;
    } else if (option == HR_OPT_TRAIN_DETERMINISTIC) {
        if (value != 0 && value != 1) return fail(HR_E_INVALID, "HR_OPT_TRAIN_DETERMINISTIC takes 0 or 1");
        m->opt_train_det = value;
    } else if (option == HR_OPT_SAMPLE_WAVES) {
        if (value != 0 && value != 4 && value != 8) return fail(HR_E_INVALID, "HR_OPT_SAMPLE_WAVES takes 0 (the plan's default), 4 or 8");
        m->opt_sample_waves = value;
    } else {
        return fail(HR_E_INVALID, "unknown or read-only option %d", option);
    }
    return HR_OK;
}

int hr_model_get_option(hr_model* m, int32_t option, int32_t* value)
{
    if (!m || !value) return fail(HR_E_INVALID, "null argument");
    if (option == HR_OPT_FRAME_KERNEL) *value = m->opt_frame_kernel;
    else if (option == HR_OPT_SAMPLE_WAVES) *value = m->opt_sample_waves;
    else if (option == HR_OPT_TRAIN_DETERMINISTIC) *value = m->opt_train_det;
    else if (option == HR_OPT_CHUNK_RAYS) *value = (int32_t)m->chunk;
    else if (option == HR_OPT_MLP_PRECISION_ACTIVE || option == HR_OPT_MLP_CALIBRATED || option == HR_OPT_MLP_OVERFLOW || option == HR_OPT_MLP_F8_SATURATED ||
             option == HR_OPT_MLP_VERIFIED || option == HR_OPT_REDO_OVERFLOW || option == HR_OPT_REDO_COUNT || option == HR_OPT_WIDE_COUNT) {
        if (!m->finalized) return fail(HR_E_STATE, "hr_model_finalize has not been called");
        if (option == HR_OPT_MLP_PRECISION_ACTIVE) *value = m->active_precision;
        else if (option == HR_OPT_MLP_CALIBRATED) *value = m->calibrated;
        else if (option == HR_OPT_MLP_VERIFIED) *value = m->verified;
        else if (option == HR_OPT_REDO_COUNT || option == HR_OPT_WIDE_COUNT) {
            unsigned n = 0;
            if (m->redo_count) HR_HIP(hipMemcpy(&n, m->redo_count + (option == HR_OPT_REDO_COUNT ? 1 : 3), sizeof(unsigned), hipMemcpyDeviceToHost));      // the pass's copy
            *value = (int32_t)n;
        } else if (option == HR_OPT_REDO_OVERFLOW) {
            unsigned f = 0;
            HR_HIP(hipMemcpy(&f, m->flags, sizeof(unsigned), hipMemcpyDeviceToHost));
            *value = (int32_t)((f >> 2) & 1u);
        } else {
            unsigned f = 0;
            HR_HIP(hipMemcpy(&f, m->flags, sizeof(unsigned), hipMemcpyDeviceToHost));
            if (m->coarse) {
                unsigned g = 0;
                HR_HIP(hipMemcpy(&g, m->coarse->flags, sizeof(unsigned), hipMemcpyDeviceToHost));
                f |= g;
            }
            *value = (int32_t)(option == HR_OPT_MLP_OVERFLOW ? (f & 1u) : ((f >> 1) & 1u));
        }
    } else if (option == HR_OPT_FRAME_KERNEL_ACTIVE) {
        if (!m->finalized) return fail(HR_E_STATE, "hr_model_finalize has not been called");
        *value = launch_frame(m, nullptr, 64, nullptr, true, nullptr) ? 1 : 0;
    } else return fail(HR_E_INVALID, "unknown option %d", option);
    return HR_OK;
}

int hr_shard_range(int64_t n_pixels, int32_t rank, int32_t world, int64_t* first, int64_t* count)
{
    if (!first || !count || world < 1 || rank < 0 || rank >= world || n_pixels < 0) return fail(HR_E_INVALID, "hr_shard_range: bad arguments");
    const int64_t base = n_pixels / world, extra = n_pixels % world;
    *first = (int64_t)rank * base + (rank < extra ? rank : extra);
    *count = base + (rank < extra ? 1 : 0);
    return HR_OK;
}

int hr_allgather_tiles(void* nccl_comm, const float* tile_dev, float* full_dev, int64_t floats_per_rank, void* stream)
{
    if (!nccl_comm || !tile_dev || !full_dev || floats_per_rank <= 0) return fail(HR_E_INVALID, "hr_allgather_tiles: null communicator / buffer or empty tile");
    // ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
    typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
    static allgather_fn fn = nullptr;
    static bool looked = false;
    if (!looked) {
        looked = true;
        void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather");        // the RCCL the process already uses (torch.distributed's, the integrator's)
        if (!sym) {
            for (const char* name : {"librccl.so.1", "librccl.so"}) {
                void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h && (sym = dlsym(h, "ncclAllGather"))) break;
            }
        }
        fn = reinterpret_cast<allgather_fn>(sym);
    }
    if (!fn) return fail(HR_E_HIP, "hr_allgather_tiles: no RCCL (ncclAllGather) in this process and librccl.so cannot be loaded");
    const int rc = fn(tile_dev, full_dev, (size_t)floats_per_rank, /* ncclFloat32 */ 7, nccl_comm, (hipStream_t)stream);
    if (rc != 0) return fail(HR_E_HIP, "ncclAllGather failed with ncclResult_t %d", rc);
    return HR_OK;
}

int hr_generate_rays(const hr_camera* cam, int32_t ray_dim, int64_t first_pixel, int64_t n_pixels, float* rays_dev, void* stream)
{
    if (!cam || (n_pixels > 0 && !rays_dev)) return fail(HR_E_INVALID, "null argument");
    if (ray_dim != 6 && ray_dim != 8) return fail(HR_E_INVALID, "ray_dim must be 6 or 8");
    if (cam->width < 1 || cam->height < 1 || cam->fx == 0.0f || cam->fy == 0.0f) return fail(HR_E_INVALID, "bad camera");
    if (first_pixel < 0 || n_pixels < 0 || first_pixel + n_pixels > (int64_t)cam->width * cam->height)
        return fail(HR_E_INVALID, "pixel range outside the image");
    hr_launch_generate_rays(*cam, ray_dim, first_pixel, n_pixels, rays_dev, (hipStream_t)stream);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int hr_upsample_plane(const float* src_dev, int32_t channels, int32_t h, int32_t w, float* dst_dev, int32_t h2, int32_t w2, void* stream)
{
    if (channels < 0 || h < 1 || w < 1 || h2 < 1 || w2 < 1) return fail(HR_E_INVALID, "bad plane shape");
    if (channels > 0 && (!src_dev || !dst_dev)) return fail(HR_E_INVALID, "null plane");
    hr_launch_upsample_plane(src_dev, channels, h, w, dst_dev, h2, w2, (hipStream_t)stream);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int hr_dense_alpha(hr_model* m, const int32_t n[3], float length, int32_t num_frames, const float* prev_volume_dev, const int32_t prev_n[3],
                   const float prev_aabb[6], float* alpha_dev, void* stream)
{
    if (!m || !n || !alpha_dev) return fail(HR_E_INVALID, "null argument");
    if (m->is_coarse) return fail(HR_E_INVALID, "the coarse level of a cascade has no grids");
    if (!m->finalized) return fail(HR_E_STATE, "hr_model_finalize has not been called (or tensors changed since)");
    if (m->cfg.grid_dtype != HR_GRID_FP32) return fail(HR_E_INVALID, "hr_dense_alpha reads float32 grids");
    if (n[0] < 1 || n[1] < 1 || n[2] < 1) return fail(HR_E_INVALID, "bad lattice size");
    if (m->cfg.video && num_frames < 1) return fail(HR_E_INVALID, "keyframe nets need num_frames");
    if (prev_volume_dev && (!prev_n || !prev_aabb || prev_n[0] < 1 || prev_n[1] < 1 || prev_n[2] < 1))
        return fail(HR_E_INVALID, "previous mask without its size / box");
    if (!m->ucfg_dev) {
        HR_HIP(hipMalloc((void**)&m->ucfg_dev, sizeof(hr_config)));
        HR_HIP(hipMemcpy(m->ucfg_dev, &m->cfg, sizeof(hr_config), hipMemcpyHostToDevice));
    }
    HrMaskArgs a = HrMaskArgs();
    a.cfg_dev = m->ucfg_dev;
    for (int j = 0; j < 3; ++j) { a.planes[j] = m->planes[j]; a.n[j] = n[j]; a.pn[j] = prev_volume_dev ? prev_n[j] : 0; }
    for (int j = 0; j < 6; ++j) a.prev_aabb[j] = prev_volume_dev ? prev_aabb[j] : 0.0f;
    a.length = length;
    a.num_frames = num_frames;
    a.prev_volume = prev_volume_dev;
    a.alpha = alpha_dev;
    hr_launch_dense_alpha(a, (hipStream_t)stream);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int hr_pack_display(const float* rgb_dev, int32_t h, int32_t w, int32_t transpose, int32_t flip, int32_t rgba8, void* out_dev, void* stream)
{
    if (h < 1 || w < 1) return fail(HR_E_INVALID, "bad image shape");
    if (!rgb_dev || !out_dev) return fail(HR_E_INVALID, "null argument");
    hr_launch_pack_display(rgb_dev, h, w, transpose ? 1 : 0, flip ? 1 : 0, rgba8 ? 1 : 0, out_dev, (hipStream_t)stream);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

size_t hr_linear_workspace(int64_t rows, int32_t in, int32_t out)
{
    if (rows < 1 || in < 1 || out < 1) return 0;
    return hr_linear_workspace_bytes(rows, in, out);
}

int hr_linear_forward(const float* x_dev, int64_t ldx, int64_t rows, int32_t in, const float* w_dev, const float* b_dev, int32_t out,
                      float leaky_slope, float* y_dev, int64_t ldy, void* stream)
{
    if (rows < 0 || in < 1 || out < 1 || ldx < in || ldy < out) return fail(HR_E_INVALID, "bad Linear shape");
    if (rows > 0 && (!x_dev || !w_dev || !y_dev)) return fail(HR_E_INVALID, "null argument");
    if (rows > 0x7fffffff) return fail(HR_E_INVALID, "more than 2^31 rows");
    hr_launch_linear_forward(x_dev, ldx, rows, in, w_dev, b_dev, out, leaky_slope, y_dev, ldy, (hipStream_t)stream);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int hr_linear_backward(const float* x_dev, int64_t ldx, const float* w_dev, const float* y_dev, int64_t ldy, const float* dy_dev, int64_t ld_dy,
                       int64_t rows, int32_t in, int32_t out, float leaky_slope, float* dx_dev, int64_t ld_dx, float* dw_dev, float* db_dev,
                       float* workspace_dev, void* stream)
{
    if (rows < 0 || in < 1 || out < 1 || ldx < in || ld_dy < out || (dx_dev && ld_dx < in) || (y_dev && ldy < out))
        return fail(HR_E_INVALID, "bad Linear shape");
    if (rows > 0 && (!x_dev || !w_dev || !dy_dev || !dw_dev || !db_dev || !workspace_dev)) return fail(HR_E_INVALID, "null argument");
    if (leaky_slope >= 0.0f && !y_dev) return fail(HR_E_INVALID, "an activated layer needs its output for the LeakyReLU mask");
    if (rows > 0x7fffffff) return fail(HR_E_INVALID, "more than 2^31 rows");
    if (rows == 0) {
        HR_HIP(hipMemsetAsync(dw_dev, 0, sizeof(float) * (size_t)out * in, (hipStream_t)stream));
        HR_HIP(hipMemsetAsync(db_dev, 0, sizeof(float) * (size_t)out, (hipStream_t)stream));
        return HR_OK;
    }
    hr_launch_linear_backward(x_dev, ldx, w_dev, y_dev, ldy, dy_dev, ld_dy, rows, in, out, leaky_slope, dx_dev, ld_dx, dw_dev, db_dev, workspace_dev,
                              (hipStream_t)stream);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int hr_plane_reg_forward(const float* plane_dev, int32_t channels, int32_t h, int32_t w, float* sums_dev, void* stream)
{
    if (channels < 0 || h < 1 || w < 1) return fail(HR_E_INVALID, "bad plane shape");
    if (!sums_dev || (channels > 0 && !plane_dev)) return fail(HR_E_INVALID, "null argument");
    hr_launch_plane_reg_forward(plane_dev, channels, h, w, sums_dev, (hipStream_t)stream);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int hr_plane_reg_backward(const float* plane_dev, int32_t channels, int32_t h, int32_t w, const float* coef_dev, float* grad_dev, void* stream)
{
    if (channels < 0 || h < 1 || w < 1) return fail(HR_E_INVALID, "bad plane shape");
    if (channels > 0 && (!plane_dev || !coef_dev || !grad_dev)) return fail(HR_E_INVALID, "null argument");
    hr_launch_plane_reg_backward(plane_dev, channels, h, w, coef_dev, grad_dev, (hipStream_t)stream);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int hr_adam_step(float* const* param_dev, const float* const* grad_dev, float* const* exp_avg_dev, float* const* exp_avg_sq_dev, const int64_t* n,
                 const double* hp, int32_t n_tensors, void* stream)
{
    if (n_tensors < 0 || (n_tensors > 0 && (!param_dev || !grad_dev || !exp_avg_dev || !exp_avg_sq_dev || !n || !hp))) return fail(HR_E_INVALID, "null argument");
    if (n_tensors == 0) return HR_OK;
    HrAdamBatch b;
    b.count = 0;
    b.first_block[0] = 0;
    auto flush = [&]() {
        hr_launch_adam(b, (hipStream_t)stream);
        b.count = 0;
        b.first_block[0] = 0;
    };
    for (int i = 0; i < n_tensors; ++i) {
        if (n[i] < 0) return fail(HR_E_INVALID, "hr_adam_step: tensor %d has a negative size", i);
        if (n[i] == 0) continue;
        if (!param_dev[i] || !grad_dev[i] || !exp_avg_dev[i] || !exp_avg_sq_dev[i]) return fail(HR_E_INVALID, "hr_adam_step: tensor %d has a null buffer", i);
        const double lr = hp[6 * i], b1 = hp[6 * i + 1], b2 = hp[6 * i + 2], eps = hp[6 * i + 3], wd = hp[6 * i + 4], step = hp[6 * i + 5];
        if (!(step >= 1.0) || !(b1 >= 0.0 && b1 < 1.0) || !(b2 >= 0.0 && b2 < 1.0)) return fail(HR_E_INVALID, "hr_adam_step: tensor %d: step >= 1 and betas in [0, 1) required", i);
        const int64_t blocks = (n[i] + 4095) / 4096;
        if (blocks > 0x3fffffff) return fail(HR_E_INVALID, "hr_adam_step: tensor %d is too large", i);
        if (b.count == HR_ADAM_MAX_TENSORS || (int64_t)b.first_block[b.count] + blocks > 0x7fffffff) flush();
        const int k = b.count++;
        b.p[k] = param_dev[i]; b.g[k] = grad_dev[i]; b.m[k] = exp_avg_dev[i]; b.v[k] = exp_avg_sq_dev[i]; b.n[k] = n[i];
        // bias corrections in double on the host (torch: python floats)
        const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
        b.step_size[k] = (float)(lr / bc1);
        b.inv_sqrt_bc2[k] = (float)(1.0 / sqrt(bc2));
        b.omb1[k] = (float)(1.0 - b1); b.beta2[k] = (float)b2; b.omb2[k] = (float)(1.0 - b2); b.eps[k] = (float)eps; b.weight_decay[k] = (float)wd;
        b.first_block[k + 1] = b.first_block[k] + (int)blocks;
    }
    if (b.count > 0) flush();
    HR_HIP(hipGetLastError());
    return HR_OK;
}

// ---------------------------------------------------------------- training path
static int check_train(hr_model* m, const float* rays, int64_t n)
{
    if (!m) return fail(HR_E_INVALID, "null model");
    if (m->is_coarse) return fail(HR_E_INVALID, "training path: pass the cascade's handle, not its coarse level");
    if (!m->finalized) return fail(HR_E_STATE, "hr_model_finalize has not been called (or tensors changed since)");
    if (const char* why = hr_train_unsupported(m->cfg)) return fail(HR_E_INVALID, "training path: %s not differentiated", why);
    if (m->ca_total > HR_TRAIN_MAX_CA) return fail(HR_E_INVALID, "training path: more than %d appearance components", HR_TRAIN_MAX_CA);
    if (n < 0 || (n > 0 && !rays)) return fail(HR_E_INVALID, "bad ray buffer");
    for (hr_model* lvl : {m, m->coarse}) {
        if (!lvl || lvl->ucfg_dev) continue;
        HR_HIP(hipMalloc((void**)&lvl->ucfg_dev, sizeof(hr_config)));
        HR_HIP(hipMemcpy(lvl->ucfg_dev, &lvl->cfg, sizeof(hr_config), hipMemcpyHostToDevice));
    }
    return HR_OK;
}

// per-sample workspace of the backward's phases (30 words per sample); grows on the first step and if the batch grows
static int ensure_tape(hr_model* m, int64_t ns, hipStream_t st)
{
    if (ns <= m->tape_samples) return HR_OK;
    HR_HIP(hipStreamSynchronize(st));
    free_dev(m->tape);
    m->tape_samples = 0;
    HR_HIP(hipMalloc((void**)&m->tape, sizeof(float) * 30 * (size_t)ns));       // HrTrainTape: 8 planes + 18 of taps + 3 of dL/d point + the grouped ray order (n_rays <= ns ints)
    m->tape_samples = ns;
    return HR_OK;
}

// the four reference-layout tensors of plane pair j: {density a, app a, density b, app b} with their channel counts
struct TrainPlaneIO {
    float* p[4];
    int ch[4];
};
static TrainPlaneIO train_plane_io(const hr_model* m, const hr_train_tensors* t, int j)
{
    const hr_config& c = m->cfg;
    int nd = c.n_den[j], na = c.n_app[j];
    if (c.video && nd == 0) na = 0;
    return TrainPlaneIO{{t->density_a[j], t->app_a[j], t->density_b[j], t->app_b[j]}, {nd, na, nd, na}};
}

int hr_train_features(hr_model* m, const float* rays_dev, int64_t n_rays, float* feats_dev, void* stream)
{
    int rc = check_train(m, rays_dev, n_rays);
    if (rc != HR_OK) return rc;
    if (n_rays > 0 && !feats_dev) return fail(HR_E_INVALID, "null feature buffer");
    // a cascade's ray MLP belongs to its coarse level
    hr_launch_features(m->coarse ? m->coarse->ucfg_dev : m->ucfg_dev, rays_dev, n_rays, feats_dev, (hipStream_t)stream);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int hr_mlp_train_forward(hr_model* m, const float* const* weights_dev, const float* const* biases_dev, const float* rays_dev, int64_t n_rays,
                         float* const* acts_dev, const int64_t* act_ld, const int32_t* act_off, float* head_dev, void* stream)
{
    int rc = check_train(m, rays_dev, n_rays);
    if (rc != HR_OK) return rc;
    if (m->coarse || m->is_coarse) return fail(HR_E_INVALID, "hr_mlp_train_forward: point_prediction cascades run their MLPs layer by layer (hr_linear_forward)");
    const hr_config& c = m->cfg;
    const int L = c.mlp_layers;
    if (L < 2 || c.mlp_hidden != 256) return fail(HR_E_INVALID, "hr_mlp_train_forward needs hidden width 256 and at least two layers");
    if (!weights_dev || !biases_dev || !acts_dev || !act_ld || !act_off || (n_rays > 0 && !head_dev)) return fail(HR_E_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    // ---- the current parameter values -> bf16 hi / lo tiles, on the device (what pack_mlp does on the host at finalize)
    const int P_user = c.preds_per_z, P_live = m->p_live;
    const int k0p = (c.mlp_in + 15) & ~15;
    const int n_out = samples_per_row(c) * P_live;
    for (int l = 0; l < L; ++l) {
        if (!weights_dev[l] || !biases_dev[l]) return fail(HR_E_INVALID, "hr_mlp_train_forward: layer %d has no weights", l);
        HrPackDesc d = {};
        d.w = weights_dev[l]; d.b = biases_dev[l];
        d.last = (l == L - 1); d.first = (l == 0); d.skip = (c.mlp_skip_mask >> l) & 1;
        d.N_user = layer_out(c, l); d.Kt = layer_in(c, l);
        d.N = d.last ? n_out : d.N_user;
        d.nt = (d.N + 31) / 32;
        d.Kp = d.first ? k0p : (d.skip ? k0p + 256 : 256);
        d.mlp_in = c.mlp_in; d.k0p = k0p; d.P_user = P_user; d.P_live = P_live;
        for (int i = 0, j = 0; i < P_user && i < 64; ++i)
            if (m->col_map.col[i] >= 0) d.live_cols[j++] = i;
        if (!m->wsplit_t[l] || m->n_tiles_t[l] != d.nt) {
            if (m->wsplit_t[l]) (void)hipFree(m->wsplit_t[l]);
            free_dev(m->bias_t[l]);
            m->wsplit_t[l] = nullptr;
            HR_HIP(hipMalloc(&m->wsplit_t[l], sizeof(uint16_t) * (size_t)(d.Kp / 16) * d.nt * 2 * 64 * 8));
            HR_HIP(hipMalloc((void**)&m->bias_t[l], sizeof(float) * (size_t)d.nt * 32));
            m->n_tiles_t[l] = d.nt;
        }
        d.wsplit = m->wsplit_t[l]; d.bias = m->bias_t[l];
        hr_launch_pack_split_bf16(d, st);
    }
    HrMlpTaps taps = {};
    for (int l = 0; l + 1 < L; ++l) { taps.act[l] = acts_dev[l]; taps.ld[l] = act_ld[l]; taps.off[l] = act_off[l]; }
    const int nq = (n_out + 3) / 4;
    for (int64_t r0 = 0; r0 < n_rays; r0 += m->chunk) {
        const int64_t n = (n_rays - r0 < m->chunk) ? (n_rays - r0) : m->chunk;
        HrMlpArgs a = {};
        a.rays = rays_dev + r0 * c.ray_dim;
        a.n_rays = n;
        a.head = m->head;
        for (int l = 0; l < L; ++l) { a.wsplit[l] = m->wsplit_t[l]; a.bias[l] = m->bias_t[l]; a.winv[l] = 1.0f; a.n_tiles[l] = m->n_tiles_t[l]; }
        a.n_out = n_out; a.nq = nq; a.k0p = k0p;
        a.trace = nullptr; a.flags = nullptr;
        HrMlpTaps tc = taps;
        for (int l = 0; l + 1 < L; ++l)
            if (tc.act[l]) tc.act[l] += r0 * tc.ld[l];
        hr_launch_mlp_train_bf16x3(m->kcfg, a, tc, st);
        hr_launch_head_export(m->head, head_dev + r0 * (int64_t)c.z_channels * P_user, n, c.z_channels, P_user, P_live, nq, 1, m->col_map, st);
    }
    HR_HIP(hipGetLastError());
    return HR_OK;
}

static void fill_train_args(const hr_model* m, HrTrainArgs& a, const float* rays, const float* head, int64_t n, int white_bg)
{
    a.f_dist = a.f_points = a.f_weights = nullptr;
    a.fx = nullptr;
    a = HrTrainArgs();
    a.cfg_dev = m->ucfg_dev;
    a.rays = rays;
    a.head = head;
    a.n_rays = n;
    for (int j = 0; j < 3; ++j) { a.planes[j] = m->planes[j]; a.g_a[j] = m->grad_a[j]; a.g_b[j] = m->grad_b[j]; }
    a.basis = m->basis;
    a.n_basis_cols = m->n_basis_cols;
    a.ca_total = m->ca_total;
    a.white_bg = white_bg ? 1 : 0;
    a.color_table = nullptr;
    if (m->cfg.color_table_views > 0) {
        auto it = m->raw.find("color_embedding");
        if (it != m->raw.end()) a.color_table = it->second.p;
    }
}

static int train_forward(hr_model* m, const hr_train_tensors* params, const float* rays_dev, const float* head_dev, int64_t n_rays,
                         int32_t white_bg, float* rgb_dev, const hr_fields* fields, void* stream);

int hr_train_forward(hr_model* m, const hr_train_tensors* params, const float* rays_dev, const float* head_dev, int64_t n_rays,
                     int32_t white_bg, float* rgb_dev, void* stream)
{
    return train_forward(m, params, rays_dev, head_dev, n_rays, white_bg, rgb_dev, nullptr, stream);
}

int hr_train_forward_fields(hr_model* m, const hr_train_tensors* params, const float* rays_dev, const float* head_dev, int64_t n_rays,
                            int32_t white_bg, float* rgb_dev, const hr_fields* fields, void* stream)
{
    if (fields && (fields->sigma_dev || fields->head_dev)) return fail(HR_E_INVALID, "hr_train_forward_fields serves distances, points and weights");
    if (fields && m && m->cfg.z_channels > 64) return fail(HR_E_INVALID, "hr_train_forward_fields: rays of more than 64 samples take the one-thread-per-ray walk, which keeps no fields");
    return train_forward(m, params, rays_dev, head_dev, n_rays, white_bg, rgb_dev, fields, stream);
}

static int train_forward(hr_model* m, const hr_train_tensors* params, const float* rays_dev, const float* head_dev, int64_t n_rays,
                         int32_t white_bg, float* rgb_dev, const hr_fields* fields, void* stream)
{
    int rc = check_train(m, rays_dev, n_rays);
    if (rc != HR_OK) return rc;
    if (n_rays > 0 && (!head_dev || !rgb_dev)) return fail(HR_E_INVALID, "null head / rgb buffer");
    hipStream_t st = (hipStream_t)stream;
    if (params) {                     // this step's parameter values -> the kernels' texel layout (no allocation, no sync)
        HrLayoutBatch batch = {};                  // all twelve tensors in one launch
        for (int j = 0; j < 3; ++j) {
            const HrGridPlane& g = m->planes[j];
            if (g.tex == 0) continue;
            const TrainPlaneIO io = train_plane_io(m, params, j);
            for (int t = 0; t < 4; ++t) {
                if (io.ch[t] == 0) continue;
                if (!io.p[t]) return fail(HR_E_INVALID, "hr_train_forward: params tensor of plane pair %d is NULL", j);
                const bool is_a = t < 2;
                batch.job[batch.n++] = HrLayoutJob{io.p[t], is_a ? m->grid_a[j] : m->grid_b[j], io.ch[t], is_a ? g.ah : g.bh, is_a ? g.aw : g.bw, g.tex,
                                                   (t & 1) ? 4 * g.cd4 : 0};
            }
        }
        hr_launch_layout_batch(batch, true, st);
        const size_t bytes = m->raw["basis_mat.weight"].bytes;
        if (bytes > 0) {
            if (!params->basis) return fail(HR_E_INVALID, "hr_train_forward: params->basis is NULL");
            HR_HIP(hipMemcpyAsync(m->basis, params->basis, bytes, hipMemcpyDeviceToDevice, st));
            // the render kernels read the column-major copy: keep it in step, so that hr_render after a training step sees the
            // same basis_mat as the planes refreshed above
            hr_launch_basis_transpose(m->basis, m->basis_t, m->cfg.app_dim, m->n_basis_cols, m->basis_ld, st);
        }
        if (m->cfg.color_table_views > 0) {       // read in place from the uploaded copy: refresh it
            if (!params->color_table) return fail(HR_E_INVALID, "hr_train_forward: params->color_table is NULL");
            DevBuf& b = m->raw["color_embedding"];
            HR_HIP(hipMemcpyAsync(b.p, params->color_table, b.bytes, hipMemcpyDeviceToDevice, st));
        }
    }
    HrTrainArgs a;
    fill_train_args(m, a, rays_dev, head_dev, n_rays, white_bg);
    a.rgb = rgb_dev;
    if (fields) { a.f_dist = fields->distances_dev; a.f_points = fields->points_dev; a.f_weights = fields->weights_dev; }
    hr_launch_train(m->cfg, a, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int hr_train_backward(hr_model* m, const float* rays_dev, const float* head_dev, const float* d_rgb_dev, int64_t n_rays,
                      int32_t white_bg, float* d_head_dev, const hr_train_tensors* grads, void* stream)
{
    int rc = check_train(m, rays_dev, n_rays);
    if (rc != HR_OK) return rc;
    if (!grads) return fail(HR_E_INVALID, "null grads");
    if (n_rays > 0 && (!head_dev || !d_rgb_dev || !d_head_dev)) return fail(HR_E_INVALID, "null head / d_rgb / d_head buffer");
    hipStream_t st = (hipStream_t)stream;
    if (!m->grad_pool) {              // packed accumulators: one allocation, made on the first step
        size_t off_a[3] = {}, off_b[3] = {}, total = 0;
        for (int j = 0; j < 3; ++j) {
            const HrGridPlane& g = m->planes[j];
            if (g.tex == 0) continue;
            off_a[j] = total; total += (sizeof(float) * (size_t)g.aw * g.ah * g.tex + 255) & ~(size_t)255;
            off_b[j] = total; total += (sizeof(float) * (size_t)g.bw * g.bh * g.tex + 255) & ~(size_t)255;
        }
        if (total > 0) {
            HR_HIP(hipMalloc((void**)&m->grad_pool, total));
            m->grad_pool_bytes = total;
            for (int j = 0; j < 3; ++j) {
                if (m->planes[j].tex == 0) continue;
                m->grad_a[j] = reinterpret_cast<float*>(reinterpret_cast<char*>(m->grad_pool) + off_a[j]);
                m->grad_b[j] = reinterpret_cast<float*>(reinterpret_cast<char*>(m->grad_pool) + off_b[j]);
            }
        }
    }
    // cleared per step on the stream, in one go (the deterministic mode overwrites them from its fixed-point sums instead)
    if (m->grad_pool && !m->opt_train_det) HR_HIP(hipMemsetAsync(m->grad_pool, 0, m->grad_pool_bytes, st));
    const size_t basis_bytes = m->raw["basis_mat.weight"].bytes;
    // basis_mat's gradient needs no re-layout: accumulate in the caller's buffer (or a scratch nobody reads)
    float* d_basis = grads->basis;
    if (!d_basis) return fail(HR_E_INVALID, "hr_train_backward: grads->basis is NULL");
    if (basis_bytes > 0) HR_HIP(hipMemsetAsync(d_basis, 0, basis_bytes, st));
    const int64_t ns = n_rays * m->cfg.z_channels;
    rc = ensure_tape(m, ns, st);
    if (rc != HR_OK) return rc;
    HrTrainArgs a;
    fill_train_args(m, a, rays_dev, head_dev, n_rays, white_bg);
    a.tape.ds = m->tape;
    a.tape.src = reinterpret_cast<int*>(m->tape + ns);
    a.tape.dfeat = m->tape + 2 * ns;
    a.tape.dpre = m->tape + 3 * ns;      // 3 planes
    a.tape.ddc = m->tape + 6 * ns;
    a.tape.dts = m->tape + 7 * ns;
    a.tape.taps = m->tape + 8 * ns;
    a.tape.dp = m->tape + 26 * ns;
    a.tape.perm = reinterpret_cast<int*>(m->tape + 29 * ns);
    a.d_rgb = d_rgb_dev;
    a.d_head = d_head_dev;
    a.d_basis = d_basis;
    if (m->cfg.color_table_views > 0) {
        if (!grads->color_table) return fail(HR_E_INVALID, "hr_train_backward: grads->color_table is NULL");
        HR_HIP(hipMemsetAsync(grads->color_table, 0, sizeof(float) * 12 * (size_t)m->cfg.color_table_views, st));
        a.d_color_table = grads->color_table;
    }
    if (m->opt_train_det) {
        // deterministic mode: every accumulator of the step is a 64-bit fixed-point word of ONE scratch buffer (integer atomics: the
        // totals do not depend on the order of the adds); converted to the float buffers the rest of the step reads
        size_t need = 0, off_a[3] = {}, off_b[3] = {}, n_a[3] = {}, n_b[3] = {};
        for (int j = 0; j < 3; ++j) {
            const HrGridPlane& g = m->planes[j];
            if (g.tex == 0) continue;
            n_a[j] = (size_t)g.aw * g.ah * g.tex; n_b[j] = (size_t)g.bw * g.bh * g.tex;
            off_a[j] = need; need += n_a[j];
            off_b[j] = need; need += n_b[j];
        }
        const size_t n_basis = basis_bytes / sizeof(float), off_basis = need;
        need += n_basis;
        const size_t n_ct = m->cfg.color_table_views > 0 ? 12 * (size_t)m->cfg.color_table_views : 0, off_ct = need;
        need += n_ct;
        if (need > m->grad_fx_elems) {
            HR_HIP(hipStreamSynchronize(st));
            if (m->grad_fx) (void)hipFree(m->grad_fx);
            m->grad_fx = nullptr; m->grad_fx_elems = 0;
            HR_HIP(hipMalloc((void**)&m->grad_fx, sizeof(long long) * need));
            m->grad_fx_elems = need;
        }
        HR_HIP(hipMemsetAsync(m->grad_fx, 0, sizeof(long long) * need, st));
        if (!m->fx_unit) HR_HIP(hipMalloc((void**)&m->fx_unit, sizeof(HrFxUnit)));
        HrTrainArgs ad = a;
        ad.fx = m->fx_unit;
        for (int j = 0; j < 3; ++j) {
            ad.g_a[j] = n_a[j] ? reinterpret_cast<float*>(m->grad_fx + off_a[j]) : nullptr;
            ad.g_b[j] = n_b[j] ? reinterpret_cast<float*>(m->grad_fx + off_b[j]) : nullptr;
        }
        ad.d_basis = reinterpret_cast<float*>(m->grad_fx + off_basis);
        ad.d_color_table = n_ct ? reinterpret_cast<float*>(m->grad_fx + off_ct) : nullptr;
        hr_launch_train_det(m->cfg, &ad, sizeof(ad), st);
        const float* fx_inv = &m->fx_unit->inv;
        const unsigned* fx_bad = &m->fx_unit->bad;
        for (int j = 0; j < 3; ++j) {
            if (n_a[j]) hr_launch_fixed_to_float(m->grad_fx + off_a[j], m->grad_a[j], (int64_t)n_a[j], fx_inv, fx_bad, st);
            if (n_b[j]) hr_launch_fixed_to_float(m->grad_fx + off_b[j], m->grad_b[j], (int64_t)n_b[j], fx_inv, fx_bad, st);
        }
        hr_launch_fixed_to_float(m->grad_fx + off_basis, d_basis, (int64_t)n_basis, fx_inv, fx_bad, st);
        if (n_ct) hr_launch_fixed_to_float(m->grad_fx + off_ct, grads->color_table, (int64_t)n_ct, fx_inv, fx_bad, st);
    } else {
        hr_launch_train(m->cfg, a, st);
    }
    HrLayoutBatch batch = {};                      // packed texel gradients -> the reference's (C, H, W) tensors, one launch
    for (int j = 0; j < 3; ++j) {
        const HrGridPlane& g = m->planes[j];
        if (g.tex == 0) continue;
        const TrainPlaneIO io = train_plane_io(m, grads, j);
        for (int t = 0; t < 4; ++t) {
            if (io.ch[t] == 0 || !io.p[t]) continue;
            const bool is_a = t < 2;
            batch.job[batch.n++] = HrLayoutJob{is_a ? m->grad_a[j] : m->grad_b[j], io.p[t], io.ch[t], is_a ? g.ah : g.bh, is_a ? g.aw : g.bw, g.tex,
                                               (t & 1) ? 4 * g.cd4 : 0};
        }
    }
    hr_launch_layout_batch(batch, false, st);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

static int fill_rows_args(hr_model* m, HrRowsArgs& a, const float* rays, const float* head, int64_t n)
{
    if (!m->coarse) return fail(HR_E_INVALID, "hr_train_rows_*: the model is not a point_prediction cascade");
    a = HrRowsArgs();
    a.cfg_dev = m->coarse->ucfg_dev;
    a.rays = rays;
    a.head = head;
    a.n_rays = n;
    a.row_dim = m->cfg.casc_row_dim;
    a.n_inputs = m->cfg.casc_n_inputs;
    for (int i = 0; i < 4; ++i) { a.kind[i] = m->cfg.casc_input_kind[i]; a.len[i] = m->cfg.casc_input_dim[i]; }
    return HR_OK;
}

int hr_train_rows_forward(hr_model* m, const float* rays_dev, const float* head_dev, int64_t n_rays, float* rows_dev, void* stream)
{
    int rc = check_train(m, rays_dev, n_rays);
    if (rc != HR_OK) return rc;
    if (n_rays > 0 && (!head_dev || !rows_dev)) return fail(HR_E_INVALID, "null head / rows buffer");
    HrRowsArgs a;
    rc = fill_rows_args(m, a, rays_dev, head_dev, n_rays);
    if (rc != HR_OK) return rc;
    a.rows = rows_dev;
    hr_launch_rows(m->coarse->cfg, a, (hipStream_t)stream);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int hr_train_rows_backward(hr_model* m, const float* rays_dev, const float* head_dev, const float* d_rows_dev, int64_t n_rays,
                           float* rows_scratch_dev, float* d_head_dev, void* stream)
{
    int rc = check_train(m, rays_dev, n_rays);
    if (rc != HR_OK) return rc;
    if (n_rays > 0 && (!head_dev || !d_rows_dev || !rows_scratch_dev || !d_head_dev)) return fail(HR_E_INVALID, "null buffer");
    HrRowsArgs a;
    rc = fill_rows_args(m, a, rays_dev, head_dev, n_rays);
    if (rc != HR_OK) return rc;
    const int64_t ns = n_rays * m->coarse->cfg.z_channels;
    rc = ensure_tape(m, ns, (hipStream_t)stream);
    if (rc != HR_OK) return rc;
    a.rows = rows_scratch_dev;
    a.d_rows = d_rows_dev;
    a.d_head = d_head_dev;
    a.tape.ds = m->tape;
    a.tape.src = reinterpret_cast<int*>(m->tape + ns);
    a.tape.dts = m->tape + 2 * ns;
    hr_launch_rows(m->coarse->cfg, a, (hipStream_t)stream);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int hr_stage_mlp(hr_model* m, const float* rays_dev, int64_t n_rays, void* stream)
{
    int rc = check_render(m, rays_dev, n_rays, rays_dev);
    if (rc != HR_OK) return rc;
    if (n_rays > m->chunk) return fail(HR_E_INVALID, "n_rays exceeds the reserved chunk (%lld)", (long long)m->chunk);
    launch_front(m, rays_dev, n_rays, (hipStream_t)stream);   // cascades: everything up to the fine head
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int hr_stage_samples(hr_model* m, const float* rays_dev, int64_t n_rays, float* rgb_dev, void* stream)
{
    int rc = check_render(m, rays_dev, n_rays, rgb_dev);
    if (rc != HR_OK) return rc;
    if (n_rays > m->chunk) return fail(HR_E_INVALID, "n_rays exceeds the reserved chunk (%lld)", (long long)m->chunk);
    HrSampleArgs sa;
    fill_sample_args(m, sa, rays_dev, n_rays, rgb_dev);
    hr_launch_samples(m->kcfg, sa, (hipStream_t)stream);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int hr_debug_trace_mlp(hr_model* m, const float* rays_dev, int64_t n_rays, unsigned long long* trace_dev, void* stream)
{
    int rc = check_render(m, rays_dev, n_rays, rays_dev);
    if (rc != HR_OK) return rc;
    if (n_rays > m->chunk) return fail(HR_E_INVALID, "n_rays exceeds the reserved chunk (%lld)", (long long)m->chunk);
    if (m->coarse) return fail(HR_E_INVALID, "hr_debug_trace_mlp does not support cascades");
    HrMlpArgs ma;
    fill_mlp_args(m, ma, rays_dev, n_rays);
    ma.trace = trace_dev;
    launch_mlp(m, m->kcfg, ma, (hipStream_t)stream);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

int64_t hr_model_device_bytes(const hr_model* m)
{
    if (!m) return 0;
    int64_t raw = 0;
    for (auto& kv : m->raw) raw += (int64_t)kv.second.bytes;
    return raw + m->packed_bytes + (int64_t)sizeof(float) * m->chunk * m->cfg.z_channels * m->p_live +
           (m->rows ? (int64_t)sizeof(float) * m->chunk * m->cfg.casc_in_z * m->cfg.casc_row_dim : 0) + hr_model_device_bytes(m->coarse);
}

void hr_model_destroy(hr_model* m)
{
    if (!m) return;
    for (auto& kv : m->raw) free_dev(kv.second.p);
    for (int l = 0; l < HR_MAX_LAYERS; ++l) {
        free_dev(reinterpret_cast<float*&>(m->wpack[l]));
        free_dev(reinterpret_cast<float*&>(m->wsplit[l]));
        free_dev(m->bias[l]);
    }
    free_safe_pack(m);
    free_dev(m->calib_rays);
    free_dev(reinterpret_cast<float*&>(m->redo_list));
    free_dev(reinterpret_cast<float*&>(m->wide_list));
    free_dev(reinterpret_cast<float*&>(m->redo_count));
    for (int j = 0; j < 3; ++j) {
        free_dev(m->grid_a[j]);
        free_dev(m->grid_b[j]);
    }
    free_dev(m->basis);
    free_dev(m->basis_t);
    free_dev(reinterpret_cast<float*&>(m->slot_col));
    free_dev(reinterpret_cast<float*&>(m->flags));
    free_dev(m->head);
    free_dev(m->rows);
    free_dev(m->occ);
    free_dev(reinterpret_cast<float*&>(m->occ_cells));
    if (m->kcfg_dev) (void)hipFree(m->kcfg_dev);
    if (m->ucfg_dev) (void)hipFree(m->ucfg_dev);
    free_dev(m->grad_pool);
    for (int j = 0; j < 3; ++j) { m->grad_a[j] = m->grad_b[j] = nullptr; free_dev(m->frame_line[j]); }
    free_dev(m->tape);
    if (m->grad_fx) (void)hipFree(m->grad_fx);
    if (m->fx_unit) (void)hipFree(m->fx_unit);
    for (int l = 0; l < HR_MAX_LAYERS; ++l) {          // the training forward's per-step weight tiles (hr_mlp_train_forward)
        if (m->wsplit_t[l]) (void)hipFree(m->wsplit_t[l]);
        m->wsplit_t[l] = nullptr;
        free_dev(m->bias_t[l]);
    }
    hr_model_destroy(m->coarse);
    delete m;
}

}  // extern "C"
