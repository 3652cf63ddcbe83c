// Internal interface between the C-ABI layer (api.hip) and the kernels.
#ifndef HR_KERNELS_H
#define HR_KERNELS_H

// samples per ray the sample kernel handles (a 256-thread block per ray at most)
#define HR_KERNEL_MAX_Z 256

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hyperreel_hip.h"

// More than 64 KiB of dynamic LDS has to be opted into per kernel AND per device (hipFuncAttributeMaxDynamicSharedMemorySize).
// `cache`: one static per kernel instantiation; a process that renders on several GPUs (viewer + trainer, DataParallel-style
// hosts) sets the attribute on each device it launches on.  Returns false when the runtime refuses.
#include <atomic>
struct HrLdsOptIn {
    std::atomic<size_t> have[32];
};
inline bool hr_lds_opt_in(HrLdsOptIn& cache, const void* kernel, size_t lds)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<size_t>& have = cache.have[dev & 31];
    if (lds <= have.load(std::memory_order_acquire)) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    size_t cur = have.load(std::memory_order_relaxed);
    while (cur < lds && !have.compare_exchange_weak(cur, lds, std::memory_order_release)) {}
    return true;
}
// compute units of the CURRENT device (persistent launches size their grid by it)
inline int hr_current_device_cus()
{
    static std::atomic<int> n[32];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int v = n[dev & 31].load(std::memory_order_relaxed);
    if (v == 0) {
        hipDeviceProp_t prop;
        v = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        n[dev & 31].store(v, std::memory_order_relaxed);
    }
    return v;
}

// ---------------------------------------------------------------- MLP (mlp_kernel.hip)
// Weights of layer L are stored as MFMA B-operand tiles for v_mfma_f32_16x16x4_f32:
//   wpack[L][((kt * n_tiles[L]) + nt) * 64 + lane] = float4{ W[n][k0], W[n][k0+1], W[n][k0+2], W[n][k0+3] }
//   with n = 16*nt + (lane & 15), k0 = 16*kt + 4*(lane >> 4)
// (W[n][k] = torch weight (out=n, in=k) in the kernel's K order; zero outside the real matrix).
// K order: plain layers: hidden index.  Layer 0: input feature index, padded to k0p.
// Skip layers: [input features padded to k0p | hidden].
// Layout of the raw MLP head in the workspace ("HQ"): rays in blocks of 64, features in quads,
//   float index(ray, n) = (((ray / 64) * nq + n / 4) * 64 + ray % 64) * 4 + n % 4,  nq = ceil(n_out / 4)
// The MLP's 32x32 accumulator tile holds 4 consecutive features of one ray per lane, so a
// store instruction covers 32 rays x 16 B = 512 contiguous bytes; the sample kernel reads a
// block's rays as RPB x 16 B contiguous runs per quad.
__host__ __device__ inline size_t hr_head_index(int64_t ray, int n, int nq)
{
    return ((((size_t)(ray >> 6) * nq + (size_t)(n >> 2)) << 6) + (size_t)(ray & 63)) * 4 + (size_t)(n & 3);
}

struct HrMlpArgs {
    const float* rays;
    int64_t n_rays;
    float* head;                 // raw output of the last Linear, HQ layout (hr_head_index)
    const float4* wpack[HR_MAX_LAYERS];   // HR_MLP_FP32: fp32 tiles (16-column tiles)
    const void* wsplit[HR_MAX_LAYERS];    // HR_MLP_BF16X3: bf16 hi/lo tiles (32-feature tiles), see mlp_bf16x3_kernel.hip
    const float* bias[HR_MAX_LAYERS];
    float winv[HR_MAX_LAYERS];   // split kernels: the packed weights of layer L are W * 2^s (fp16 modes: keeps the low halves out of
                                 //   the subnormal range); the epilogue multiplies the accumulator by winv = 2^-s (exact).  1 for bf16
    int xexp[HR_MAX_LAYERS];     // f16 + fp8 split (mlp_f16f8_kernel.hip): the fp8 images of hidden Linear L's output are e4m3(x * 2^-xexp[L]) and
                                 //   e4m3((x - half(x)) * 2^(11 - xexp[L])); from the activation-range calibration, 0 in the other modes
    int n_tiles[HR_MAX_LAYERS];  // output tiles of layer L: 16 columns (fp32) or 32 features (bf16x3)
    int n_out;                   // Z * P
    int nq;                      // ceil(n_out / 4)
    int k0p;                     // mlp_in padded to a multiple of 16
    unsigned long long* trace;   // bf16x3 kernel: optional phase timeline, 64 stamps per wave (hr_debug_trace_mlp)
    unsigned* flags;             // sticky status word of the model: bit 0 = an fp16-split kernel saw an input feature or hidden activation
                                 //   at or beyond the IEEE-half range (HR_OPT_MLP_OVERFLOW); bit 1 = the f16 + fp8 split saturated an fp8 image
                                 //   of a hidden activation (HR_OPT_MLP_F8_SATURATED); bit 2 = the redo list overflowed
    // verified fast path (split kernels only).  First pass: redo_list != NULL -- a tile in which a range bit was raised (half
    // overflow, fp8 saturation) appends its rays as ray0 + index (ray0: where `rays` starts in the caller's buffer).  Second pass: `rays` is the
    // caller's whole buffer and ray_index != NULL -- head row i is the caller's ray ray_index[i], *n_rays_dev of them (n_rays: capacity).
    int64_t ray0;
    const int* ray_index;
    const unsigned* n_rays_dev;
    int64_t list_off;            // second pass in slices of the workspace's capacity: ray_index starts at entry list_off of the list, *n_rays_dev - list_off remain
    unsigned* n_rays_copy;       // second pass: workgroup 0 copies *n_rays_dev here (what the second pass's sample kernel reads, so that IT can clear the
                                 //   counter for the next call -- a memset node between the calls does not survive hipGraph replay: the second
                                 //   replay found 0x40404040 there)
    int* redo_list;
    unsigned* redo_count;
    int redo_cap;
};

// ---------------------------------------------------------------- sample stage (sample_kernel.hip)
#include "hr_grid.h"

struct HrSampleArgs {
    const hr_config* cfg_dev;   // device copy of the configuration: the sample kernel indexes it per lane (samples[k] ...), and a
                                // 2 KB by-value kernel argument indexed dynamically can end up copied to scratch by the compiler
    const float* rays;
    const float* head;      // HQ layout (hr_head_index)
    int nq;                 // ceil(Z*P / 4)
    int64_t n_rays;
    float* rgb;
    hr_fields fields;       // optional diagnostics (NULL pointers when unused)
    HrGridPlane planes[3];
    const float* basis;     // (app_dim, n_basis_cols) row-major, torch layout
    const float* basis_t;   // the same matrix column-major: basis_t[col * basis_ld + row] -- the 27 (3) values a decode-matrix slot folds are contiguous
    const int* slot_col;    // padded appearance slot -> column of basis_mat, or -1 (ca_total entries)
    int basis_ld;           // app_dim rounded up to a multiple of 4
    int n_basis_cols;       // sum of the real appearance channels of the sampled planes
    int ca_total;           // padded appearance slots (multiple of 4) = sum 4*ca4
    const float* color_table;  // (color_table_views, 12) per-camera [3x3 | shift], or NULL
    // point_prediction cascades (hr_model_create_cascade)
    int rows_per_ray;       // head rows per ray: 1, or casc_in_z when the head comes from the point MLP
                            //   (sample k reads row ray*rows_per_ray + k / M at columns (k % M) * P.., M = Z / rows_per_ray)
    float* rows_out;        // coarse pass only: input rows of the point MLP, (n_rays * Z, row_dim); no colour is produced
    int row_dim, n_row_inputs;
    int row_kind[4], row_len[4];   // HR_PIN_* and columns of each input
    // optional occupancy early-reject (hr_model_set_occupancy): AlphaGridMask.sample_alpha of the raw point (utils/tensorf_utils.py:
    // 459-484) must be > 0 for a sample to be gathered -- the test the reference carries at tensorf_no_sample.py:171-177
    const float* occ;       // (D, H, W) float volume or NULL
    const unsigned* occ_cells;   // one bit per lattice CELL (W-1, H-1, D-1; x fastest): any of its 8 corners set -- for a point strictly
                                 //   inside a cell of a 0/1 volume the trilinear sample is > 0 exactly when that bit is set; NULL: volume not binary
    int occ_w, occ_h, occ_d;
    float occ_lo[3], occ_inv[3];   // g = (p - lo) * inv - 1
    int dbg_mode;           // measurement builds only (-DHR_TUNING, HR_SAMPLE_DBG): 1 = skip the feature gather
    // verified fast path.  First pass: redo_list != NULL -- a ray with a comparison within redo_band of flipping is appended as
    // ray0 + (its index in this launch): ray0 = where the launch's `rays` / `rgb` start in the caller's buffers (hr_render walks them in chunks).
    // Second pass: `rays` / `rgb` are the caller's whole buffers and ray_index != NULL -- position i of the launch (head row i) is the caller's
    // ray ray_index[i]; *n_rays_dev of them (n_rays is then the launch's capacity).
    int64_t ray0;
    const int* ray_index;
    const unsigned* n_rays_dev;
    int64_t list_off;       // second pass in slices: ray_index starts at entry list_off of the list, *n_rays_dev - list_off remain
    unsigned* zero_word;    // second pass: cleared by workgroup 0 (the list's counter, for the next call; this launch reads its copy)
    int* redo_list;
    unsigned* redo_count;
    int redo_cap;
    float redo_band;        // HrRisk::band_zc: the model's calibrated margin of  z * scale + anchor  (hr_math.h)
    float redo_band_q;      // HrRisk::band_q: margin of a point coordinate per unit of amplification
    float redo_band_off;    // HrRisk::band_off: margin of the point-offset / flow heads
    float redo_amp_cut;     // HrRisk::amp_cut: a live sample conditioned worse than the calibration's rays lists its ray
    unsigned* flags;        // the model's sticky status word (bit 2: the redo list overflowed)
#ifdef HR_DEBUG_HSUM        // measurement builds (tools/hsum_bisect.py): per ray, the XOR of the bits of every head value the sample stage read, and of its sorted distances
    unsigned* dbg_hsum;     // [n_rays][2]
#endif
};

// training forward (mlp_split_impl.inc, HR_SPLIT_TRAIN_KERNEL): where the output of hidden Linear l (after its LeakyReLU) goes besides LDS --
// act[l] + ray * ld[l] + off[l] + feature, fp32 (NULL: not kept)
struct HrMlpTaps {
    float* act[HR_MAX_LAYERS];
    int64_t ld[HR_MAX_LAYERS];
    int off[HR_MAX_LAYERS];
};
void hr_launch_mlp_train_bf16x3(const hr_config& cfg, const HrMlpArgs& args, const HrMlpTaps& taps, hipStream_t stream);
// one layer's reference-layout weights (N_user, Kt) / bias (N_user) in device memory -> the split kernels' bf16 hi / lo tiles and padded
// bias, on the device (what pack_mlp does on the host at finalize; the training step re-packs every step)
struct HrPackDesc {
    const float* w;
    const float* b;
    void* wsplit;
    float* bias;
    int N_user, Kt;         // the torch matrix
    int N, nt, Kp;          // rows the kernel computes, their 32-row tiles, padded K
    int first, skip, last;
    int mlp_in, k0p;
    int P_user, P_live;
    int live_cols[64];      // last layer: live column c' of a sample -> the user's column
};
void hr_launch_pack_split_bf16(const HrPackDesc& d, hipStream_t stream);
void hr_launch_mlp(const hr_config& cfg, const HrMlpArgs& args, hipStream_t stream);
// split-precision form: wsplit[L][(((kt * n_tiles[L] + nt) * 2 + part) * 64 + lane)] = 8 bf16 of
//   W[n = 32*nt + (lane & 31)][k = 16*kt + 8*(lane >> 5) + 0..7], part 0 = hi (bf16(w)), 1 = lo (bf16(w - hi))
void hr_launch_mlp_bf16x3(const hr_config& cfg, const HrMlpArgs& args, hipStream_t stream);
void hr_launch_mlp_f16x3(const hr_config& cfg, const HrMlpArgs& args, hipStream_t stream);   // same layouts, fp16 halves
void hr_launch_mlp_f16x2(const hr_config& cfg, const HrMlpArgs& args, hipStream_t stream);   // fp16, weights unsplit
void hr_launch_mlp_f16f8(const hr_config& cfg, const HrMlpArgs& args, hipStream_t stream);   // fp16 leading product, the two correction products as one fp8 K=64 MFMA
void hr_launch_samples(const hr_config& cfg, const HrSampleArgs& args, hipStream_t stream);
// fused frame kernel (fused_impl.inc): MLP + sample stage of all rays in one persistent launch, head tile in LDS.
// Returns false when the model does not fit it (nothing launched); probe: only answer.
bool hr_launch_frame_bf16x3(const hr_config& cfg, const HrMlpArgs& ma, const HrSampleArgs& sa, int sample_waves, int frame_mode, int n_cus, bool probe,
                            hipStream_t stream);
bool hr_launch_frame_f16x3(const hr_config& cfg, const HrMlpArgs& ma, const HrSampleArgs& sa, int sample_waves, int frame_mode, int n_cus, bool probe,
                            hipStream_t stream);
bool hr_launch_frame_f16x2(const hr_config& cfg, const HrMlpArgs& ma, const HrSampleArgs& sa, int sample_waves, int frame_mode, int n_cus, bool probe,
                            hipStream_t stream);
bool hr_launch_frame_f16f8(const hr_config& cfg, const HrMlpArgs& ma, const HrSampleArgs& sa, int sample_waves, int frame_mode, int n_cus, bool probe,
                            hipStream_t stream);

// activation range of the MLP on a set of rays (range_kernel.hip): act_max[0] = max |input feature|, act_max[l + 1] = max |pre-activation|
// of hidden Linear l; w / b: the uploaded reference-layout tensors (out, in) / (out)
struct HrRangeArgs {
    const float* rays;
    int64_t n_rays;
    const float* w[HR_MAX_LAYERS];
    const float* b[HR_MAX_LAYERS];
    float* act_max;              // [HR_MAX_LAYERS] floats, zeroed by the caller; atomically maximised
};
void hr_launch_mlp_range(const hr_config& cfg, const HrRangeArgs& a, hipStream_t stream);
bool hr_mlp_range_supported(const hr_config& cfg);
void hr_launch_synthetic_rays(float* rays, int64_t n, int ray_dim, const float lo[3], const float hi[3], unsigned seed, hipStream_t stream);

// band probe of the verified fast path (band_kernel.hip): f16f8 vs f16x3 heads of the same rays in the user's (n, Z, P) layout ->
// the largest differences of what the sample stage compares.  stats: HR_BAND_WORDS unsigned words, zeroed by the caller
// (non-negative floats as their bit patterns, atomically maximised; counters)
enum { HR_BAND_ZC = 0,        // max |zc(a) - zc(b)|: the length before the inverse contraction
       HR_BAND_DIST_N = 1,    // max |distance(a) - distance(b)| / (dlen amp): the same thing seen through the intersection (the margins' model, checked)
       HR_BAND_GEO_N = 2,     // max |point(dist a, head b) - point(dist b, head b)| / amp: the points' share of a distance error, per unit of amplification
       HR_BAND_OFF = 3,       // max |point(dist a, head a) - point(dist a, head b)|: offset + flow heads
       HR_BAND_COUNTED = 4, HR_BAND_FLIPPED = 5, HR_BAND_SHAKY = 6,
       HR_BAND_DIST = 7,      // max |distance(a) - distance(b)| as it is (reported)
       HR_BAND_HEAD0 = 8, HR_BAND_WORDS = 72 };
struct HrBandArgs {
    const hr_config* cfg_dev;    // the caller's configuration (user column order) with isect_mask_off = 1
    const float* rays;
    const float* head_a;         // (n, Z, P) raw head, cheap arithmetic
    const float* head_b;         // the same rays, reference-grade arithmetic
    int64_t n_rays;
    int mask_on;                 // the model masks distances outside (near, far): only samples alive under both heads count
    float flip_cut;              // a normalised distance difference beyond this is a decision that fell the other way, not arithmetic error
    unsigned* stats;
    unsigned char* ray_ok;       // one byte per ray: 1 = every live sample of the ray has amp <= amp_cut (a well-conditioned ray: every statistic
    float amp_cut;               //   is taken over these).  phase 0: cleared by the samples that say otherwise; phase 1: read
    int phase;                   // 0: only ray_ok; 1: the statistics, over the rays phase 0 left marked
};
void hr_launch_band_probe(const HrBandArgs& a, int z_channels, hipStream_t stream);

void hr_launch_generate_rays(const hr_camera& cam, int ray_dim, int64_t first_pixel, int64_t n_pixels, float* rays, hipStream_t stream);

// layout kernels (pack_kernels.hip)
// dst[y][x][c_off + c] = src[c][y][x] for c < C  (dst texel stride = tex floats)
// Per-sample head columns the path actually reads (hr_model_finalize drops the others from the
// last Linear): col[c] = position of user column c among the live ones, or -1.
struct HrColMap {
    int col[64];
};
// diagnostics export of the raw head in the user's (n, Z*P) layout; pruned columns read as 0
void hr_launch_head_export(const float* head, float* out, int64_t n_rays, int Z, int P, int P_live, int nq, int rows_per_ray,
                           const HrColMap& map,
                           hipStream_t stream);
void hr_launch_upsample_plane(const float* src, int C, int H, int W, float* dst, int H2, int W2, hipStream_t stream);
void hr_launch_pack_display(const float* rgb, int h, int w, int transpose, int flip, int rgba8, void* out, hipStream_t stream);
void hr_launch_plane_reg_forward(const float* p, int C, int H, int W, float* sums, hipStream_t stream);
void hr_launch_plane_reg_backward(const float* p, int C, int H, int W, const float* coef, float* grad, hipStream_t stream);
// optimizer step of the training loop (utils/__init__.py:49-76 get_optimizer -> torch.optim.Adam(betas=(0.9, 0.99), eps=1e-8)): every parameter
// tensor of a step in ONE launch.  A block owns 4096 consecutive elements of one tensor; first_block[] is the prefix sum of the tensors' block counts
constexpr int HR_ADAM_MAX_TENSORS = 40;
struct HrAdamBatch {
    float* p[HR_ADAM_MAX_TENSORS];
    const float* g[HR_ADAM_MAX_TENSORS];
    float* m[HR_ADAM_MAX_TENSORS];
    float* v[HR_ADAM_MAX_TENSORS];
    int64_t n[HR_ADAM_MAX_TENSORS];
    float step_size[HR_ADAM_MAX_TENSORS];      // lr / (1 - beta1^t)
    float inv_sqrt_bc2[HR_ADAM_MAX_TENSORS];   // 1 / sqrt(1 - beta2^t)
    float omb1[HR_ADAM_MAX_TENSORS], beta2[HR_ADAM_MAX_TENSORS], omb2[HR_ADAM_MAX_TENSORS];     // 1 - beta1, beta2, 1 - beta2 (the differences formed in double)
    float eps[HR_ADAM_MAX_TENSORS], weight_decay[HR_ADAM_MAX_TENSORS];
    int first_block[HR_ADAM_MAX_TENSORS + 1];
    int count;
};
void hr_launch_adam(const HrAdamBatch& b, hipStream_t stream);
// the training step's re-layouts in ONE launch per direction (reference (C, H, W) tensors <-> packed channel-last texels): up to 12 jobs
struct HrLayoutJob { const float* src; float* dst; int C, H, W, tex, c_off; };
struct HrLayoutBatch { HrLayoutJob job[12]; int n; };
void hr_launch_layout_batch(const HrLayoutBatch& b, bool to_packed, hipStream_t stream);
void hr_launch_basis_transpose(const float* basis, float* basis_t, int app_dim, int n_cols, int ld, hipStream_t stream);
void hr_launch_blend_rows(const float* b, float* line, int row_floats, int i0, int i1, float w0, float w1, hipStream_t stream);
void hr_launch_interleave(const float* src, void* dst, int half, int C, int H, int W, int tex, int c_off, hipStream_t stream);


// ---------------------------------------------------------------- training path (train_kernel.hip, hr_train.h)
struct HrTrainArgs;
struct HrMaskArgs;
struct HrRowsArgs;
void hr_launch_rows(const hr_config& cfg, const HrRowsArgs& args, hipStream_t stream);
void hr_launch_dense_alpha(const HrMaskArgs& args, hipStream_t stream);
void hr_launch_train(const hr_config& cfg, const HrTrainArgs& args, hipStream_t stream);
// the deterministic build (train_det_kernel.hip): `args_flt` is the default build's HrTrainArgs whose accumulator pointers (g_a, g_b,
// d_basis, d_color_table) point to 64-bit fixed-point buffers of the same element counts (hr_train.h: hr_acc_t; the unit is chosen per step and
// kept in args.fx, the model's own)
void hr_launch_train_det(const hr_config& cfg, const void* args_flt, size_t args_bytes, hipStream_t stream);
void hr_launch_fixed_to_float(const long long* src, float* dst, int64_t n, const float* inv_dev, const unsigned* bad_dev, hipStream_t stream);   // dst[i] = src[i] * *inv_dev (NaN if *bad_dev)
void hr_launch_features(const hr_config* cfg_dev, const float* rays, int64_t n, float* out, hipStream_t stream);
// training GEMMs of the MLP (train_gemm_kernel.hip)
size_t hr_linear_workspace_bytes(int64_t rows, int in, int out);
void hr_launch_linear_forward(const float* x, int64_t ldx, int64_t rows, int in, const float* w, const float* b, int out, float slope,
                              float* y, int64_t ldy, hipStream_t stream);
void hr_launch_linear_backward(const float* x, int64_t ldx, const float* w, const float* y, int64_t ldy, const float* dy, int64_t ld_dy,
                               int64_t rows, int in, int out, float slope, float* dx, int64_t ld_dx, float* dw, float* db, float* workspace,
                               hipStream_t stream);

#endif
