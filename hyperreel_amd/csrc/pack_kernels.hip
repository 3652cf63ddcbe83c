// One-time layout transforms run by hr_model_finalize (not on the render path).
#include "hr_kernels.h"
#include "hr_math.h"

// Reference planes are channel-first (1, C, H, W) (nlf/nets/tensorf_base.py:911-948,
// nlf/nets/tensorf_dynamic.py:126-173).  The sample kernel wants channel-last texels with
// the density and appearance channels of one plane side by side:
//   dst[(y*W + x)*tex + c_off + c] = src[(c*H + y)*W + x]
__global__ void hr_interleave_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W,
                                     int tex, int c_off)
{
    const int64_t n = (int64_t)C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        // i enumerates the destination order (y, x, c) so that writes are contiguous per texel
        const int c = (int)(i % C);
        const int64_t yx = i / C;
        dst[yx * tex + c_off + c] = src[(int64_t)c * H * W + yx];
    }
}

// float16 texels (HR_GRID_FP16): same layout in halfs, values rounded to nearest even once, here.
__global__ void hr_interleave_half_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, int C, int H, int W,
                                          int tex, int c_off)
{
    const int64_t n = (int64_t)C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t yx = i / C;
        dst[yx * tex + c_off + c] = (_Float16)src[(int64_t)c * H * W + yx];
    }
}

// out[r][k*P + c] = head[r][k*P_live + col_map[c]] (0 where the column was pruned)
__global__ void hr_head_export_kernel(const float* __restrict__ head, float* __restrict__ out, int64_t n_rays, int Z, int P,
                                      int P_live, int nq, int rows_per_ray, HrColMap map)
{
    const int M = Z / rows_per_ray;   // samples per head row (Z unless the head comes from a point MLP)
    const int n_out = Z * P;
    const int64_t total = n_rays * n_out;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n_out;
        const int n = (int)(i - r * n_out);
        const int k = n / P, c = n - k * P;
        const int cn = map.col[c];
        out[i] = (cn >= 0) ? head[hr_head_index(r * rows_per_ray + k / M, (k % M) * P_live + cn, nq)] : 0.0f;
    }
}

void hr_launch_head_export(const float* head, float* out, int64_t n_rays, int Z, int P, int P_live, int nq, int rows_per_ray,
                           const HrColMap& map, hipStream_t stream)
{
    const int64_t total = n_rays * Z * P;
    if (total <= 0) return;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(hr_head_export_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, head, out, n_rays, Z, P, P_live, nq,
                       rows_per_ray, map);
}

// Camera -> rays (utils/ray_utils.py:98-135, datasets/base.py:485-518): pixel centres +0.5,
// directions (x, -y, -1) / focal, rotated by the pose, normalised; origin = pose translation.
__global__ void hr_generate_rays_kernel(const hr_camera cam, int ray_dim, int64_t first_pixel, int64_t n_pixels, float* __restrict__ rays)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_pixels; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = first_pixel + t;
        const float i = (float)(p % cam.width), j = (float)(p / cam.width);
        const float dx = (i - cam.cx + 0.5f) / cam.fx;
        const float dy = -(j - cam.cy + 0.5f) / cam.fy;
        const float dz = -1.0f;
        float wx = dx * cam.c2w[0] + dy * cam.c2w[1] + dz * cam.c2w[2];
        float wy = dx * cam.c2w[4] + dy * cam.c2w[5] + dz * cam.c2w[6];
        float wz = dx * cam.c2w[8] + dy * cam.c2w[9] + dz * cam.c2w[10];
        const float nrm = fmaxf(sqrtf(wx * wx + wy * wy + wz * wz), 1e-12f);   // F.normalize(p=2, eps=1e-12)
        wx = wx / nrm; wy = wy / nrm; wz = wz / nrm;
        float* r = rays + t * ray_dim;
        r[0] = cam.c2w[3]; r[1] = cam.c2w[7]; r[2] = cam.c2w[11];
        r[3] = wx; r[4] = wy; r[5] = wz;
        if (ray_dim == 8) { r[6] = cam.cam_id; r[7] = cam.time; }
    }
}

void hr_launch_generate_rays(const hr_camera& cam, int ray_dim, int64_t first_pixel, int64_t n_pixels, float* rays, hipStream_t stream)
{
    if (n_pixels <= 0) return;
    int64_t blocks = (n_pixels + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(hr_generate_rays_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, cam, ray_dim, first_pixel, n_pixels, rays);
}

// upsample_bilinear2d with align_corners=True, ATen's arithmetic (UpSample.h area_pixel_compute_scale / _source_index,
// UpSampleBilinear2d): scale = (in - 1) / (out - 1) in fp32 (0 when out == 1), src = scale * dst_index,
// i0 = (int)src, i1 = i0 + (i0 < in - 1), l1 = src - i0, l0 = 1 - l1,
// out = l0h * (l0w * v00 + l1w * v01) + l1h * (l0w * v10 + l1w * v11).
__global__ void hr_upsample_plane_kernel(const float* __restrict__ src, int C, int H, int W, float* __restrict__ dst, int H2, int W2)
{
    const float sh = (H2 > 1) ? (float)(H - 1) / (float)(H2 - 1) : 0.0f;
    const float sw = (W2 > 1) ? (float)(W - 1) / (float)(W2 - 1) : 0.0f;
    const int64_t n = (int64_t)C * H2 * W2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x2 = (int)(i % W2);
        const int y2 = (int)((i / W2) % H2);
        const int c = (int)(i / ((int64_t)W2 * H2));
        const float fy = sh * (float)y2, fx = sw * (float)x2;
        const int y0 = (int)fy, x0 = (int)fx;
        const int yp = (y0 < H - 1) ? 1 : 0, xp = (x0 < W - 1) ? 1 : 0;
        const float l1h = fy - (float)y0, l0h = 1.0f - l1h;
        const float l1w = fx - (float)x0, l0w = 1.0f - l1w;
        const float* p = src + ((int64_t)c * H + y0) * W + x0;
        const float v00 = p[0], v01 = p[xp], v10 = p[(int64_t)yp * W], v11 = p[(int64_t)yp * W + xp];
        dst[i] = l0h * (l0w * v00 + l1w * v01) + l1h * (l0w * v10 + l1w * v11);
    }
}

void hr_launch_upsample_plane(const float* src, int C, int H, int W, float* dst, int H2, int W2, hipStream_t stream)
{
    const int64_t n = (int64_t)C * H2 * W2;
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(hr_upsample_plane_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, C, H, W, dst, H2, W2);
}

void hr_launch_interleave(const float* src, void* dst, int half, int C, int H, int W, int tex, int c_off, hipStream_t stream)
{
    const int64_t n = (int64_t)C * H * W;
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (half)
        hipLaunchKernelGGL(hr_interleave_half_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src,
                           reinterpret_cast<_Float16*>(dst), C, H, W, tex, c_off);
    else
        hipLaunchKernelGGL(hr_interleave_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src,
                           reinterpret_cast<float*>(dst), C, H, W, tex, c_off);
}

// All re-layouts of a training step in one launch: blockIdx.y = job, blockIdx.x strides over the job's elements.
//   TO_PACKED:  dst[(y*W + x)*tex + c_off + c] = src[(c*H + y)*W + x]   (parameters -> texels, before the forward)
//   else:       dst[(c*H + y)*W + x] = src[(y*W + x)*tex + c_off + c]   (texel gradients -> parameter gradients)
template <bool TO_PACKED>
__global__ __launch_bounds__(256) void hr_layout_batch_kernel(const HrLayoutBatch b)
{
    const HrLayoutJob j = b.job[blockIdx.y];
    const int64_t hw = (int64_t)j.H * j.W, n = hw * j.C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (TO_PACKED) {                           // i enumerates the destination order (y, x, c): contiguous writes per texel
            const int c = (int)(i % j.C);
            const int64_t yx = i / j.C;
            j.dst[yx * j.tex + j.c_off + c] = j.src[(int64_t)c * hw + yx];
        } else {                                   // i enumerates (c, y, x), x fastest: coalesced stores
            const int c = (int)(i / hw);
            const int64_t yx = i - (int64_t)c * hw;
            j.dst[i] = j.src[yx * j.tex + j.c_off + c];
        }
    }
}

// The same through LDS, for jobs of up to HR_LAYOUT_MAX_C channels: a workgroup takes HR_LAYOUT_TILE consecutive texels of a job, reads
// them the way the source is laid out (planar: one run of the tile per channel; packed: the job's C channels of each texel, C * 4
// contiguous bytes) and writes them the way the destination is.  The direct form above reads ONE float of a 64-byte texel per lane on
// the packed side: 81 us for the 46 MB of the 600^3 scene's gradients (1.1 TB/s); this one 16-byte..64-byte runs on both sides.
#define HR_LAYOUT_TILE 256
#define HR_LAYOUT_MAX_C 32
template <bool TO_PACKED>
__global__ __launch_bounds__(256) void hr_layout_tiled_kernel(const HrLayoutBatch b)
{
    extern __shared__ float tile[];                // [C][HR_LAYOUT_TILE + 8]  (row stride = 8 mod 32 banks: the channel-fastest phase spreads 8 channels x 8 texels over all banks)
    const HrLayoutJob j = b.job[blockIdx.y];
    const int64_t hw = (int64_t)j.H * j.W;
    const int64_t tiles = (hw + HR_LAYOUT_TILE - 1) / HR_LAYOUT_TILE;
    const int C = j.C;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t yx0 = t * HR_LAYOUT_TILE;
        const int nt = (int)((hw - yx0 < HR_LAYOUT_TILE) ? hw - yx0 : HR_LAYOUT_TILE);
        __syncthreads();
        if (TO_PACKED) {
            for (int e = threadIdx.x; e < C * HR_LAYOUT_TILE; e += 256) {          // planar source: x fastest
                const int c = e / HR_LAYOUT_TILE, x = e - c * HR_LAYOUT_TILE;
                if (x < nt) tile[c * (HR_LAYOUT_TILE + 8) + x] = j.src[(int64_t)c * hw + yx0 + x];
            }
            __syncthreads();
            for (int e = threadIdx.x; e < C * nt; e += 256) {                      // packed destination: channel fastest
                const int x = e / C, c = e - x * C;
                j.dst[(yx0 + x) * j.tex + j.c_off + c] = tile[c * (HR_LAYOUT_TILE + 8) + x];
            }
        } else {
            for (int e = threadIdx.x; e < C * nt; e += 256) {                      // packed source: channel fastest
                const int x = e / C, c = e - x * C;
                tile[c * (HR_LAYOUT_TILE + 8) + x] = j.src[(yx0 + x) * j.tex + j.c_off + c];
            }
            __syncthreads();
            for (int e = threadIdx.x; e < C * HR_LAYOUT_TILE; e += 256) {          // planar destination: x fastest
                const int c = e / HR_LAYOUT_TILE, x = e - c * HR_LAYOUT_TILE;
                if (x < nt) j.dst[(int64_t)c * hw + yx0 + x] = tile[c * (HR_LAYOUT_TILE + 8) + x];
            }
        }
    }
}

void hr_launch_layout_batch(const HrLayoutBatch& b, bool to_packed, hipStream_t stream)
{
    if (b.n <= 0) return;
    int64_t most = 0, most_hw = 0;
    int max_c = 0;
    for (int i = 0; i < b.n; ++i) {
        const int64_t hw = (int64_t)b.job[i].H * b.job[i].W;
        most = most > hw * b.job[i].C ? most : hw * b.job[i].C;
        most_hw = most_hw > hw ? most_hw : hw;
        max_c = max_c > b.job[i].C ? max_c : b.job[i].C;
    }
    if (most <= 0) return;
    if (max_c <= HR_LAYOUT_MAX_C) {
        int64_t blocks = (most_hw + HR_LAYOUT_TILE - 1) / HR_LAYOUT_TILE;
        if (blocks > 1024) blocks = 1024;
        const dim3 grid((unsigned)blocks, (unsigned)b.n);
        const size_t lds = sizeof(float) * (size_t)max_c * (HR_LAYOUT_TILE + 8);
        if (to_packed) hipLaunchKernelGGL(hr_layout_tiled_kernel<true>, grid, dim3(256), lds, stream, b);
        else hipLaunchKernelGGL(hr_layout_tiled_kernel<false>, grid, dim3(256), lds, stream, b);
        return;
    }
    int64_t blocks = (most + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const dim3 grid((unsigned)blocks, (unsigned)b.n);
    if (to_packed) hipLaunchKernelGGL(hr_layout_batch_kernel<true>, grid, dim3(256), 0, stream, b);
    else hipLaunchKernelGGL(hr_layout_batch_kernel<false>, grid, dim3(256), 0, stream, b);
}

// ---------------------------------------------------------------- TensoRF plane regularisers
// TVLoss (nlf/regularizers/tensorf.py:14-34) and density_L1 (nlf/nets/tensorf_base.py:1024-1035) of one (1, C, H, W)
// plane in one pass over it: sums += { sum (x[y] - x[y-1])^2, sum (x[x] - x[x-1])^2, sum |x| }.  The reference's torch
// expression reads the plane five times and materialises four temporaries of its size; this is one read (neighbours
// come from cache) and 3 atomics per workgroup.
__global__ __launch_bounds__(256) void hr_plane_reg_fwd_kernel(const float* __restrict__ p, int64_t n, int H, int W, float* __restrict__ sums)
{
    float sh = 0.0f, sw = 0.0f, sl = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const float v = p[i];
        if (y > 0) { const float d = v - p[i - W]; sh = __builtin_fmaf(d, d, sh); }
        if (x > 0) { const float d = v - p[i - 1]; sw = __builtin_fmaf(d, d, sw); }
        sl += fabsf(v);
    }
    for (int d = 32; d > 0; d >>= 1) { sh += __shfl_xor(sh, d, 64); sw += __shfl_xor(sw, d, 64); sl += __shfl_xor(sl, d, 64); }
    __shared__ float part[4][3];
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][0] = sh; part[threadIdx.x >> 6][1] = sw; part[threadIdx.x >> 6][2] = sl; }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(sums + threadIdx.x, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// grad = coef[0] * d(sum_h)/dx + coef[1] * d(sum_w)/dx + coef[2] * sign(x), coef on the device (the upstream gradient)
__global__ __launch_bounds__(256) void hr_plane_reg_bwd_kernel(const float* __restrict__ p, int64_t n, int H, int W, const float* __restrict__ coef,
                                                               float* __restrict__ grad)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float ch = coef[0], cw = coef[1], cl = coef[2];
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const float v = p[i];
    float gh = 0.0f, gw = 0.0f;
    if (y > 0) gh += v - p[i - W];
    if (y < H - 1) gh -= p[i + W] - v;
    if (x > 0) gw += v - p[i - 1];
    if (x < W - 1) gw -= p[i + 1] - v;
    const float sgn = (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f);
    grad[i] = ch * (2.0f * gh) + cw * (2.0f * gw) + cl * sgn;
}

void hr_launch_plane_reg_forward(const float* p, int C, int H, int W, float* sums, hipStream_t stream)
{
    const int64_t n = (int64_t)C * H * W;
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(hr_plane_reg_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, n, H, W, sums);
}

void hr_launch_plane_reg_backward(const float* p, int C, int H, int W, const float* coef, float* grad, hipStream_t stream)
{
    const int64_t n = (int64_t)C * H * W;
    if (n <= 0) return;
    hipLaunchKernelGGL(hr_plane_reg_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, n, H, W, coef, grad);
}

// ---------------------------------------------------------------- display pack
// rgb (h * w, 3) fp32 -> the viewer's buffer: optional transpose + vertical flip (utils/gui_utils.py:199-205) and either
// 8-bit RGBA (to8b, utils/__init__.py:47; alpha 255) or fp32 RGB (what dearpygui's raw texture takes).  One pass on the
// device instead of `.cpu().numpy()` + numpy transpose / flip / ascontiguousarray on the host; a quarter of the bytes
// cross PCIe when the consumer wants 8-bit pixels.
__global__ __launch_bounds__(256) void hr_pack_display_kernel(const float* __restrict__ rgb, int h, int w, int transpose, int flip, int rgba8,
                                                              void* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)h * w) return;
    const int ow = transpose ? h : w;
    const int64_t src = hr_display_src_pixel((int)(i / ow), (int)(i % ow), h, w, transpose, flip);
    const float r = rgb[src * 3 + 0], g = rgb[src * 3 + 1], b = rgb[src * 3 + 2];
    if (rgba8) {
        uchar4 px;
        px.x = hr_to8b(r); px.y = hr_to8b(g); px.z = hr_to8b(b); px.w = 255;
        reinterpret_cast<uchar4*>(out)[i] = px;
    } else {
        float* o = reinterpret_cast<float*>(out) + i * 3;
        o[0] = r; o[1] = g; o[2] = b;
    }
}

void hr_launch_pack_display(const float* rgb, int h, int w, int transpose, int flip, int rgba8, void* out, hipStream_t stream)
{
    const int64_t n = (int64_t)h * w;
    if (n <= 0) return;
    hipLaunchKernelGGL(hr_pack_display_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rgb, h, w, transpose, flip, rgba8, out);
}


// basis_mat (app_dim, n_cols) row-major -> the column-major copy the render kernels fold decode matrices from (HrSampleArgs::basis_t)
__global__ void hr_basis_transpose_kernel(const float* __restrict__ basis, float* __restrict__ basis_t, int app_dim, int n_cols, int ld)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= app_dim * n_cols) return;
    const int r = i / n_cols, col = i - r * n_cols;
    basis_t[(size_t)col * ld + r] = basis[i];
}

void hr_launch_basis_transpose(const float* basis, float* basis_t, int app_dim, int n_cols, int ld, hipStream_t stream)
{
    const int n = app_dim * n_cols;
    if (n <= 0) return;
    hipLaunchKernelGGL(hr_basis_transpose_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, basis, basis_t, app_dim, n_cols, ld);
}

// hr_render_frame: the two keyframe rows a frame's time blends between, folded into one line per time plane:
// line[x][ch] = B[row0][x][ch] * w0 + B[row1][x][ch] * w1 (float32 texels)
__global__ __launch_bounds__(256) void hr_blend_rows_kernel(const float* __restrict__ b, float* __restrict__ line, int n, int i0, int i1, float w0, float w1)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    line[i] = fmaf(b[(size_t)i1 * n + i], w1, b[(size_t)i0 * n + i] * w0);
}

void hr_launch_blend_rows(const float* b, float* line, int row_floats, int i0, int i1, float w0, float w1, hipStream_t stream)
{
    if (row_floats <= 0) return;
    hipLaunchKernelGGL(hr_blend_rows_kernel, dim3((unsigned)((row_floats + 255) / 256)), dim3(256), 0, stream, b, line, row_floats, i0, i1, w0, w1);
}

// 64-bit fixed-point gradient totals of the deterministic training build (hr_train.h: the step's unit, kept per model) -> float, once, after the last add
__global__ __launch_bounds__(256) void hr_fixed_to_float_kernel(const long long* __restrict__ src, float* __restrict__ dst, int64_t n,
                                                                 const float* __restrict__ inv_dev, const unsigned* __restrict__ bad_dev)
{
    const double inv = (double)*inv_dev;                 // the step's unit (a power of two: exact)
    const bool bad = *bad_dev != 0u;                     // a non-finite contribution: the fp32 path would hold inf / NaN somewhere -- say so everywhere
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dst[i] = bad ? __builtin_nanf("") : (float)((double)src[i] * inv);
}

void hr_launch_fixed_to_float(const long long* src, float* dst, int64_t n, const float* inv_dev, const unsigned* bad_dev, hipStream_t stream)
{
    if (n <= 0) return;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(hr_fixed_to_float_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, n, inv_dev, bad_dev);
}


// ---------------------------------------------------------------- device-side weight packing for the training step's fused MLP forward
// Same layout and roundings as pack_mlp (api.hip) for the bf16 split: wsplit[(((kt * nt + t) * 2 + part) * 64 + lane) * 8 + j] =
// W[n = 32 t + (lane & 31)][k = 16 kt + 8 (lane >> 5) + j], part 0 = bf16(w), part 1 = bf16(w - hi); K order: layer 0 the input features
// padded to k0p, skip layers [input padded to k0p | hidden]; last layer: kernel row n = k * P_live + c' is the user's row
// k * P_user + live_cols[c'] (BaseMLP's weights, nlf/nets/mlp.py:127-172, as torch stores them: (out, in)).
__global__ __launch_bounds__(256) void hr_pack_split_bf16_kernel(const HrPackDesc d)
{
    const int64_t total = (int64_t)(d.Kp / 16) * d.nt * 64 * 8;
    uint16_t* out = reinterpret_cast<uint16_t*>(d.wsplit);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const int64_t tt = i >> 9;
        const int t = (int)(tt % d.nt), kt = (int)(tt / d.nt);
        const int n = 32 * t + (lane & 31), kk = 16 * kt + 8 * (lane >> 5) + j;
        int col = -1;
        if (d.first) { if (kk < d.mlp_in) col = kk; }
        else if (d.skip) { if (kk < d.k0p) { if (kk < d.mlp_in) col = kk; } else col = d.mlp_in + (kk - d.k0p); }
        else col = kk;
        float v = 0.0f;
        if (n < d.N && col >= 0 && col < d.Kt) {
            const int row = d.last ? (n / d.P_live) * d.P_user + d.live_cols[n % d.P_live] : n;
            v = d.w[(int64_t)row * d.Kt + col];
        }
        const __bf16 hi = (__bf16)v;
        const __bf16 lo = (__bf16)(v - (float)hi);
        const int64_t base = (((tt * 2) * 64 + lane) * 8) + j;
        out[base] = __builtin_bit_cast(uint16_t, hi);
        out[base + 64 * 8] = __builtin_bit_cast(uint16_t, lo);
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < d.nt * 32; i += 256)
            d.bias[i] = i < d.N ? d.b[d.last ? (i / d.P_live) * d.P_user + d.live_cols[i % d.P_live] : i] : 0.0f;
}

void hr_launch_pack_split_bf16(const HrPackDesc& d, hipStream_t stream)
{
    const int64_t total = (int64_t)(d.Kp / 16) * d.nt * 64 * 8;
    if (total <= 0) return;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(hr_pack_split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d);
}


// ---------------------------------------------------------------------------------------------------------
// Adam over every parameter tensor of a training step in one launch (torch.optim.Adam's single-tensor arithmetic, torch/optim/adam.py
// _single_tensor_adam: grad += wd * p; exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) grad^2;
// denom = sqrt(exp_avg_sq) / sqrt(bias_correction2) + eps; p -= (lr / bias_correction1) * exp_avg / denom).  HBM-bound: 16 bytes read and
// 12 written per parameter; the foreach form torch runs by default makes eleven passes over the same arrays.
__global__ __launch_bounds__(256) void hr_adam_kernel(const HrAdamBatch b)
{
    int t = 0;
    const int blk = (int)blockIdx.x;
    while (t + 1 < b.count && b.first_block[t + 1] <= blk) ++t;              // (wave-uniform: scalar loads from the kernel arguments)
    float* __restrict__ p = b.p[t];
    const float* __restrict__ g = b.g[t];
    float* __restrict__ m = b.m[t];
    float* __restrict__ v = b.v[t];
    const int64_t n = b.n[t];
    const float step_size = b.step_size[t], isb2 = b.inv_sqrt_bc2[t], omb1 = b.omb1[t], b2 = b.beta2[t], omb2 = b.omb2[t], eps = b.eps[t], wd = b.weight_decay[t];
    const int64_t base = (int64_t)(blk - b.first_block[t]) * 4096;
    auto one = [&](float& pv, float gv, float& mv, float& vv) {
        if (wd != 0.0f) gv = __builtin_fmaf(wd, pv, gv);
        mv = mv + omb1 * (gv - mv);                          // lerp_(grad, 1 - beta1), weight < 0.5
        vv = vv * b2 + (omb2 * gv) * gv;                     // mul_(beta2).addcmul_(grad, grad.conj(), value = 1 - beta2)
        const float denom = __builtin_sqrtf(vv) * isb2 + eps;
        pv = pv - step_size * (mv / denom);
    };
    const bool vec = (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = base + (int64_t)j * 1024 + 4 * threadIdx.x;
        if (vec && i + 4 <= n) {
            float4 pv = *reinterpret_cast<const float4*>(p + i), mv = *reinterpret_cast<const float4*>(m + i), vv = *reinterpret_cast<const float4*>(v + i);
            const float4 gv = *reinterpret_cast<const float4*>(g + i);
            one(pv.x, gv.x, mv.x, vv.x); one(pv.y, gv.y, mv.y, vv.y); one(pv.z, gv.z, mv.z, vv.z); one(pv.w, gv.w, mv.w, vv.w);
            *reinterpret_cast<float4*>(p + i) = pv; *reinterpret_cast<float4*>(m + i) = mv; *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (int e = 0; e < 4; ++e)
                if (i + e < n) one(p[i + e], g[i + e], m[i + e], v[i + e]);
        }
    }
}

void hr_launch_adam(const HrAdamBatch& b, hipStream_t stream)
{
    if (b.count <= 0 || b.first_block[b.count] <= 0) return;
    hipLaunchKernelGGL(hr_adam_kernel, dim3((unsigned)b.first_block[b.count]), dim3(256), 0, stream, b);
}
