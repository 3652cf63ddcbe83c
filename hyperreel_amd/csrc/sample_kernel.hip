// Stand-alone sample kernel: the sample stage (sample_core.inc) over a head that the MLP kernel left in the HBM
// workspace.  Used where the fused frame kernel (fused_impl.inc) does not apply: diagnostics (hr_render_fields),
// point_prediction cascades, heads too wide for the LDS hand-over, the exact-fp32 MLP.
#define HR_GATHER_FENCED 1
#include "sample_core.inc"

template <int ZP, bool HALF, int PC, int NB>
__global__ __launch_bounds__(256, (HrGatherTune<ZP, HALF>::MIN_BLOCKS)) void hr_sample_kernel(const hr_config* __restrict__ cfgp, const HrSampleArgs a)
{
    // the configuration lives in device memory (2 KB: too large to index dynamically as a by-value kernel argument
    // without the compiler copying it to scratch); uniform reads of it become scalar loads
    const hr_config& cfg = *cfgp;
    constexpr int RPB = 256 / ZP;   // rays per block
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int CA = a.ca_total;                   // padded appearance slots (multiple of 4)
    const int HS = a.nq * 4 + 4;                   // LDS row stride of a head row (+4: conflict-free float4 fills)
    const int RPR = a.rows_per_ray;                // head rows per ray (1 unless the head comes from a point MLP)
    float* s_head = lds;                           // [RPB * RPR][HS]
    float* s_M = lds + RPB * RPR * HS;             // [RPB][3][CA]
    float* s_x = s_M + RPB * 3 * CA;               // [256] cross-wave scratch, ZP > 64 only

    const int tid = threadIdx.x;
    const int rib = tid / ZP;
    const int k = tid % ZP;
    // XCD-aware block order: the dispatcher places block b on XCD b % 8, so consecutive
    // blocks (neighbouring rays, overlapping texel footprints) would land on 8 different L2s.
    // Give each XCD a contiguous range of the ray list instead (bijective for any grid size).
    unsigned bid = blockIdx.x;
    {
        const unsigned nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int64_t ray_base = (int64_t)bid * RPB;
    const int64_t lrow = ray_base + rib;           // position in this launch = row of the head
    // second pass of the verified fast path: the launch is sized for the list's capacity, *n_rays_dev rays are there.  (Workgroups that walk
    // the list from a fixed grid would save the ~14 us of dispatching empty workgroups -- and cost every launch of this kernel 44 registers and
    // 52 bytes of scratch: the loop makes the compiler hoist the body's invariants.)
    int64_t n_rays = a.n_rays;
    if (a.zero_word && blockIdx.x == 0 && threadIdx.x == 0) *a.zero_word = 0u;
    if (a.n_rays_dev) {
        const int64_t nd_raw = (int64_t)*a.n_rays_dev;
        const int64_t nd = nd_raw > a.list_off ? nd_raw - a.list_off : 0;
        n_rays = nd < n_rays ? nd : n_rays;
    }
    if (ray_base >= n_rays) return;                // (block-uniform, before any barrier)
    const bool ray_ok = lrow < n_rays;
    const int64_t ray = (a.ray_index && ray_ok) ? (int64_t)a.ray_index[lrow] : lrow;     // the caller's ray

    // ---- stage this block's head into LDS: per feature quad the block's RPB rays are RPB x 16
    //      contiguous bytes in the HQ layout (RPB divides 64, so a block never straddles a 64-ray group)
    if (RPR == 1) {
        const float4* src4 = reinterpret_cast<const float4*>(a.head) + ((size_t)(ray_base >> 6) * a.nq << 6) + (ray_base & 63);
        const int total = a.nq * RPB;
        for (int i = tid; i < total; i += 256) {
            const int q = i / RPB, r = i - q * RPB;
            *reinterpret_cast<float4*>(s_head + r * HS + 4 * q) = src4[((size_t)q << 6) + r];
        }
    } else {                                       // cascade: RPB * RPR rows of the point MLP's head, any alignment
        const float4* src4 = reinterpret_cast<const float4*>(a.head);
        const int NR = RPB * RPR;
        const int64_t row0 = ray_base * RPR, n_rows = n_rays * RPR;
        for (int i = tid; i < a.nq * NR; i += 256) {
            const int q = i / NR, r = i - q * NR;
            const int64_t row = row0 + r;
            if (row < n_rows) *reinterpret_cast<float4*>(s_head + r * HS + 4 * q) = src4[hr_head_index(row, 4 * q, a.nq) >> 2];
        }
    }

    // ---- per-ray quantities (computed redundantly by the ray's lanes) and the ray's decode matrix
    // RGB shading: the decode matrix is basis_mat itself, the same for every ray -- the block keeps ONE copy, filled by its
    // first ray's lanes (SH: one per ray, folded with that ray's view direction)
    // The ray record (origin, direction, time, contracted origin, keyframe time, the quadratic's ray terms: sample_core.inc, HrRayLane) is
    // computed by ONE lane per ray -- the first RPB lanes of the workgroup, a ray each -- and read back from LDS by the ray's lanes after the
    // barrier: in the lane-per-sample mapping every lane of the ray would otherwise repeat it.
    __shared__ __attribute__((aligned(16))) float s_ray[RPB * HR_RAY_RECORD];
    if (tid < RPB) {
        const int64_t lrow_r = ray_base + tid;
        const bool ok_r = lrow_r < n_rays;
        const int64_t ray_r = (a.ray_index && ok_r) ? (int64_t)a.ray_index[lrow_r] : lrow_r;
        HrRayLane R = hr_load_ray(cfg, a, ray_r, ok_r);
        hr_ray_constants(cfg, R);
        hr_store_ray_record(R, s_ray + tid * HR_RAY_RECORD);
    }
    const bool per_ray_M = (cfg.shading == HR_SHADING_SH);
    float* M = s_M + (per_ray_M ? rib * 3 * CA : 0);
#ifdef HR_TUNING
    if (!(a.dbg_mode & 4))
#endif
    if (per_ray_M) {                               // SH: folded with the ray's view direction, which its lanes read themselves (the record is not published yet)
        HrRayLane V = hr_load_ray(cfg, a, 0, false);
        if (ray_ok) {
            const float* r = a.rays + ray * cfg.ray_dim;
            V.vd[0] = r[3]; V.vd[1] = r[4]; V.vd[2] = r[5];
        }
        hr_fill_decode<ZP>(cfg, a, V, k, M);
    } else if (rib == 0) {
        hr_fill_decode<ZP>(cfg, a, hr_load_ray(cfg, a, 0, false), k, M);
    }
    __shared__ __attribute__((aligned(16))) float s_ones[HR_GATHER_ONES];
    hr_gather_ones_init(s_ones);
    __syncthreads();
    const HrRayLane L = hr_read_ray_record(s_ray + rib * HR_RAY_RECORD);

#ifdef HR_TUNING
    unsigned long long sph__[12] = {};
#endif
    hr_sample_body<ZP, HALF, 1, NB, PC>(cfg, a, L, ray, ray_ok, k, s_head + rib * RPR * HS, HS, M, s_ones, s_x HR_SPH_ARG);
}

static size_t hr_sample_lds_bytes(int nq, int ca_total, int ZP, int rows_per_ray)
{
    const int RPB = 256 / ZP;
    size_t bytes = ((size_t)RPB * rows_per_ray * (nq * 4 + 4) + (size_t)RPB * 3 * ca_total + (ZP > 64 ? 256 : 0)) * sizeof(float);
#ifdef HR_SAMPLE_LDS_FLOOR     // measurement builds: fewer workgroups per CU than the registers allow (profiles/r06_k2_occupancy_ab.txt)
    if (bytes < HR_SAMPLE_LDS_FLOOR) bytes = HR_SAMPLE_LDS_FLOOR;
#endif
    return bytes;
}

void hr_launch_samples(const hr_config& cfg, const HrSampleArgs& args, hipStream_t stream)
{
    if (args.n_rays <= 0) return;
    const int Z = cfg.z_channels;
    int ZP = 8;
    while (ZP < Z) ZP <<= 1;
    const int RPB = 256 / ZP;
    const unsigned blocks = (unsigned)((args.n_rays + RPB - 1) / RPB);
    const size_t lds = hr_sample_lds_bytes(args.nq, args.ca_total, ZP, args.rows_per_ray);
    HrSampleArgs args2 = args;
#ifdef HR_TUNING       // measurement builds only (tools/): HR_SAMPLE_DBG=1 skips the feature gather
    static const int dbg = [] { const char* e = getenv("HR_SAMPLE_DBG"); return e ? atoi(e) : 0; }();
    args2.dbg_mode = dbg;
#endif
    // few samples x many head columns can exceed the 64 KiB a kernel gets by default (e.g. 32 rays x 8 x 64 floats)
    const bool big_lds = lds > 64 * 1024;
    // the shipped [8, 4, 4] / [8, 0, 0] decompositions get the class-specialised gather of their texel format (sample_core.inc); ZP >= 8
    // keeps a quad inside one ray, video nets additionally need two keyframes
    const int pclass = (args.rows_out == nullptr && (!cfg.video || cfg.num_keyframes >= 2)) ? hr_plane_class(args.planes, 0, args.ca_total) : 0;
    // every second factor a line (static nets; a keyframe net inside hr_render_frame): the gather compiled for two line taps
    bool all_lines = pclass != 0;
    for (int j = 0; j < 3; ++j)
        if (args.planes[j].cd4 + args.planes[j].ca4 > 0 && args.planes[j].bw != 1) all_lines = false;
#define HR_LAUNCH_SAMPLES_N(Z_, H_, P_, N_) \
    do { \
        if (big_lds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hr_sample_kernel<Z_, H_, P_, N_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((hr_sample_kernel<Z_, H_, P_, N_>), dim3(blocks), dim3(256), lds, stream, args2.cfg_dev, args2); \
    } while (0)
#define HR_LAUNCH_SAMPLES_T(Z_, H_, P_) \
    do { \
        if (P_ != 0 && all_lines) HR_LAUNCH_SAMPLES_N(Z_, H_, P_, (P_ != 0 ? 2 : 4)); else HR_LAUNCH_SAMPLES_N(Z_, H_, P_, 4); \
    } while (0)
#define HR_LAUNCH_SAMPLES(Z_) \
    do { \
        if (cfg.grid_dtype == HR_GRID_FP16) { \
            if (pclass == 1) HR_LAUNCH_SAMPLES_T(Z_, true, 1); else if (pclass == 2) HR_LAUNCH_SAMPLES_T(Z_, true, 2); else HR_LAUNCH_SAMPLES_T(Z_, true, 0); \
        } else { \
            if (pclass == 1) HR_LAUNCH_SAMPLES_T(Z_, false, 1); else if (pclass == 2) HR_LAUNCH_SAMPLES_T(Z_, false, 2); else HR_LAUNCH_SAMPLES_T(Z_, false, 0); \
        } \
    } while (0)
    switch (ZP) {
        case 8: HR_LAUNCH_SAMPLES(8); break;
        case 16: HR_LAUNCH_SAMPLES(16); break;
        case 32: HR_LAUNCH_SAMPLES(32); break;
        case 64: HR_LAUNCH_SAMPLES(64); break;
        case 128: HR_LAUNCH_SAMPLES(128); break;
        case 256: HR_LAUNCH_SAMPLES(256); break;
        default: break;  // Z > 256 is rejected by hr_model_create
    }
#undef HR_LAUNCH_SAMPLES
#undef HR_LAUNCH_SAMPLES_T
#undef HR_LAUNCH_SAMPLES_N
}
