// The consumer of the co-resident pair ("duo" plan, hr_kernels.h): the sample stage (sample_core.inc) as a persistent kernel that
// runs BESIDE the persistent MLP kernel (fused_impl.inc, HR_DUO_KERNEL) on the same CUs -- two kernels, two register allocations.
#include "sample_core.inc"

// ---------------------------------------------------------------------------------------------------------------------
// The consumer of the co-resident pair (hr_kernels.h, "duo" plan): the stand-alone sample kernel with one difference -- a block
// first waits for the flag of the 64-ray tile its rays belong to, then stages their head rows with sc1 loads.  An ordinary grid:
// the hardware's in-order workgroup dispatch IS the ticket queue.  Block b serves queue x = b % X (the XCD the dispatcher places
// it on: speed only) and takes that queue's j-th piece, j = b / X -- the tiles XCD x's producer workgroups make, in the order
// they make them.  It waits for nothing but a producer flag and gives up (status bit 1) after HR_DUO_TIMEOUT_TICKS.
template <int ZP, bool HALF, int PC, int NB, int MINW>
__global__ __launch_bounds__(256, MINW) void hr_sampleq_kernel(const hr_config* __restrict__ cfgp, const HrSampleArgs a, const HrDuoArgs q)
{
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const hr_config& cfg = *cfgp;
    constexpr int RPB = 256 / ZP;                  // rays per block
    constexpr int BPT = 64 / RPB;                  // blocks per tile
    static_assert(ZP <= 64, "a ray stays inside one wavefront");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int CA = a.ca_total;
    const int HS = a.nq * 4 + 4;
    float* s_head = lds;                           // [RPB][HS]
    float* s_M = lds + RPB * HS;                   // [RPB or 1][3][CA]
    int* s_ok = reinterpret_cast<int*>(s_M + RPB * 3 * CA);
    const int tid = threadIdx.x;
    const int rib = tid / ZP;
    const int k = tid % ZP;
    const int X = q.n_queues;
    const int x = (int)(blockIdx.x % (unsigned)X), j = (int)(blockIdx.x / (unsigned)X);
    const int lo = hr_duo_tile_lo(q.n_tiles, X, x);
    if (j >= (hr_duo_tile_lo(q.n_tiles, X, x + 1) - lo) * BPT) return;          // (queues differ by at most one tile)
    const int tile = lo + j / BPT, r0 = (j % BPT) * RPB;
    unsigned long long* times = reinterpret_cast<unsigned long long*>(q.ctl + 272);
    if (tid == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        if (blockIdx.x == 0) times[2] = ~t0;                                      // measurement (hr_debug_duo_times)
        const unsigned* flag = q.flags + tile;
        // "the pair is over" word of this queue (own 128-byte line; one word read by all 80 000 blocks of a frame serialises at
        // ~40 ns per access and costs milliseconds): looked at only by blocks that have already waited for a while
        unsigned* over = q.ctl + 256 + 32 * x;
        int ok = 1;
        unsigned nap = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            // one lane polls one word, relaxed, and sleeps longer the longer it waits (MI355X_MICROARCH.md "polling-cost")
            ++nap;
            if (nap <= 4) { __builtin_amdgcn_s_sleep(8); continue; }
            __builtin_amdgcn_s_sleep(64);
            if ((nap & 15u) != 0u) continue;
            if (__builtin_amdgcn_s_memrealtime() - t0 > HR_DUO_TIMEOUT_TICKS || __hip_atomic_load(over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                atomicOr(q.status, 2u);
                for (int i = 0; i < 8; ++i) __hip_atomic_store(q.ctl + 256 + 32 * i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *s_ok = ok;
    }
    // meanwhile: the rays and the decode matrix (they do not depend on the head)
    const int64_t ray = (int64_t)tile * 64 + r0 + rib;
    const bool ray_ok = ray < a.n_rays;
    const HrRayLane L = hr_load_ray(cfg, a, ray, ray_ok);
    const bool per_ray_M = (cfg.shading == HR_SHADING_SH);
    float* M = s_M + (per_ray_M ? rib * 3 * CA : 0);
    if (per_ray_M || rib == 0) hr_fill_decode<ZP>(cfg, a, L, k, M);
    __syncthreads();
    if (*s_ok == 0) return;
    // ---- stage the head rows of rays r0 .. r0 + RPB - 1 of the tile: per feature quad RPB x 16 contiguous bytes (HQ layout).
    //      sc1 loads: served past this CU's L1, which may still hold the previous frame's lines of the same workspace
    {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.head)) + (size_t)tile * ((size_t)a.nq * 64 * 16), 0,
                                              a.nq * 64 * 16, 0x00020000);
        const int total = a.nq * RPB;
        for (int i = tid; i < total; i += 256) {
            const int qd = i / RPB, r = i - qd * RPB;
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((qd << 6) + r0 + r) * 16, 0, 16);
            *reinterpret_cast<u4*>(s_head + r * HS + 4 * qd) = v;
        }
    }
    __syncthreads();
#ifdef HR_TUNING
    unsigned long long sph__[12] = {};
#endif
    hr_sample_body<ZP, HALF, 1, NB, PC>(cfg, a, L, ray, ray_ok, k, s_head + rib * HS, HS, M, nullptr HR_SPH_ARG);
    if (tid == 0 && (j & 63) == 63) atomicMax(times + 3, __builtin_amdgcn_s_memrealtime());      // measurement; (one word: not from every block)
}

// The gate in front of the consumer grid: one wavefront that ends once every producer workgroup is resident.  Without it the small
// consumer blocks of a concurrently dispatched grid fill some CUs before those CUs' (large) producer workgroup has been placed, and that
// producer then starves behind an endless supply of small blocks (measured: 4.2 instead of 1.8 ms for the producer).
__global__ void hr_duo_gate_kernel(const HrDuoArgs q)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(q.ctl + 264, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)q.producers) {
        __builtin_amdgcn_s_sleep(16);
        if (__builtin_amdgcn_s_memrealtime() - t0 > HR_DUO_TIMEOUT_TICKS) {
            atomicOr(q.status, 2u);
            for (int i = 0; i < 8; ++i) __hip_atomic_store(q.ctl + 256 + 32 * i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

// sample workgroups per CU beside one producer workgroup: its VGPRs (200 with the four-slot weight ring, 168 with three) leave
// 312 / 344 of the SIMD's 512 per lane -- three or four wavefronts of <= 80
bool hr_launch_duo_consumer(const hr_config& cfg, const HrSampleArgs& args, const HrDuoArgs& q, int consumers_per_cu, int n_cus, bool probe, hipStream_t stream)
{
    const int Z = cfg.z_channels;
    int ZP = 8;
    while (ZP < Z) ZP <<= 1;
    if (ZP > 64 || ZP < 16) return false;
    if (args.rows_per_ray != 1 || args.rows_out) return false;                       // point_prediction cascades keep the chunked plans
    const bool half = (cfg.grid_dtype == HR_GRID_FP16);
    const int pclass = (!cfg.video || cfg.num_keyframes >= 2) ? hr_plane_class(args.planes, 0, args.ca_total) : 0;
    if (pclass == 0) return false;                                                   // the class-specialised gathers only
    bool all_lines = true;
    for (int j = 0; j < 3; ++j)
        if (args.planes[j].cd4 + args.planes[j].ca4 > 0 && args.planes[j].bw != 1) all_lines = false;
    const int RPB = 256 / ZP;
    size_t lds = ((size_t)RPB * (args.nq * 4 + 4) + (size_t)RPB * 3 * args.ca_total) * sizeof(float) + 16;
    if (lds > 64 * 1024) return false;
    if (probe || args.n_rays <= 0) return true;
    (void)n_cus;
    // measurement knob: an LDS request that lets exactly `consumers_per_cu` blocks share the 79 KB a producer leaves (0: as many as
    // the register file admits)
    if (consumers_per_cu > 0) {
        const size_t want = ((size_t)79 * 1024 / consumers_per_cu) & ~(size_t)255;
        if (want > lds && want <= 64 * 1024) lds = want;
    }
    const int X = q.n_queues;
    const int per_queue = ((q.n_tiles + X - 1) / X) * (64 / RPB);                    // blocks of the longest queue
    const unsigned grid = (unsigned)per_queue * (unsigned)X;
    HrSampleArgs args2 = args;
#ifdef HR_TUNING       // measurement builds only (tools/): HR_SAMPLE_DBG bit flags switch phases of the sample stage off
    static const int dbg = [] { const char* e = getenv("HR_SAMPLE_DBG"); return e ? atoi(e) : 0; }();
    args2.dbg_mode = dbg;
#endif
#define HR_Q_LAUNCH(Z_, H_, P_, N_, W_) \
    hipLaunchKernelGGL((hr_sampleq_kernel<Z_, H_, P_, N_, W_>), dim3(grid), dim3(256), lds, stream, args2.cfg_dev, args2, q)
#define HR_Q_W(Z_, H_, P_, N_) HR_Q_LAUNCH(Z_, H_, P_, N_, 6)
#define HR_Q_N(Z_, H_, P_) do { if (all_lines) HR_Q_W(Z_, H_, P_, 2); else HR_Q_W(Z_, H_, P_, 4); } while (0)
#define HR_Q_P(Z_, H_) do { if (pclass == 1) HR_Q_N(Z_, H_, 1); else HR_Q_N(Z_, H_, 2); } while (0)
#define HR_Q_Z(Z_) do { if (half) HR_Q_P(Z_, true); else HR_Q_P(Z_, false); } while (0)
    hipLaunchKernelGGL(hr_duo_gate_kernel, dim3(1), dim3(64), 0, stream, q);
#ifdef HR_Q_PROBE      // register-allocation probes (tools/kres.sh -DHR_Q_PROBE): one instantiation
    (void)half; (void)all_lines;
    HR_Q_LAUNCH(32, false, 1, 2, 6);
#else
    switch (ZP) {
        case 16: HR_Q_Z(16); break;
        case 32: HR_Q_Z(32); break;
        case 64: HR_Q_Z(64); break;
        default: return false;
    }
#endif
#undef HR_Q_Z
#undef HR_Q_P
#undef HR_Q_N
#undef HR_Q_W
#undef HR_Q_LAUNCH
    return true;
}
