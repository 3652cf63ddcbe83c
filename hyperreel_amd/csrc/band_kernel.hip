// Band probe of the verified fast path: how far the cheap MLP arithmetic (f16f8) moves the quantities the per-sample stage
// COMPARES -- primitive lengths (plane depth / radius), distances, points -- measured per model on its calibration rays against the
// reference-grade arithmetic (f16x3), in the normalisation the sample stage's per-sample margins use (hr_math.h, HrRisk).  hr_model_finalize / hr_model_calibrate set the model's "at risk" band from it (api.hip,
// calibrate_band): a comparison further than the band from flipping falls the same way under both arithmetics.
// The reference has no counterpart: its BaseMLP is fp32 (nlf/nets/mlp.py:159-172) and its decisions are exact comparisons
// (nlf/intersect/base.py:194, utils/intersect_utils.py:45-150).
// Not a hot path: two head exports and this kernel over <= 65 536 rays at finalize / calibrate.
#include "hr_kernels.h"
#include "hr_math.h"

// non-negative floats order like their bit patterns
__device__ __forceinline__ void hr_band_max(unsigned* slot, float v)
{
    if (!(v >= 0.0f) || !(v < 1e30f)) return;
    const unsigned b = __float_as_uint(v);
    if (b > *reinterpret_cast<volatile unsigned*>(slot)) atomicMax(slot, b);
}

// cfg: the model's configuration in the USER's head-column order with isect_mask_off = 1 (the probe wants the distance before
// the near/far mask; the mask is applied here, `mask_on`, to decide which samples count).  The margins the sample stage derives per sample
// (hr_math.h, HrRisk) are  band_zc dlen amp  for distances and  band_q max(amp) + band_off  for points; this kernel measures the
// differences in exactly those normalisations, through the same functions.
__global__ __launch_bounds__(256) void hr_band_probe_kernel(const hr_config* __restrict__ cfgp, const HrBandArgs a)
{
    const hr_config& cfg = *cfgp;
    const int Z = cfg.z_channels, P = cfg.preds_per_z;
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_rays * Z) return;
    const int64_t ray = s / Z;
    const int k = (int)(s - ray * Z);
    const float* r = a.rays + ray * cfg.ray_dim;
    const float ro[3] = {r[0] - cfg.isect_origin[0], r[1] - cfg.isect_origin[1], r[2] - cfg.isect_origin[2]};
    const float rd[3] = {r[3], r[4], r[5]};
    const float t_ray = r[cfg.ray_dim - 1];
    const float* hA = a.head_a + s * P;
    const float* hB = a.head_b + s * P;

    if (a.phase == 1 && !a.ray_ok[ray]) return;                  // an ill-conditioned ray (grazing a plane, tangent to a sphere): no statistic is taken from it
    HrRisk ra = HR_RISK_INIT(0.0f, 0.0f, 0.0f, 0.0f), rb = HR_RISK_INIT(0.0f, 0.0f, 0.0f, 0.0f);
    const float dA = hr_sample_distance(cfg, hA, k, ro, rd, &ra);
    const float dB = hr_sample_distance(cfg, hB, k, ro, rd, &rb);
    // samples the mask keeps under BOTH arithmetics: their distances are what the next comparisons see
    const bool live = (!a.mask_on || (dA > cfg.near && dA < cfg.far && dB > cfg.near && dB < cfg.far)) && ra.amp > 0.0f && rb.amp > 0.0f;
    const float amp = fmaxf(ra.amp, rb.amp), dlen = fmaxf(ra.dlen, rb.dlen);
    if (a.phase == 0) {
        if (live && amp > a.amp_cut) a.ray_ok[ray] = 0;
        return;
    }
    // the raw head, column by column (pruned columns are 0 in both exports)
    for (int c = 0; c < P && c < 64; ++c) hr_band_max(a.stats + HR_BAND_HEAD0 + c, fabsf(hA[c] - hB[c]));
    // the length before the inverse contraction: a smooth function of one head column -- every sample of the ray counts, masked or not
    hr_band_max(a.stats + HR_BAND_ZC, fabsf(ra.zc - rb.zc));
    if (!live) return;
    atomicAdd(a.stats + HR_BAND_COUNTED, 1u);
    // (band 0: `hit` = a radius or near root of exactly 0, a discriminant of exactly 0 -- not a continuous point of the distance)
    if (ra.hit || rb.hit) { atomicAdd(a.stats + HR_BAND_SHAKY, 1u); return; }
    const float dn = fabsf(dA - dB) / (dlen * amp);
    // a normalised difference this large is a decision that fell the other way (root choice), not arithmetic error
    if (!(dn <= a.flip_cut)) { atomicAdd(a.stats + HR_BAND_FLIPPED, 1u); return; }
    hr_band_max(a.stats + HR_BAND_DIST_N, dn);
    hr_band_max(a.stats + HR_BAND_DIST, fabsf(dA - dB));

    float oc[3] = {0.f, 0.f, 0.f};
    if (cfg.contract_type != HR_CONTRACT_IDENTITY) hr_contract_point(cfg, ro[0], ro[1], ro[2], oc);
    float time_off = 0.0f;
    if (cfg.advect) time_off = t_ray - hr_base_time(cfg, t_ray);
    float pAA[3], pAB[3], pBB[3], cd;
    hr_sample_point(cfg, hA, dA, ro, rd, oc, time_off, pAA, &cd);
    hr_sample_point(cfg, hB, dA, ro, rd, oc, time_off, pAB, &cd);
    hr_sample_point(cfg, hB, dB, ro, rd, oc, time_off, pBB, &cd);
    const float geo = fmaxf(fabsf(pAB[0] - pBB[0]), fmaxf(fabsf(pAB[1] - pBB[1]), fabsf(pAB[2] - pBB[2])));
    const float off = fmaxf(fabsf(pAA[0] - pAB[0]), fmaxf(fabsf(pAA[1] - pAB[1]), fabsf(pAA[2] - pAB[2])));
    hr_band_max(a.stats + HR_BAND_GEO_N, geo / amp);
    hr_band_max(a.stats + HR_BAND_OFF, off);
}

void hr_launch_band_probe(const HrBandArgs& a, int z_channels, hipStream_t stream)
{
    const int64_t n = a.n_rays * z_channels;
    if (n <= 0) return;
    hipLaunchKernelGGL(hr_band_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a.cfg_dev, a);
}
