"""CPU check of the formulas the HIP kernels call (hyperreel_amd/csrc/hr_math.h, compiled
for the host by g++ into a test-only library) against the oracle, on the golden scenes.
This exercises the kernels' arithmetic and the YAML -> hr_config compiler without a GPU;
the cross-lane parts, indexing and MFMA layouts are covered by the `-m gpu` tests."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import Golden, build_host_lib, golden_cases, linf, sweep_cases
from hyperreel_amd import plan
from hyperreel_oracle import HyperReelOracle, eval_sh_bases_deg2, grid_sample_2d

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'host_math', 'hr_math_host.cpp')
OUT = os.path.join(HERE, 'host_math', '_build', 'libhr_math_host.so')

FP = C.POINTER(C.c_float)
IP = C.POINTER(C.c_int)


def fp(a):
    return a.ctypes.data_as(FP)


@pytest.fixture(scope='module')
def hm():
    deps = [SRC, os.path.join(HERE, '..', 'hyperreel_amd', 'csrc', 'hr_math.h'),
            os.path.join(HERE, '..', 'include', 'hyperreel_hip.h')]
    build_host_lib(OUT, SRC, deps)
    lib = C.CDLL(OUT)
    lib.hm_normalize_time.restype = C.c_float
    lib.hm_normalize_time.argtypes = [C.c_void_p, C.c_float]
    return lib


def test_config_struct_layout_matches_c(hm):
    assert hm.hm_sizeof_config() == C.sizeof(plan.hr_config)


SMALL = [c for c in golden_cases() if c.endswith('_small') or c.startswith('config1')]


@pytest.mark.parametrize('case', SMALL + sweep_cases())
def test_features_and_embedding_match_oracle(hm, case):
    g = Golden(case)
    if plan.is_cascade(g.cfg):
        pytest.skip('two-level models are exercised end to end on the GPU')
    hc = plan.compile_config(g.cfg, g.dataset, g.grid, iteration=g.iteration)
    orc = HyperReelOracle(g.cfg, g.dataset, g.state_dict, iteration=g.iteration)
    rays = np.ascontiguousarray(g.rays[:300], np.float32)
    n = rays.shape[0]
    assert rays.shape[1] == hc.ray_dim
    # MLP input features
    feat = np.zeros((n, hc.mlp_in), np.float32)
    hm.hm_features(C.byref(hc), fp(rays), n, fp(feat))
    ref = orc._param_pe(rays)
    assert ref.shape == feat.shape
    assert float(np.max(np.abs(feat - ref) / (1 + np.abs(ref)))) <= 2e-6
    # embedding from the oracle's own raw head: isolates everything after the MLP
    x = orc.embed(rays)
    head = np.ascontiguousarray(x['_head_raw'], np.float32)
    Z = hc.z_channels
    assert head.shape == (n, Z * hc.preds_per_z)
    dist = np.zeros((n, Z), np.float32)
    pts = np.zeros((n, Z, 3), np.float32)
    valid = np.zeros((n, Z), np.int32)
    bt = np.zeros((n,), np.float32)
    hm.hm_embed(C.byref(hc), fp(rays), fp(head), n, fp(dist), fp(pts), valid.ctypes.data_as(IP), fp(bt))
    d_ref = x['distances'].reshape(n, Z)
    assert float(np.max(np.abs(dist - d_ref) / (1 + np.abs(d_ref)))) <= 1e-6
    # DoNeRFContract.contract_points is 0 / 0 for a point at the centre (a masked sample at distance 0): NaN in the reference, here and there
    both_nan = np.isnan(pts) & np.isnan(x['points'])
    assert float(np.max(np.where(both_nan, 0.0, np.abs(pts - x['points']) / (1 + np.abs(x['points']))))) <= 1e-6
    if 'base_times' in x:
        assert linf(bt, x['base_times'][:, 0, 0]) == 0.0
    col = orc.color(x)
    assert (valid.astype(bool) == col['valid']).mean() > 0.999


@pytest.mark.parametrize('case', SMALL)
def test_per_ray_quadratic_terms_give_the_same_distances_bit_for_bit(hm, case):
    """The sample kernel computes o.o, d.d, o.d of the sphere / cylinder intersection once per ray (csrc/sample_core.inc: hr_ray_constants)
    and hands them to hr_sample_distance, which otherwise forms them per sample (the reference's order: primitive.py:425-431,
    intersect_utils.py:45-125): the same operations either way."""
    g = Golden(case)
    if plan.is_cascade(g.cfg):
        pytest.skip('two-level models are exercised end to end on the GPU')
    hc = plan.compile_config(g.cfg, g.dataset, g.grid, iteration=g.iteration)
    orc = HyperReelOracle(g.cfg, g.dataset, g.state_dict, iteration=g.iteration)
    rays = np.ascontiguousarray(g.rays[:300], np.float32)
    n = rays.shape[0]
    head = np.ascontiguousarray(orc.embed(rays)['_head_raw'], np.float32)
    Z = hc.z_channels
    a = np.zeros((n, Z), np.float32)
    b = np.zeros((n, Z), np.float32)
    hm.hm_distance_both(C.byref(hc), fp(rays), fp(head), n, fp(a), fp(b))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_taps_sh_density_normalize(hm):
    rng = np.random.default_rng(0)
    g = np.concatenate([rng.uniform(-1.2, 1.2, 500), [-1.0, 1.0, 0.0, 0.99999994, -0.99999994]]).astype(np.float32)
    for size in (1, 2, 7, 600):
        i0 = np.zeros(g.shape, np.int32); i1 = np.zeros(g.shape, np.int32)
        w0 = np.zeros(g.shape, np.float32); w1 = np.zeros(g.shape, np.float32)
        hm.hm_taps(fp(g), g.size, size, i0.ctypes.data_as(IP), i1.ctypes.data_as(IP), fp(w0), fp(w1))
        # a 1-channel "image" whose texel value is its index: the line sample must equal grid_sample
        line = np.arange(size, dtype=np.float32).reshape(1, size, 1) + 1.0
        ref = grid_sample_2d(line, np.zeros_like(g), g)[0]
        got = w0 * (i0 + 1.0) + w1 * (i1 + 1.0)
        assert np.max(np.abs(got - ref)) <= 1e-4 * size
    d = rng.standard_normal((64, 3)).astype(np.float32)
    sh = np.zeros((64, 9), np.float32)
    hm.hm_sh(fp(np.ascontiguousarray(d)), 64, fp(sh))
    assert np.max(np.abs(sh - eval_sh_bases_deg2(d))) <= 1e-6


def test_contraction_properties(hm):
    """Size-independent properties of the mip-NeRF-360 contraction (nlf/contract.py:113-192) as the kernels compute it:
    points land strictly inside radius 2 and keep their direction, the inner ball is only rescaled, and
    inverse_contract_distance undoes contract_distance on the anchors' range."""
    from hyperreel_amd import config as Cfg
    cfg, ds = Cfg.model_config('donerf_sphere'), Cfg.dataset_scalars('donerf')
    hc = plan.compile_config(cfg, ds, [8, 8, 8])
    rng = np.random.default_rng(3)
    d = rng.standard_normal((4000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    r = np.exp(rng.uniform(np.log(1e-3), np.log(1e6), (4000, 1))).astype(np.float32)
    p = np.ascontiguousarray(d * r)
    q = np.zeros_like(p)
    hm.hm_contract_points(C.byref(hc), fp(p), p.shape[0], fp(q))
    qn = np.linalg.norm(q, axis=-1)
    # radius r1 maps to exactly 2; beyond the end radius the linear-in-disparity shell overshoots, bounded by r -> inf
    contract = plan._MipNerf(cfg.embedding.embeddings.ray_intersect_0.intersect.contract, ds)
    r0, r1 = float(contract.r0), float(contract.r1)
    assert np.all(np.isfinite(q))
    assert np.all(qn[r[:, 0] <= r1] <= 2.0 + 1e-6)
    assert np.all(qn <= 2.0 + r0 / (r1 - r0) + 1e-5)
    cosang = np.sum(q * d, -1) / np.maximum(qn, 1e-30)
    assert np.min(cosang) > 1.0 - 1e-5                                   # direction preserved
    inner = r[:, 0] < hc.c_r0 * 0.999
    assert np.allclose(q[inner], p[inner] / np.float32(hc.c_r0), rtol=1e-6, atol=0)
    order = np.argsort(r[:, 0])
    assert np.all(np.diff(qn[order]) >= -1e-6)                           # monotone in the radius
    # distances: contract (host, plan._MipNerf) then inverse-contract (kernel arithmetic)
    d1 = float(contract.d1)                                              # beyond it inverse_contract clamps (contract.py:147)
    dist = np.exp(rng.uniform(np.log(1e-2), np.log(0.98 * d1), 2000)).astype(np.float32)
    cd = np.asarray([contract.contract_distance(v) for v in dist], np.float32)
    back = np.zeros_like(cd)
    hm.hm_inverse_contract_distance(C.byref(hc), fp(cd), cd.size, fp(back))
    assert np.max(np.abs(back - dist) / dist) < 2e-4                     # the outer shell compresses: ~1/d^2 conditioning
    assert np.all(np.abs(cd) <= 2.0)
    far = np.full(8, 1e4, np.float32)                                    # past the end distance: saturates at d1
    cf = np.asarray([contract.contract_distance(v) for v in far], np.float32)
    hm.hm_inverse_contract_distance(C.byref(hc), fp(cf), cf.size, fp(far))
    assert np.allclose(far, d1, rtol=1e-5)


def test_display_pack_matches_the_viewer_host_code(hm):
    """hr_to8b == utils/__init__.py:47 to8b; hr_display_src_pixel == NeRFGUI.test_step's transpose(1, 0, 2) then
    np.flip(axis=0) (utils/gui_utils.py:199-205).  The device kernel is these two functions per output pixel."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-0.2, 1.2, 20000), np.arange(256) / 255.0, np.nextafter(np.arange(256) / 255.0, 2),
                        [0.0, 1.0, -0.0, 0.5, 1e-9, 0.999999, np.float32(1) - np.float32(2 ** -24)]]).astype(np.float32)
    got = np.zeros(x.shape, np.uint8)
    hm.hm_to8b(fp(x), x.size, got.ctypes.data_as(C.POINTER(C.c_ubyte)))
    assert np.array_equal(got, (255 * np.clip(x, 0, 1)).astype(np.uint8))
    h, w = 5, 7
    img = np.arange(h * w).reshape(h, w)
    for transpose in (0, 1):
        for flip in (0, 1):
            ref = img
            if transpose:
                ref = ref.transpose(1, 0)
            if flip:
                ref = np.flip(ref, axis=0)
            src = np.zeros(h * w, np.int64)
            hm.hm_display_map(h, w, transpose, flip, src.ctypes.data_as(C.POINTER(C.c_longlong)))
            assert np.array_equal(src.reshape(ref.shape), ref)


def test_row_features_equal_the_kernel_encoding(hm):
    """hyperreel_amd.train.row_features (torch ops, because a cascade's rows carry a gradient) == hr_ray_features of the
    point MLP's configuration on the same rows."""
    import torch
    from helpers import trainable_sweep_cases
    from hyperreel_amd.train import row_features
    for case in trainable_sweep_cases(cascades=True):
        g = Golden(case)
        _, fine = plan.compile_model(g.cfg, g.dataset, g.grid)
        rows = np.random.default_rng(0).standard_normal((64, fine.casc_row_dim)).astype(np.float32)
        kc = plan.hr_config.from_buffer_copy(fine)
        kc.ray_dim = fine.casc_row_dim                     # the point MLP's "rays" are the rows (api.hip launch_cascade_front)
        want = np.zeros((64, fine.mlp_in), np.float32)
        hm.hm_features(C.byref(kc), fp(rows), 64, fp(want))
        assert np.abs(row_features(fine, torch.from_numpy(rows)).numpy() - want).max() <= 2e-7, case


def test_bilinear_taps_partition_unity_and_stay_in_range(hm):
    """Property of hr_make_tap (one axis of F.grid_sample, align_corners=True, zeros padding) over random coordinates and
    sizes: indices are valid texels, weights are non-negative, sum to 1 strictly inside the image and to at most 1 at
    and beyond its border, and vanish entirely more than one texel outside."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    @settings(max_examples=60, deadline=None)
    @given(st.integers(2, 1100), st.integers(0, 2 ** 31 - 1))
    def prop(size, seed):
        rng = np.random.default_rng(seed)
        g = np.concatenate([rng.uniform(-1.6, 1.6, 256), [-1.0, 1.0, 0.0, -1.0 - 2.0 / (size - 1), 1.0 + 2.0 / (size - 1)]]).astype(np.float32)
        n = g.size
        i0, i1 = np.zeros(n, np.int32), np.zeros(n, np.int32)
        w0, w1 = np.zeros(n, np.float32), np.zeros(n, np.float32)
        hm.hm_taps(fp(g), n, size, i0.ctypes.data_as(IP), i1.ctypes.data_as(IP), fp(w0), fp(w1))
        assert ((0 <= i0) & (i0 < size) & (0 <= i1) & (i1 < size)).all()
        assert (w0 >= 0).all() and (w1 >= 0).all() and (w0 + w1 <= 1 + 1e-6).all()
        ix = (g.astype(np.float64) + 1) / 2 * (size - 1)
        inside = (ix > 0.01) & (ix < size - 1.01)
        assert np.abs((w0 + w1)[inside] - 1).max(initial=0) <= 1e-5
        far = (ix < -1.01) | (ix > size + 0.01)
        assert not (w0 + w1)[far].any()

    prop()


def test_clamped_taps_give_the_same_lerp_bit_for_bit(hm):
    """hr_make_tap_c (taps addressed as base, base + 1 -- what the class-specialised gather uses so that one offset per
    texel pair is enough) against hr_make_tap: both taps exist, and v[i0]*w0 + v[i1]*w1 is the same float for random
    texel values, including coordinates on and beyond both borders."""
    rng = np.random.default_rng(5)
    for size in (2, 3, 7, 64, 600, 1007):
        g = np.concatenate([rng.uniform(-1.3, 1.3, 4000), [-1.0, 1.0, 0.0, -1.0 - 2.0 / (size - 1), 1.0 + 2.0 / (size - 1),
                                                           np.nextafter(np.float32(1.0), np.float32(2.0)), np.nextafter(np.float32(-1.0), np.float32(-2.0))]]).astype(np.float32)
        n = g.size
        i0, i1, c0 = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        w0, w1, cw0, cw1 = (np.zeros(n, np.float32) for _ in range(4))
        hm.hm_taps(fp(g), n, size, i0.ctypes.data_as(IP), i1.ctypes.data_as(IP), fp(w0), fp(w1))
        hm.hm_taps_c(fp(g), n, size, c0.ctypes.data_as(IP), fp(cw0), fp(cw1))
        assert ((0 <= c0) & (c0 <= size - 2)).all()
        v = rng.standard_normal(size).astype(np.float32)
        ref = (v[i0] * w0) + (v[i1] * w1)              # fp32 products and sum, like the kernels' fma chain up to the zero terms
        got = (v[c0] * cw0) + (v[c0 + 1] * cw1)
        assert np.array_equal(ref, got), size


def test_in_range_taps_equal_the_clamped_taps_where_a_valid_sample_can_be(hm):
    """hr_make_tap_in (three cases instead of the general border logic) against hr_make_tap_c for every coordinate whose
    floor(ix) is -1 .. n-1: inside the aabb, on both faces, and the few ulps beyond them that rounding can produce."""
    rng = np.random.default_rng(6)
    for size in (2, 3, 7, 64, 600, 640, 1007):
        edge = [np.float32(-1.0), np.float32(1.0), np.float32(0.0)]
        for _ in range(6):
            edge += [np.nextafter(edge[-3], np.float32(-4.0)), np.nextafter(edge[-2], np.float32(4.0)), np.float32(0.0)]
        lim = 1.0 + 1.9 / (size - 1)                       # floor(ix) stays within [-1, n-1]
        g = np.concatenate([rng.uniform(-1.0, 1.0, 6000), rng.uniform(-lim, lim, 2000), rng.uniform(0.999, 1.0, 500), rng.uniform(-1.0, -0.999, 500),
                            np.array(edge, np.float64)]).astype(np.float32)
        n = g.size
        a0, b0 = np.zeros(n, np.int32), np.zeros(n, np.int32)
        aw0, aw1, bw0, bw1 = (np.zeros(n, np.float32) for _ in range(4))
        hm.hm_taps_c(fp(g), n, size, a0.ctypes.data_as(IP), fp(aw0), fp(aw1))
        hm.hm_taps_in(fp(g), n, size, b0.ctypes.data_as(IP), fp(bw0), fp(bw1))
        assert np.array_equal(a0, b0) and np.array_equal(aw0.view(np.int32), bw0.view(np.int32)) and np.array_equal(aw1.view(np.int32), bw1.view(np.int32)), size
