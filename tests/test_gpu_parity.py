"""Parity of the HIP path (through the C ABI) against the golden fixtures made from the
reference itself, and against the CPU oracle on the same seeded inputs.  `-m gpu` only.

Tolerance: BASELINE.json north_star -- 1e-4 L-inf on RGB.  Intermediates use relative
tolerances that localise a failure (distances, points, compositing weights)."""
import numpy as np
import pytest
import torch

from helpers import Golden, golden_cases, initialiser_golden_cases, linf, sweep_cases
from hyperreel_amd import config as C
from hyperreel_amd import scenes

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4


def _oracle(g):
    from hyperreel_oracle import HyperReelOracle
    return HyperReelOracle(g.cfg, g.dataset, g.state_dict, iteration=g.iteration)


@pytest.fixture(scope='module')
def fns():
    cache = {}

    def get(case, precision='auto'):
        if (case, precision) not in cache:
            from gpu_common import make_render_fn
            g = Golden(case)
            cache[(case, precision)] = (g, make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=precision, iteration=g.iteration))
        return cache[(case, precision)]

    yield get
    cache.clear()
    torch.cuda.empty_cache()


# the shipped default ('auto': f16f8 first pass + verification where it applies), the bf16 split (fp32 exponent range: the overflow
# fallback), the fp16 split (reference grade: the second pass's arithmetic), the exact fp32 MFMA path
PRECISIONS = ['auto', 'bf16x3', 'f16x3', 'fp32']


def _matrix_cases():
    # bf16x3 carries 2^-17 per product: on MLPs whose head is 15 - 128 x the initialiser's (scenes.MLP_VARIANTS) that is beyond the 1e-4 bar by
    # itself, and 'auto' never picks it there (their activations stay inside the half range) -- those fixtures are for the other three
    out = []
    for c in golden_cases():
        for p in PRECISIONS:
            if p == 'bf16x3' and c.endswith(('_hostile', '_stiff')):
                continue
            out.append((c, p))
    return out


@pytest.mark.parametrize('case,precision', _matrix_cases())
def test_rgb_matches_reference_golden(fns, case, precision):
    from gpu_common import render_np
    g, fn = fns(case, precision)
    out = render_np(fn, g.rays)
    assert out['rgb'].shape == g.rgb.shape
    assert np.isfinite(out['rgb']).all()
    err = np.abs(out['rgb'] - g.rgb).max(-1)
    assert err.max() <= RGB_TOL, f'{case}: L-inf {err.max():.3e} at ray {int(err.argmax())} ({(err > RGB_TOL).sum()} rays over)'


@pytest.mark.parametrize('case', sweep_cases())
def test_every_accepted_shipped_yaml_matches_the_reference(case):
    """One fixture per shipped conf/experiment/model/*.yaml the plan compiler accepts: the
    reference built from that very YAML vs the HIP path built from the same parsed group."""
    from gpu_common import make_render_fn, render_np
    g = Golden(case)
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict, iteration=g.iteration)
    out = render_np(fn, g.rays)
    err = np.abs(out['rgb'] - g.rgb).max(-1)
    assert np.isfinite(out['rgb']).all()
    assert err.max() <= RGB_TOL, f'{case}: L-inf {err.max():.3e} at ray {int(err.argmax())}'


@pytest.mark.parametrize('precision', PRECISIONS)
@pytest.mark.parametrize('case', [c for c in golden_cases() if c.endswith('_small')])
def test_intermediates_match_reference_golden(fns, case, precision):
    from gpu_common import render_np
    g, fn = fns(case, precision)
    out = render_np(fn, g.rays, want=('distances', 'points', 'render_weights'))
    d_ref, p_ref, w_ref = g.arrays['distances'], g.arrays['points'], g.arrays['render_weights']
    rel = lambda a, b: float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))
    assert rel(out['distances'], d_ref) <= 2e-5, f'distances {rel(out["distances"], d_ref):.3e}'
    assert rel(out['points'], p_ref) <= 2e-5, f'points {rel(out["points"], p_ref):.3e}'
    assert linf(out['render_weights'], w_ref) <= 5e-5, f'weights {linf(out["render_weights"], w_ref):.3e}'


@pytest.mark.parametrize('precision', PRECISIONS)
@pytest.mark.parametrize('case', ['donerf_sphere_small', 'neural_3d_z_plane_small', 'technicolor_z_plane_small'])
def test_mlp_head_matches_oracle(fns, case, precision):
    """K1 alone: raw head of the MFMA MLP against the oracle's numpy fp32 matmul chain.
    fp32 MFMA differs by summation order only; the 3-product bf16 split adds ~2^-17 per product."""
    from gpu_common import render_np
    g, fn = fns(case, precision)
    orc = _oracle(g)
    out = render_np(fn, g.rays, want=('head',))
    ref = orc.embed(g.rays)['_head_raw']
    # columns the path never reads are not computed (reported as 0): compare the live ones
    from hyperreel_amd import plan
    hc = fn.model._hc
    live = np.asarray(plan.live_head_columns(hc))
    assert live.sum() == {'donerf_sphere_small': 11, 'neural_3d_z_plane_small': 15, 'technicolor_z_plane_small': 15}[case]
    mask = np.tile(live, hc.z_channels)
    assert np.all(out['head'][:, ~mask] == 0.0)
    scale = np.abs(ref).max()
    err = np.max(np.abs(out['head'][:, mask] - ref[:, mask])) / scale
    assert err <= (3e-5 if precision == 'bf16x3' else 5e-6), f'{err:.3e}'


@pytest.mark.parametrize('precision', PRECISIONS)
@pytest.mark.parametrize('model', C.MODEL_NAMES)
def test_seeded_scene_matches_oracle(model, precision):
    """Fresh seeded scene per model family, odd ray counts, oracle as the checker."""
    from gpu_common import make_render_fn, render_np
    from hyperreel_oracle import HyperReelOracle
    cfg, ds = C.model_config(model), C.dataset_scalars(model)
    grid = [37, 41, 29]
    sd = scenes.make_state_dict(cfg, ds, grid, seed=123, density='dense', app_scale=1.0)
    video = not model.startswith('donerf')
    if 'z_plane' in model:
        rays = scenes.random_rays(777, 5, video, pos_mean=(0, 0, 1.0), pos_std=0.15, dir_mean=(0, 0, -1.2), dir_std=0.5)
    else:
        rays = scenes.random_rays(777, 5, video)
    fn = make_render_fn(cfg, ds, sd, mlp_precision=precision)
    out = render_np(fn, rays, want=('distances', 'render_weights', 'sigma'))
    ref = HyperReelOracle(cfg, ds, sd).render(rays, keep='all')
    assert linf(out['rgb'], ref['rgb']) <= RGB_TOL
    Z = ref['distances'].shape[1]
    assert float(np.max(np.abs(out['distances'] - ref['distances'].reshape(-1, Z)) / (1 + np.abs(ref['distances'].reshape(-1, Z))))) <= 2e-5
    assert linf(out['render_weights'], ref['render_weights']) <= 5e-5


def test_default_initialiser_and_z16(fns):
    """BASELINE config 1: Z=16, 4096 random rays, 64^3 grid, the reference's own (tiny) density init."""
    from gpu_common import render_np
    g, fn = fns('config1_random_z16')
    out = render_np(fn, g.rays)
    assert linf(out['rgb'], g.rgb) <= RGB_TOL


@pytest.mark.parametrize('n', [0, 1, 7, 63, 64, 65, 257])
def test_ragged_ray_counts(fns, n):
    from gpu_common import render_np
    g, fn = fns('donerf_sphere_small')
    full = render_np(fn, g.rays)['rgb']
    part = render_np(fn, g.rays[:n])['rgb'] if n <= g.rays.shape[0] else None
    if part is not None:
        assert part.shape == (n, 3)
        assert np.array_equal(part, full[:n])


def test_internal_chunking_is_invisible(fns):
    """The library's workspace chunking must not change a single bit."""
    from gpu_common import render_np
    g, fn = fns('immersive_sphere_small')
    rays = np.concatenate([g.rays] * 3, 0)
    fn.model.reserve(64)
    assert fn.model.chunk_rays() == 64                  # HR_OPT_CHUNK_RAYS: what hr_stage_* accept (bench.py sizes its stage timing from it)
    a = render_np(fn, rays)['rgb']
    fn.model.reserve(32768)
    assert fn.model.chunk_rays() == 32768
    b = render_np(fn, rays)['rgb']
    assert np.array_equal(a, b)
    assert np.array_equal(a[:g.rays.shape[0]], a[g.rays.shape[0]:2 * g.rays.shape[0]])


def test_ray_order_independence_and_determinism(fns):
    from gpu_common import render_np
    g, fn = fns('technicolor_z_plane_small')
    perm = np.random.default_rng(0).permutation(g.rays.shape[0])
    a = render_np(fn, g.rays)['rgb']
    b = render_np(fn, g.rays[perm])['rgb']
    assert np.array_equal(a[perm], b)
    assert np.array_equal(a, render_np(fn, g.rays)['rgb'])


def test_render_fn_surface(fns):
    """forward / forward_multiple / embed keep the reference's dict-of-(B,k) contract."""
    g, fn = fns('donerf_sphere_small')
    rays = torch.from_numpy(g.rays).cuda()
    out = fn(rays)
    assert set(out.keys()) == {'rgb'} and out['rgb'].shape == (rays.shape[0], 3)
    assert torch.equal(fn.forward_multiple(rays)['rgb'], out['rgb'])
    emb = fn.embed(rays)
    Z = 32
    assert emb['points'].shape == (rays.shape[0], 3 * Z) and emb['distances'].shape == (rays.shape[0], Z)
    assert linf(emb['distances'].cpu().numpy(), g.arrays['distances']) <= 1e-4
    assert linf(emb['color_shift'].cpu().numpy(), g.arrays['color_shift'].reshape(rays.shape[0], -1)) <= 1e-5
    fields = fn(rays, fields=['render_weights', 'distances'])
    assert linf(fields['render_weights'].cpu().numpy(), g.arrays['render_weights']) <= 5e-5
    from hyperreel_amd.render import render_chunked
    chunked = render_chunked(rays, fn, {}, 100)
    assert torch.equal(chunked['rgb'], out['rgb'])


def test_errors_are_loud(fns):
    g, fn = fns('donerf_sphere_small')
    with pytest.raises(RuntimeError):
        fn(torch.from_numpy(g.rays))          # CPU tensor: no fallback
    with pytest.raises(ValueError):
        fn(torch.zeros(4, 3, device='cuda'))
    import ctypes
    from hyperreel_amd import lib, plan
    L = lib.load()
    hc = plan.compile_config(g.cfg, g.dataset, g.grid, iteration=g.iteration)
    h = ctypes.c_void_p()
    assert L.hr_model_create(ctypes.byref(hc), ctypes.byref(h)) == 0
    assert L.hr_model_finalize(h) == -4 and b'never uploaded' in L.hr_last_error()
    buf = torch.zeros(16, device='cuda')
    assert L.hr_model_upload(h, b'mlp.0.bias', ctypes.c_void_p(buf.data_ptr()), 64) == -1
    assert L.hr_model_upload(h, b'nope', ctypes.c_void_p(buf.data_ptr()), 64) == -1
    assert L.hr_render(h, ctypes.c_void_p(buf.data_ptr()), 1, ctypes.c_void_p(buf.data_ptr()), None) == -2
    L.hr_model_destroy(h)
    hc.mlp_hidden = 100
    assert L.hr_model_create(ctypes.byref(hc), ctypes.byref(h)) == -1


def test_state_dict_roundtrip_keys(fns):
    """Parameter names are the reference's checkpoint keys."""
    g, fn = fns('technicolor_z_plane_small')
    keys = set(fn.state_dict().keys())
    for k in g.state_dict:
        assert k in keys, k
    assert 'model.embedding_model.embeddings.0.net.layers.5.weight' in keys
    assert 'model.color_model.net.basis_mat_density.weight' in keys


@pytest.mark.parametrize('precision', PRECISIONS)
def test_full_frame_800x800_properties(precision):
    """BASELINE config 2 at full size (640 000 rays, 600^3 grid): oracle on a random
    subset, plus size-independent properties on the whole frame."""
    from gpu_common import make_render_fn
    from hyperreel_oracle import HyperReelOracle
    cfg, ds = C.model_config('donerf_sphere'), C.dataset_scalars('donerf')
    sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
    rays = scenes.benchmark_rays('donerf_sphere', 800, 800)
    fn = make_render_fn(cfg, ds, sd, mlp_precision=precision)
    r = torch.from_numpy(rays).cuda()
    rgb = fn(r)['rgb']
    torch.cuda.synchronize()
    assert rgb.shape == (640000, 3) and torch.isfinite(rgb).all()
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0
    assert float(rgb.std()) > 0.02
    # determinism and order independence on the whole frame
    assert torch.equal(fn(r)['rgb'], rgb)
    perm = torch.randperm(r.shape[0], device='cuda', generator=torch.Generator('cuda').manual_seed(1))
    assert torch.equal(fn(r[perm].contiguous())['rgb'], rgb[perm])
    # oracle on 4096 random pixels
    idx = np.random.default_rng(3).choice(rays.shape[0], 4096, replace=False)
    ref = HyperReelOracle(cfg, ds, sd).render(rays[idx])['rgb']
    got = rgb[torch.from_numpy(idx).cuda()].cpu().numpy()
    err = np.abs(got - ref).max(-1)
    assert err.max() <= RGB_TOL, f'L-inf {err.max():.3e}, {(err > RGB_TOL).sum()} of 4096 over'
    psnr = -10.0 * np.log10(np.mean((got - ref) ** 2) + 1e-20)
    assert psnr > 90.0


def _variant(name, edit, Z=None):
    cfg = C.model_config(name, z_channels=Z)
    edit(cfg)
    return cfg


def _emb(cfg):
    return cfg.embedding.embeddings


VARIANTS = {
    # Z that is not a power of two: lanes beyond Z idle, sort pads with +inf
    'z24_sphere': lambda: _variant('donerf_sphere', lambda c: None, Z=24),
    'z7_zplane': lambda: _variant('technicolor_z_plane', lambda c: None, Z=7),
    # more than 64 samples per ray: a ray spans 2 or 4 wavefronts (LDS hand-overs in sort / scan / sum)
    'z96_sphere': lambda: _variant('donerf_sphere', lambda c: None, Z=96),
    'z128_zplane': lambda: _variant('technicolor_z_plane', lambda c: None, Z=128),
    'z200_cylinder': lambda: _variant('donerf_cylinder', lambda c: None, Z=200),
    'z256_video_sphere': lambda: _variant('immersive_sphere', lambda c: None, Z=256),
    # sphere origins actually driven by the network (origin_scale_factor != 0): no column pruning
    'sphere_origin_scale': lambda: _variant('donerf_sphere', lambda c: _emb(c).ray_intersect_0.intersect.update(origin_scale_factor=0.05)),
    # hidden width 128: exact fp32-MFMA kernel (the split kernel needs 256)
    'hidden128': lambda: _variant('donerf_cylinder', lambda c: _emb(c).ray_prediction_0.net.update(hidden_channels=128)),
    # two skip layers, deeper net
    'skips_2_4_depth8': lambda: _variant('donerf_sphere', lambda c: _emb(c).ray_prediction_0.net.update(depth=8, skips=[2, 4])),
    # BasicPE with 3 frequencies on Pluecker coordinates
    'basic_pe': lambda: _variant('donerf_sphere', lambda c: _emb(c).ray_prediction_0.params.ray.update(
        pe=C.to_cfg({'type': 'basic', 'n_freqs': 2, 'freq_multiplier': 2.0}))),
    # no contraction at all on a sphere scene, explicit near/far and anchors
    'sphere_no_contract': lambda: _variant('donerf_sphere', lambda c: (
        _emb(c).ray_intersect_0.intersect.pop('contract'),
        _emb(c).ray_intersect_0.intersect.update(use_dataset_bounds=False, initial=0.3, end=2.5, near=0.1, far=2.4))),
    # white background + softplus density + SH shading on the static net
    'static_sh_softplus_white': lambda: _variant('donerf_sphere', lambda c: c.color.net.update(
        white_bg=1, fea2denseAct='softplus', density_shift=-1.0, shadingMode='SH', data_dim_color=27)),
    # RGB shading on the keyframe net, unsorted samples, weight threshold > 0
    'video_rgb_unsorted_thr': lambda: _variant('immersive_sphere', lambda c: (
        c.color.net.update(shadingMode='RGB', data_dim_color=3, rm_weight_mask_thre=1e-3),
        _emb(c).ray_intersect_0.intersect.update(sort=False))),
    # channel counts that are not multiples of four and differ between density and appearance
    'odd_channels': lambda: _variant('donerf_sphere', lambda c: c.color.net.update(n_lamb_sigma=[6, 3, 5], n_lamb_sh=[10, 2, 7])),
    # plane pairs with a single 16-byte channel group (own-sample gather) next to an 8-group pair (quads, two passes)
    'single_group_planes': lambda: _variant('donerf_sphere', lambda c: c.color.net.update(n_lamb_sigma=[3, 12, 2], n_lamb_sh=[0, 20, 0])),
    'video_odd_channels': lambda: _variant('neural_3d_z_plane', lambda c: c.color.net.update(n_lamb_sigma=[5, 0, 3], n_lamb_sh=[9, 0, 2]), Z=16),
}


@pytest.mark.parametrize('variant', sorted(VARIANTS))
def test_config_variants_match_oracle(variant):
    """Options of the supported path that the five shipped families do not exercise."""
    from gpu_common import make_render_fn, render_np
    from hyperreel_oracle import HyperReelOracle
    cfg = VARIANTS[variant]()
    base = 'immersive' if ('video_rgb' in variant or 'video_sphere' in variant) else ('neural_3d' if 'video_odd' in variant else
                                                        ('technicolor' if 'zplane' in variant else 'donerf'))
    ds = C.dataset_scalars(base)
    grid = [33, 27, 30]
    sd = scenes.make_state_dict(cfg, ds, grid, seed=321, density='dense', app_scale=1.0)
    video = cfg.color.net.type == 'tensor_vm_split_time'
    zp = cfg.embedding.embeddings.ray_intersect_0.intersect.type == 'z_plane'
    rays = scenes.random_rays(515, 6, video, pos_mean=(0, 0, 1.0), pos_std=0.15, dir_mean=(0, 0, -1.2), dir_std=0.5) if zp \
        else scenes.random_rays(515, 6, video)
    fn = make_render_fn(cfg, ds, sd)
    out = render_np(fn, rays, want=('distances', 'render_weights'))
    ref = HyperReelOracle(cfg, ds, sd).render(rays, keep='all')
    Z = ref['distances'].shape[1]
    d_ref = ref['distances'].reshape(-1, Z)
    assert float(np.max(np.abs(out['distances'] - d_ref) / (1 + np.abs(d_ref)))) <= 2e-5
    assert linf(out['render_weights'], ref['render_weights']) <= 5e-5
    assert linf(out['rgb'], ref['rgb']) <= RGB_TOL
    assert ref['rgb'].std() > 0.01


def test_all_samples_masked_and_degenerate_rays(fns):
    """Rays whose every sample is masked (outside the box / behind near) must give the reference's
    value: colour_shift-weighted zero, i.e. exactly what the oracle returns."""
    from gpu_common import render_np
    g, fn = fns('donerf_sphere_small')
    orc = _oracle(g)
    rays = np.asarray([[50.0, 50.0, 50.0, 0.57735027, 0.57735027, 0.57735027],
                       [0.0, 0.0, 0.0, 1.0, 0.0, 0.0],
                       [1.999, 1.999, 1.999, 0.0, 0.0, 1.0],
                       [0.0, 0.0, 0.0, 1e-6, 1e-6, 1.0]], np.float32)
    out = render_np(fn, rays, want=('render_weights',))
    ref = orc.render(rays, keep='all')
    assert linf(out['rgb'], ref['rgb']) <= RGB_TOL
    assert linf(out['render_weights'], ref['render_weights']) <= 5e-5


def test_hipgraph_capture_and_replay(fns):
    """Render calls only enqueue kernels on the caller's stream (no allocation, no sync inside the
    library), so a frame can be captured once and replayed -- the viewer path of BASELINE config 5."""
    g, fn = fns('immersive_sphere_small')
    rays = torch.from_numpy(np.concatenate([g.rays] * 4, 0)).cuda()
    ref = fn(rays)['rgb'].clone()
    fn.model.native()
    static_rays = rays.clone()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(static_rays)            # warm-up on the side stream
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(graph):
        static_out = fn(static_rays)['rgb']
    static_out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, ref)
    # new rays through the same captured graph
    static_rays.copy_(torch.flip(rays, dims=[0]))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, torch.flip(ref, dims=[0]))


def test_device_ray_generation_matches_ray_utils(fns):
    """hr_generate_rays == get_ray_directions_K(centered) + get_rays (restated in scenes.pinhole_rays),
    for whole images and for pixel sub-ranges (image-parallel shards), static and video rays."""
    import math
    for case in ('donerf_sphere_small', 'immersive_sphere_small'):
        g, fn = fns(case)
        H, W, fov = 37, 53, 40.0
        pose = scenes.look_at_pose((0.3, -0.1, 0.2), (1.0, 0.1, 0.05))
        focal = 0.5 * W / math.tan(0.5 * math.radians(fov))
        K = np.asarray([[focal, 0, W / 2.0], [0, focal, H / 2.0], [0, 0, 1]], np.float32)
        video = case.startswith('immersive')
        ref = scenes.pinhole_rays(H, W, fov, pose, cam_id=0 if video else None, time=0.25 if video else None)
        got = fn.model.generate_rays(pose, K, W, H, time=0.25 if video else None).cpu().numpy()
        assert got.shape == ref.shape
        assert np.max(np.abs(got - ref)) <= 2e-7
        part = fn.model.generate_rays(pose, K, W, H, time=0.25 if video else None, pixel_range=(500, 1500)).cpu().numpy()
        assert np.array_equal(part, got[500:1500])
        a = fn.model.render_camera(pose, K, W, H, time=0.25 if video else None)
        rays_t = torch.from_numpy(got).cuda()
        # render_camera of a keyframe net states the frame's time (hr_render_frame: bit-identical to render(rays, frame_time=t), within
        # 5e-6 of the general path); a static net takes hr_render either way
        b = fn.model.render(rays_t, frame_time=0.25 if video else None)['rgb']
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        assert float((fn.model.render(rays_t)['rgb'] - a).abs().max()) <= (5e-6 if video else 0.0)


def _round_grids_to_half(sd):
    out = {}
    for k, v in sd.items():
        is_grid = ('_plane' in k or '_line' in k) and v.dtype == np.float32
        out[k] = v.astype(np.float16).astype(np.float32) if is_grid else v
    return out


@pytest.mark.parametrize('case', ['donerf_sphere_small', 'technicolor_z_plane_small', 'immersive_sphere_small',
                                  'sweep/catacaustics_cylinder', 'sweep/donerf_voxel'])
def test_fp16_grids_equal_fp32_path_on_rounded_grids(case):
    """grid_dtype='fp16' (viewer path, BASELINE config 5) only changes the STORAGE of the texels: the
    result must equal the reference algorithm run on the same grids rounded to float16."""
    from gpu_common import make_render_fn, render_np
    from hyperreel_oracle import HyperReelOracle
    g = Golden(case)
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict, grid_dtype='fp16')
    out = render_np(fn, g.rays, want=('render_weights',))
    ref = HyperReelOracle(g.cfg, g.dataset, _round_grids_to_half(g.state_dict)).render(g.rays, keep='all')
    assert linf(out['rgb'], ref['rgb']) <= RGB_TOL
    assert linf(out['render_weights'], ref['render_weights']) <= 5e-5
    # and it is a genuinely different (rounded) scene, close to the fp32 one
    d = linf(out['rgb'], g.rgb)
    assert 0.0 < d < 5e-3, d


def test_fp16_grids_full_frame_psnr():
    """800x800 / 600^3 frame: float16 texels vs float32 texels, PSNR of the image (white-noise grids at
    full resolution are the worst case for texel rounding)."""
    from gpu_common import make_render_fn
    name = 'donerf_sphere'
    cfg, ds = C.model_config(name), C.dataset_scalars(name)
    grid = C.final_grid_size(cfg)
    sd = scenes.make_state_dict(cfg, ds, grid, seed=5, density='dense', app_scale=1.0)
    rays = torch.from_numpy(scenes.benchmark_rays(name)).cuda()
    a = make_render_fn(cfg, ds, sd)(rays)['rgb']
    b = make_render_fn(cfg, ds, sd, grid_dtype='fp16')(rays)['rgb']
    mse = torch.mean((a - b) ** 2).item()
    psnr = 10.0 * np.log10(1.0 / max(mse, 1e-20))
    assert psnr > 60.0, psnr
    assert (a - b).abs().max().item() < 2e-2


@pytest.mark.parametrize('precision', PRECISIONS)
@pytest.mark.parametrize('case', ['sweep/technicolor_cascaded', 'sweep/shiny_z_plane_feedback'])
def test_cascade_intermediates_and_ragged_counts(case, precision):
    """point_prediction cascades (coarse MLP -> coarse intersect -> per-point MLP -> fine intersect): a ray count
    that is not a multiple of anything, both MLP precisions, intermediates of the FINE level against the oracle."""
    from gpu_common import make_render_fn, render_np
    from hyperreel_oracle import HyperReelOracle
    g = Golden(case)
    video = g.rays.shape[1] == 8
    rays = scenes.random_rays(515, 9, video, pos_mean=(0, 0, 1.0), pos_std=0.15, dir_mean=(0, 0, -1.2), dir_std=0.5)
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=precision, iteration=g.iteration)
    out = render_np(fn, rays, want=('distances', 'render_weights'))
    ref = HyperReelOracle(g.cfg, g.dataset, g.state_dict, iteration=g.iteration).render(rays, keep='all')
    Z = ref['distances'].shape[1]
    assert Z == 32
    d_ref = ref['distances'].reshape(-1, Z)
    assert float(np.max(np.abs(out['distances'] - d_ref) / (1 + np.abs(d_ref)))) <= 2e-5
    assert linf(out['render_weights'], ref['render_weights']) <= 5e-5
    assert linf(out['rgb'], ref['rgb']) <= RGB_TOL
    # chunked internally == one shot (the chunk boundary falls inside the row buffer of the point MLP)
    fn.model.reserve(128)
    assert np.array_equal(render_np(fn, rays)['rgb'], out['rgb'])
    # embed(): the fine level's head fields come back in (ray, sample, channel) order
    emb = fn.embed(torch.from_numpy(rays).cuda())
    assert emb['points'].shape == (rays.shape[0], 3 * Z)
    for key in ('color_scale', 'color_shift', 'point_offset'):
        if key in emb and key in ref:
            assert linf(emb[key].cpu().numpy(), ref[key].reshape(rays.shape[0], -1)) <= 2e-5, key


@pytest.mark.parametrize('case', ['donerf_sphere_small', 'immersive_sphere_small', 'sweep/bom_sphere'])
def test_dead_column_pruning_changes_nothing(case, monkeypatch):
    """hr_model_finalize drops head columns the path never reads (origin channels scaled by zero, unread sigma heads) from
    the last Linear; with HR_PRUNE=0 it keeps them.  Same arithmetic on the live columns -> bit-identical images."""
    from gpu_common import make_render_fn, render_np
    g = Golden(case)
    pruned = render_np(make_render_fn(g.cfg, g.dataset, g.state_dict, iteration=g.iteration), g.rays)['rgb']
    monkeypatch.setenv('HR_PRUNE', '0')
    full = render_np(make_render_fn(g.cfg, g.dataset, g.state_dict, iteration=g.iteration), g.rays)['rgb']
    assert np.array_equal(pruned, full)


@pytest.mark.parametrize('precision', PRECISIONS)
def test_cylinder_frame_threshold_decisions(precision):
    """20 000 rays of the 800x800 / 600^3 cylinder frame.  With 1-ulp rcp/sqrt in the distance arithmetic one of them
    lands on the other side of `dist <= near` (RGB off by 5e-2): everything that feeds a threshold is IEEE arithmetic."""
    from gpu_common import make_render_fn, render_np
    from hyperreel_oracle import HyperReelOracle
    name = 'donerf_cylinder'
    cfg, ds = C.model_config(name), C.dataset_scalars(name)
    grid = C.final_grid_size(cfg)
    sd = scenes.make_state_dict(cfg, ds, grid, seed=7, density='dense', app_scale=1.0)
    rays = scenes.benchmark_rays(name, 800, 800, frame=7)
    idx = np.random.default_rng(0).choice(rays.shape[0], 20000, replace=False)
    r = np.ascontiguousarray(rays[idx])
    got = render_np(make_render_fn(cfg, ds, sd, mlp_precision=precision), r)['rgb']
    ref = HyperReelOracle(cfg, ds, sd).render(r)['rgb']
    err = np.abs(got - ref).max(-1)
    assert err.max() <= RGB_TOL, f'{int((err > RGB_TOL).sum())} rays over, worst {err.max():.3e} at frame pixel {int(idx[err.argmax()])}'


@pytest.mark.parametrize('case', initialiser_golden_cases())
def test_f16x2_mode_stays_inside_the_bar(fns, case):
    """mlp_precision='f16x2' (weights rounded once to half, two MFMA products): opt-in speed mode, 0.94 vs 1.28 ms for the
    MLP of an 800x800 frame; its RGB must still be within the north-star tolerance of the reference."""
    from gpu_common import render_np
    g, fn = fns(case, 'f16x2')
    err = np.abs(render_np(fn, g.rays)['rgb'] - g.rgb).max()
    assert err <= RGB_TOL, f'{case}: {err:.3e}'


@pytest.mark.parametrize('model', ['donerf_sphere', 'neural_3d_z_plane'])
def test_full_frame_compositing_invariants(model):
    """Whole 800x800 frames at the shipped grid size, properties that need no oracle: the sorted distances of a ray
    ascend (masked samples are exactly 0 and come first), compositing weights are in [0,1] and sum to at most 1,
    sigma >= 0, and the colour is the weighted sum bounded by sum(w) * max possible colour."""
    from gpu_common import make_render_fn
    cfg, ds = C.model_config(model), C.dataset_scalars(model)
    sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
    rays = torch.from_numpy(scenes.benchmark_rays(model, 800, 800, frame=7)).cuda()
    fn = make_render_fn(cfg, ds, sd)
    Z = int(cfg.embedding.embeddings.ray_prediction_0.z_channels)
    for lo in range(0, rays.shape[0], 160000):                    # (B, Z) diagnostics in slices
        out = fn.model.render(rays[lo:lo + 160000], want=('distances', 'render_weights', 'sigma'))
        d, w, sg = out['distances'], out['render_weights'], out['sigma']
        assert d.shape[1] == Z and torch.isfinite(d).all() and torch.isfinite(w).all()
        assert (sg >= 0).all()
        assert (w >= 0).all() and (w <= 1.0 + 1e-6).all()
        assert float(w.sum(-1).max()) <= 1.0 + 1e-5
        assert (d >= 0).all()
        # the reference sorts the pre-contraction distances; the contraction is monotone, so the final ones ascend too
        assert (d[:, 1:] >= d[:, :-1] - 1e-6).all()
        zero = d == 0
        assert (zero[:, 1:] <= zero[:, :-1]).all()                # zeros form a prefix
        assert (w[zero] == 0).all()                               # a masked sample has no density (distances > 0 test)


@pytest.mark.parametrize('case', ['donerf_sphere_small', 'immersive_sphere_small'])
def test_upsample_volume_grid_matches_interpolate(fns, case):
    """hr_upsample_plane == F.interpolate(mode='bilinear', align_corners=True) on every plane and line, and the model
    renders at the new size like a reference model whose grids were grown by torch (tensorf_base.py:1152-1188)."""
    import torch.nn.functional as F
    from gpu_common import make_render_fn, render_np
    from hyperreel_oracle import HyperReelOracle
    g = Golden(case)
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict, iteration=g.iteration)
    net = fn.model.color_model.net
    old = {k: v.detach().cpu().clone() for k, v in net.named_parameters() if 'plane' in k or 'line' in k}
    target = [57, 49, 41]
    fn.model.upsample_volume_grid(target)
    assert fn.model.grid_size == target
    new_sd = dict(g.state_dict)
    for k, v in old.items():
        got = dict(net.named_parameters())[k].detach().cpu()
        ref = F.interpolate(v, size=tuple(got.shape[2:]), mode='bilinear', align_corners=True)
        assert got.shape == ref.shape
        assert float((got - ref).abs().max()) <= 1e-6 * float(ref.abs().max() + 1), k
        new_sd['model.color_model.net.' + k] = ref.numpy()
    new_sd['model.color_model.net.gridSize'] = np.asarray(target, np.int64)
    out = render_np(fn, g.rays)['rgb']
    ref_rgb = HyperReelOracle(g.cfg, g.dataset, new_sd).render(g.rays)['rgb']
    assert linf(out, ref_rgb) <= RGB_TOL


@pytest.mark.parametrize('case', ['sweep/variant_ease_iter0', 'sweep/variant_ease_iter6000', 'sweep/variant_pe_window_iter3000',
                                  'sweep/variant_mask_stop_iter3'])
def test_set_iter_moves_an_existing_model_through_its_schedules(case):
    """INRSystem.set_train_iter -> model.set_iter(i) every step (nlf/__init__.py:608-614): the same native handle renders
    the converged image, the in-window image of the reference (golden rendered at that iteration) and the converged one
    again -- hr_model_update_config swaps constants only."""
    from gpu_common import make_render_fn, render_np
    from hyperreel_oracle import HyperReelOracle
    g = Golden(case)
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict)              # never told an iteration: converged
    converged = HyperReelOracle(g.cfg, g.dataset, g.state_dict).render(g.rays)['rgb']
    handle = fn.model.native().value
    assert linf(render_np(fn, g.rays)['rgb'], converged) <= RGB_TOL
    fn.model.set_iter(g.iteration)
    assert linf(render_np(fn, g.rays)['rgb'], g.rgb) <= RGB_TOL
    assert linf(g.rgb, converged) > 1e-3              # the schedules are active at that iteration
    fn.model.set_iter(10_000_000)
    assert linf(render_np(fn, g.rays)['rgb'], converged) <= RGB_TOL
    assert fn.model.native().value == handle                          # no re-creation, no re-upload


# ---------------------------------------------------------------------------------------------------------
# Full-size frames of BASELINE configs[1..4] at the shipped final grids: how many of >= 131 072 rays of the 800x800
# benchmark frame differ from the CPU restatement of the reference (oracle/torch_port.py, itself within 2e-6 of the
# reference on every golden set and pinned on `*_full` fixtures at these very grids) by more than the north-star bar?
# A flipped threshold decision of the reference (`dist <= near`, intersect/base.py:194-203; the strict aabb test) moves a
# ray by up to 0.06, so this counts flips.  Must be ZERO in the exact fp32 mode and in the mode a model gets by default.
FULL_FRAME_MODELS = ['donerf_sphere', 'technicolor_z_plane', 'immersive_sphere', 'neural_3d_z_plane']
_full_cache = {}


def _full_frame(model):
    """(cfg, ds, sd, rays of the frame, indices of the checked subset, oracle rgb on them) -- the oracle runs once per model."""
    if model not in _full_cache:
        from torch_port import TorchPort
        cfg, ds = C.model_config(model), C.dataset_scalars(model)
        sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
        rays = scenes.benchmark_rays(model, 800, 800, frame=7)
        idx = np.sort(np.random.default_rng(11).choice(rays.shape[0], 131072, replace=False))
        import os
        torch.set_num_threads(min(16, os.cpu_count() or 8))
        ref = TorchPort(cfg, ds, sd).render(rays[idx], chunk=16384)['rgb']
        _full_cache.clear()                       # one model's grids at a time (the 823x617x514 scene is 0.4 GB)
        _full_cache[model] = (cfg, ds, sd, rays, idx, np.asarray(ref))
    return _full_cache[model]


@pytest.mark.parametrize('precision', ['fp32', 'auto', 'f16x3'])
@pytest.mark.parametrize('model', FULL_FRAME_MODELS)
def test_full_size_frames_have_no_ray_over_the_bar(model, precision):
    from gpu_common import make_render_fn
    cfg, ds, sd, rays, idx, ref = _full_frame(model)
    fn = make_render_fn(cfg, ds, sd, mlp_precision=precision)
    rgb = fn.model.render(torch.from_numpy(rays).cuda())['rgb']           # the whole 800x800 frame
    torch.cuda.synchronize()
    assert torch.isfinite(rgb).all()
    got = rgb[torch.from_numpy(idx).cuda()].cpu().numpy()
    err = np.abs(got - ref).max(-1)
    over = int((err > RGB_TOL).sum())
    assert over == 0, f'{model} / {precision}: {over} of {idx.size} rays over 1e-4 (worst {err.max():.3e} at ray {int(idx[err.argmax()])})'
    if model != 'donerf_sphere':
        # the same frame through hr_render_frame -- the entry bench.py's `families` figures use, a different summation order
        # (nlf/nets/tensorf_dynamic.py:287-371: the two keyframe rows blended into a line first) -- held to the same count
        rgb = fn.model.render(torch.from_numpy(rays).cuda(), frame_time=float(rays[0, -1]))['rgb']
        torch.cuda.synchronize()
        assert torch.isfinite(rgb).all()
        err = np.abs(rgb[torch.from_numpy(idx).cuda()].cpu().numpy() - ref).max(-1)
        over = int((err > RGB_TOL).sum())
        assert over == 0, f'{model} / {precision} through hr_render_frame: {over} of {idx.size} rays over 1e-4 (worst {err.max():.3e})'


def test_two_product_mode_on_the_benchmark_frame():
    """The opt-in `f16x2` arithmetic (weights rounded once to half: two MFMA products per GEMM instead of three) that bench.py
    reports as `value_f16x2`: on the benchmark frame it stays inside the bar with margin.  (On the keyframe families it is inside
    with little margin -- 7.5e-5 / 8.7e-5 measured -- which is why it is not the default; no assertion is made there.)"""
    from gpu_common import make_render_fn
    cfg, ds, sd, rays, idx, ref = _full_frame('donerf_sphere')
    fn = make_render_fn(cfg, ds, sd, mlp_precision='f16x2')
    rgb = fn.model.render(torch.from_numpy(rays).cuda())['rgb']
    err = np.abs(rgb[torch.from_numpy(idx).cuda()].cpu().numpy() - ref).max(-1)
    assert int((err > RGB_TOL).sum()) == 0 and float(err.max()) <= 6e-5, f'worst {err.max():.3e}'


@pytest.mark.parametrize('model', ['immersive_sphere', 'donerf_sphere'])
def test_full_size_frames_with_float16_texels(model):
    """BASELINE configs[4] (viewer path, fp16 grids): the same count against the reference algorithm run on the rounded grids."""
    from gpu_common import make_render_fn
    from torch_port import TorchPort
    cfg, ds, sd, rays, idx, _ = _full_frame(model)
    sd16 = {k: (v.astype(np.float16).astype(np.float32) if ('_plane' in k or '_line' in k) else v) for k, v in sd.items()}
    ref = np.asarray(TorchPort(cfg, ds, sd16).render(rays[idx[:65536]], chunk=16384)['rgb'])
    fn = make_render_fn(cfg, ds, sd, grid_dtype='fp16')
    rgb = fn.model.render(torch.from_numpy(rays).cuda())['rgb']
    got = rgb[torch.from_numpy(idx[:65536]).cuda()].cpu().numpy()
    err = np.abs(got - ref).max(-1)
    assert int((err > RGB_TOL).sum()) == 0, f'{int((err > RGB_TOL).sum())} rays over, worst {err.max():.3e}'
