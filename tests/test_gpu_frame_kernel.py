"""`-m gpu`: the persistent frame kernel (csrc/fused_impl.inc) -- MLP wavefronts and sample wavefronts of one workgroup,
head tile handed over in LDS -- against the two-kernel path through the HBM workspace.  Both run the same arithmetic
(mlp_split_core.inc, sample_core.inc), so the images must agree BIT FOR BIT; any difference is a hand-over bug
(stale LDS, a counter race, an overlay written too early).  The hand-over is exercised under uneven load: whole 800x800
frames (10 000 tiles over 256 persistent workgroups), ragged tails, every word compared, repeated launches.
Reference path being fused: nlf/embedding/ray.py:332-337 -> nlf/intersect/base.py:142-259."""
import numpy as np
import pytest
import torch

from helpers import Golden
from hyperreel_amd import config as C
from hyperreel_amd import scenes

pytestmark = pytest.mark.gpu

FUSABLE = ['donerf_sphere_small', 'donerf_cylinder_small', 'config1_random_z16']


def _fns(case, precision='f16x3', grid_dtype='fp32'):      # ('auto' is the verified two-pass plan since round 5: it never takes the frame kernel)
    from gpu_common import make_render_fn
    g = Golden(case)
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=precision, grid_dtype=grid_dtype, iteration=g.iteration)
    return g, fn


def _render(fn, rays_t, frame_kernel, sample_waves=None):
    fn.model.set_execution(frame_kernel=frame_kernel, sample_waves=sample_waves)
    # rendered into a NaN-filled buffer: a ray that no wavefront processed must not inherit a plausible value from whatever
    # the allocator handed back (an earlier image of the same rays, typically)
    out = torch.full((rays_t.shape[0], 3), float('nan'), dtype=torch.float32, device=rays_t.device)
    fn.model.render(rays_t, out=out)
    torch.cuda.synchronize()
    assert not torch.isnan(out).any()
    return out


@pytest.mark.parametrize('case', FUSABLE)
def test_fusable_models_take_the_frame_kernel(case):
    g, fn = _fns(case)
    assert not fn.model.frame_kernel_active()          # the default plan is two kernels per chunk (round 5: level in time, tighter tail)
    fn.model.set_execution(frame_kernel=True)
    assert fn.model.frame_kernel_active()              # opt-in: fits wherever the head tile fits
    fn.model.set_execution(frame_kernel=False)
    assert not fn.model.frame_kernel_active()
    fn.model.set_execution(frame_kernel=True)
    assert fn.model.frame_kernel_active()


KEYFRAME = ['technicolor_z_plane_small', 'immersive_sphere_small', 'neural_3d_z_plane_small']


def test_cascades_keep_the_two_kernel_path():
    """a point_prediction cascade runs two MLPs with a sample pass between them"""
    g, fn = _fns('sweep/shiny_z_plane_cascaded')
    fn.model.set_execution(frame_kernel=2)
    assert not fn.model.frame_kernel_active()


@pytest.mark.parametrize('case', KEYFRAME)
def test_keyframe_families_take_the_32_ray_tile_frame_kernel(case):
    """TensorVMKeyframeTime heads (nlf/nets/tensorf_dynamic.py:645-839) are 480 columns wide (960 at 64 samples per ray): 32-ray tiles,
    two head buffers at 32 samples per ray, one at 64.  Measured as fast as or slower than the two-kernel plan (every weight
    crosses the CU once per 32 rays): not the default -- the plan `frame_kernel=2` asks for (no head workspace traffic)"""
    g, fn = _fns(case)
    assert not fn.model.frame_kernel_active()
    fn.model.set_execution(frame_kernel=2)
    assert fn.model.frame_kernel_active()
    fn.model.set_execution(frame_kernel=False)
    assert not fn.model.frame_kernel_active()


@pytest.mark.parametrize('grid_dtype', ['fp32', 'fp16'])
@pytest.mark.parametrize('precision', ['bf16x3', 'f16x3', 'f16x2', 'f16f8'])
@pytest.mark.parametrize('case', KEYFRAME)
def test_keyframe_frame_kernel_equals_two_kernel_path_bit_for_bit(case, precision, grid_dtype):
    g, fn = _fns(case, precision, grid_dtype)
    rays = torch.from_numpy(np.concatenate([g.rays] * 3 + [g.rays[:37]], 0)).cuda()       # a few tiles per workgroup, ragged tail
    two = _render(fn, rays, False)
    one = _render(fn, rays, 2)
    assert fn.model.frame_kernel_active()
    assert torch.equal(one, two), f'{int((one != two).any(-1).sum())} rays differ'
    for _ in range(3):
        assert torch.equal(_render(fn, rays, 2), two)
    if precision != 'f16x2' and grid_dtype == 'fp32':
        assert np.abs(one[:g.rays.shape[0]].cpu().numpy() - g.rgb).max() <= 1e-4


@pytest.mark.parametrize('model', ['technicolor_z_plane', 'immersive_sphere', 'neural_3d_z_plane'])
def test_keyframe_frame_kernel_full_frame_every_word_and_repeats(model):
    """the families' 800x800 frames at their shipped grids: 20 000 32-ray tiles over 256 persistent workgroups, every word equal to
    the two-kernel path, repeated launches reproduce it"""
    from gpu_common import make_render_fn
    cfg, ds = C.model_config(model), C.dataset_scalars(model)
    sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
    fn = make_render_fn(cfg, ds, sd, mlp_precision="f16x3")
    rays = torch.from_numpy(scenes.benchmark_rays(model, 800, 800, frame=7)).cuda()
    two = _render(fn, rays, False)
    one = _render(fn, rays, 2)
    assert fn.model.frame_kernel_active()
    assert torch.equal(one, two), f'{int((one != two).any(-1).sum())} rays differ'
    for _ in range(4):
        assert torch.equal(_render(fn, rays, 2), two)


@pytest.mark.parametrize('grid_dtype', ['fp32', 'fp16'])
@pytest.mark.parametrize('precision', ['bf16x3', 'f16x3', 'f16x2', 'f16f8'])
@pytest.mark.parametrize('waves', [4, 8])
@pytest.mark.parametrize('case', FUSABLE)
def test_frame_kernel_equals_two_kernel_path_bit_for_bit(case, waves, precision, grid_dtype):
    g, fn = _fns(case, precision, grid_dtype)
    rays = torch.from_numpy(np.concatenate([g.rays] * 3, 0)).cuda()       # a few tiles per workgroup, ragged tail
    two = _render(fn, rays, False)
    one = _render(fn, rays, True, waves)
    assert fn.model.frame_kernel_active()
    assert torch.equal(one, two), f'{int((one != two).any(-1).sum())} rays differ'
    if precision != 'f16x2' and grid_dtype == 'fp32':
        err = (one[:g.rays.shape[0]].cpu().numpy() - g.rgb)
        assert np.abs(err).max() <= 1e-4


@pytest.mark.parametrize('n', [0, 1, 7, 63, 64, 65, 127, 129, 257])
def test_frame_kernel_ragged_ray_counts(n):
    g, fn = _fns('donerf_sphere_small')
    rays = torch.from_numpy(np.concatenate([g.rays] * 2, 0)).cuda()
    full = _render(fn, rays, True)
    part = _render(fn, rays[:n].contiguous(), True)
    assert part.shape == (n, 3) and torch.equal(part, full[:n])


@pytest.mark.parametrize('waves', [4, 8])
def test_frame_kernel_full_frame_every_word_and_repeats(waves):
    """BASELINE configs[1] at full size: 640 000 rays = 10 000 tiles, 39-40 per persistent workgroup.  Every rgb word equals
    the two-kernel path; thirty more launches (hand-over timing differs from run to run) reproduce it exactly; rays in a
    different order (other tiles share a workgroup) give the same pixels.
    waves = 8: until round 5 a launch could differ from the two-kernel image in ONE ray (the last 16 lanes of a sample wavefront that shares its
    SIMD with MFMA wavefronts lost one term of a sum of products, about once per 10^7 rays: one compiler-formed packed-fp32 instruction); the
    library is now built without the compiler's packed-fp32 instructions (hyperreel_amd/build.py) and the form is held to the same
    "every word, every launch" again."""
    from gpu_common import make_render_fn
    cfg, ds = C.model_config('donerf_sphere'), C.dataset_scalars('donerf')
    sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
    fn = make_render_fn(cfg, ds, sd, mlp_precision="f16x3")
    rays = torch.from_numpy(scenes.benchmark_rays('donerf_sphere', 800, 800, frame=7)).cuda()
    two = _render(fn, rays, False)

    def same(a, b):
        return torch.equal(a, b)
    one = _render(fn, rays, True, waves)
    assert same(one, two), f'{int((one != two).any(-1).sum())} rays differ'
    for _ in range(30):
        assert same(_render(fn, rays, True, waves), two)
    perm = torch.randperm(rays.shape[0], device='cuda', generator=torch.Generator('cuda').manual_seed(5))
    assert same(_render(fn, rays[perm].contiguous(), True, waves), two[perm])
    odd = rays[:555555].contiguous()                                  # 8680 tiles + a 35-ray tail
    assert same(_render(fn, odd, True, waves), two[:555555])


def test_frame_kernel_under_a_concurrent_stream():
    """Uneven load: another stream keeps the CUs busy with a memory-bound kernel while the frame kernel runs."""
    from gpu_common import make_render_fn
    cfg, ds = C.model_config('donerf_sphere'), C.dataset_scalars('donerf')
    sd = scenes.make_state_dict(cfg, ds, [96, 96, 96], seed=3, density='dense', app_scale=1.0)
    fn = make_render_fn(cfg, ds, sd, mlp_precision="f16x3")
    rays = torch.from_numpy(scenes.benchmark_rays('donerf_sphere', 400, 400, frame=3)).cuda()
    two = _render(fn, rays, False)
    fn.model.set_execution(frame_kernel=True)
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device='cuda')
    for _ in range(5):
        with torch.cuda.stream(side):
            for _ in range(20):
                junk.add_(1.0)
        out = torch.full_like(two, float('nan'))
        fn.model.render(rays, out=out)
        torch.cuda.synchronize()
        assert torch.equal(out, two)


def test_frame_kernel_in_a_hipgraph():
    g, fn = _fns('donerf_sphere_small')
    rays = torch.from_numpy(np.concatenate([g.rays] * 8, 0)).cuda()
    ref = _render(fn, rays, False)
    fn.model.set_execution(frame_kernel=True)
    fn.model.render(rays)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn.model.render(rays)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fn.model.render(rays)['rgb']
    for _ in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
