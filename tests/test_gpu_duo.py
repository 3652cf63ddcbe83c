"""`-m gpu`: the co-resident pair (HR_OPT_FRAME_KERNEL 3; csrc/fused_impl.inc HR_DUO_KERNEL + csrc/sampleq_kernel.hip) -- a persistent
MLP kernel and the sample kernel's grid on two streams, the head handed over tile by tile through a write-through workspace and
per-tile flags -- against the two-kernel path.  Same arithmetic sources (mlp_split_core.inc, sample_core.inc), so images must agree
BIT FOR BIT; a difference is a hand-over bug (a stale line, a flag that overtook its payload, a block that read the wrong tile).
Exercised the way MI355X_MICROARCH.md asks for hand-offs to be tested: every word compared, repeated launches over the SAME
workspace addresses (the consumer's caches have seen the previous frame's lines), uneven load, ragged tails, rays in another order.
Reference path being pipelined: nlf/embedding/ray.py:332-337 -> nlf/intersect/base.py:142-259."""
import numpy as np
import pytest
import torch

from helpers import Golden
from hyperreel_amd import config as C
from hyperreel_amd import scenes

pytestmark = pytest.mark.gpu

CASES = ['donerf_sphere_small', 'donerf_cylinder_small', 'config1_random_z16', 'technicolor_z_plane_small', 'immersive_sphere_small',
         'neural_3d_z_plane_small']


def _fns(case, precision='auto', grid_dtype='fp32'):
    from gpu_common import make_render_fn
    g = Golden(case)
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=precision, grid_dtype=grid_dtype, iteration=g.iteration)
    return g, fn


def _render(fn, rays_t, plan, duo=None, frame_time=None):
    fn.model.set_execution(frame_kernel=plan, duo=duo)
    out = torch.full((rays_t.shape[0], 3), float('nan'), dtype=torch.float32, device=rays_t.device)
    fn.model.render(rays_t, out=out, frame_time=frame_time)
    torch.cuda.synchronize()
    assert not torch.isnan(out).any()
    return out


@pytest.mark.parametrize('case', CASES)
def test_every_benchmark_family_can_take_the_pair(case):
    g, fn = _fns(case)
    fn.model.set_execution(frame_kernel='duo')
    assert fn.model.plan_active() == 'duo'
    fn.model.set_execution(frame_kernel=False)
    assert fn.model.plan_active() == 'two_kernels'


def test_cascades_and_the_exact_fp32_mlp_keep_their_plans():
    g, fn = _fns('sweep/shiny_z_plane_cascaded')
    fn.model.set_execution(frame_kernel='duo')
    assert fn.model.plan_active() != 'duo'
    g, fn = _fns('donerf_sphere_small', precision='fp32')
    fn.model.set_execution(frame_kernel='duo')
    assert fn.model.plan_active() != 'duo'
    rays = torch.from_numpy(g.rays).cuda()
    assert np.abs(_render(fn, rays, 'duo').cpu().numpy() - g.rgb).max() <= 1e-4          # (rendered by the plan it fell back to)


@pytest.mark.parametrize('grid_dtype', ['fp32', 'fp16'])
@pytest.mark.parametrize('precision', ['bf16x3', 'f16x3', 'f16x2'])
@pytest.mark.parametrize('case', CASES)
def test_pair_equals_two_kernel_path_bit_for_bit(case, precision, grid_dtype):
    g, fn = _fns(case, precision, grid_dtype)
    rays = torch.from_numpy(np.concatenate([g.rays] * 40 + [g.rays[:37]], 0)).cuda()      # a few tiles per producer, ragged tail
    two = _render(fn, rays, False)
    duo = _render(fn, rays, 'duo')
    assert fn.model.plan_active() == 'duo'
    assert torch.equal(duo, two), f'{int((duo != two).any(-1).sum())} rays differ'
    for _ in range(3):                                     # the same workspace addresses again
        assert torch.equal(_render(fn, rays, 'duo'), two)
    assert not fn.model.plan_faulted()
    if precision != 'f16x2' and grid_dtype == 'fp32':
        assert np.abs(duo[:g.rays.shape[0]].cpu().numpy() - g.rgb).max() <= 1e-4


@pytest.mark.parametrize('waves', [3, 4, 6, 8])
def test_every_producer_shape_gives_the_same_image(waves):
    g, fn = _fns('donerf_sphere_small')
    rays = torch.from_numpy(np.concatenate([g.rays] * 60, 0)).cuda()
    two = _render(fn, rays, False)
    assert torch.equal(_render(fn, rays, 'duo', {'mlp_waves': waves}), two)
    assert not fn.model.plan_faulted()


@pytest.mark.parametrize('n', [0, 1, 7, 63, 64, 65, 127, 129, 257, 4097])
def test_pair_ragged_ray_counts(n):
    g, fn = _fns('donerf_sphere_small')
    rays = torch.from_numpy(np.concatenate([g.rays] * 50, 0)).cuda()
    full = _render(fn, rays, 'duo')
    part = _render(fn, rays[:n].contiguous(), 'duo')
    assert part.shape == (n, 3) and torch.equal(part, full[:n])
    assert not fn.model.plan_faulted()


@pytest.mark.parametrize('model', ['donerf_sphere', 'technicolor_z_plane', 'neural_3d_z_plane'])
def test_pair_full_frame_every_word_and_repeats(model):
    """the 800x800 frames at the shipped grids (10 000 tiles, 80 000 / 160 000 consumer blocks): every rgb word equals the two-kernel
    path, five more launches reproduce it, a permuted ray list gives the same pixels, an odd count leaves a ragged tail; the keyframe
    nets also through hr_render_frame (the frame's keyframe rows read as lines)"""
    from gpu_common import make_render_fn
    cfg, ds = C.model_config(model), C.dataset_scalars(model)
    sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
    fn = make_render_fn(cfg, ds, sd)
    rays_np = scenes.benchmark_rays(model, 800, 800, frame=7)
    rays = torch.from_numpy(rays_np).cuda()
    two = _render(fn, rays, False)
    duo = _render(fn, rays, 'duo')
    assert torch.equal(duo, two), f'{int((duo != two).any(-1).sum())} rays differ'
    for _ in range(5):
        assert torch.equal(_render(fn, rays, 'duo'), two)
    perm = torch.randperm(rays.shape[0], device='cuda', generator=torch.Generator('cuda').manual_seed(5))
    assert torch.equal(_render(fn, rays[perm].contiguous(), 'duo'), two[perm])
    odd = rays[:555555].contiguous()
    assert torch.equal(_render(fn, odd, 'duo'), two[:555555])
    if model != 'donerf_sphere':
        t = float(rays_np[0, -1])
        assert torch.equal(_render(fn, rays, 'duo', frame_time=t), _render(fn, rays, False, frame_time=t))
    assert not fn.model.plan_faulted()


def test_pair_under_a_concurrent_stream():
    """Uneven load: a third stream keeps the memory system busy while the pair runs."""
    from gpu_common import make_render_fn
    cfg, ds = C.model_config('donerf_sphere'), C.dataset_scalars('donerf')
    sd = scenes.make_state_dict(cfg, ds, [96, 96, 96], seed=3, density='dense', app_scale=1.0)
    fn = make_render_fn(cfg, ds, sd)
    rays = torch.from_numpy(scenes.benchmark_rays('donerf_sphere', 400, 400, frame=3)).cuda()
    two = _render(fn, rays, False)
    fn.model.set_execution(frame_kernel='duo')
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
    assert not fn.model.plan_faulted()


def test_a_captured_call_takes_a_one_stream_plan():
    """a replayed hipGraph serialises the two kernel nodes: inside a capture the library must not use the pair"""
    g, fn = _fns('donerf_sphere_small')
    rays = torch.from_numpy(np.concatenate([g.rays] * 8, 0)).cuda()
    ref = _render(fn, rays, False)
    fn.model.set_execution(frame_kernel='duo')
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
    assert not fn.model.plan_faulted()


def test_one_stream_measurement_mode_and_the_timestamps():
    """HR_OPT_DUO_MODE 1 (producer, then consumer, on one stream) renders the same image; hr_debug_duo_times reports both kernels"""
    import ctypes
    from hyperreel_amd import lib as hl
    g, fn = _fns('donerf_sphere_small')
    rays = torch.from_numpy(np.concatenate([g.rays] * 60, 0)).cuda()
    two = _render(fn, rays, False)
    assert torch.equal(_render(fn, rays, 'duo', {'mode': 1}), two)
    t = (ctypes.c_uint64 * 8)()
    hl.check(hl.load().hr_debug_duo_times(fn.model.native(), t), 'hr_debug_duo_times')
    assert t[0] and t[1] > t[0] and t[2] >= t[1]                      # serialised: the consumer started after the producer had ended
    assert torch.equal(_render(fn, rays, 'duo', {'mode': 0}), two)
    hl.check(hl.load().hr_debug_duo_times(fn.model.native(), t), 'hr_debug_duo_times')
    assert t[0] and t[2] and t[3] > t[2]
    assert not fn.model.plan_faulted()
