"""`-m gpu`: no kernel of the render path reads a vector register or an LDS word it has not written.

VERDICT r4 item 2: two models rendering on two streams gave, rarely, a frame in which ONE ray differed -- the signature of an uninitialised
read: alone on a stream the previous occupant of a CU's registers / LDS is always the same kernel (same garbage, reproducible image), beside
another launch it is that launch's data.  tests/c_abi/poison.hip makes the garbage a test input: it leaves a chosen 32-bit pattern in every
VGPR and every LDS word of every CU; the same rays are then rendered after each of several patterns -- zeros, a signalling NaN, infinity,
fp16 1.0 pairs (what an f16 MFMA kernel leaves), all ones -- and every word of every image must be the same.  The poison goes in front of
each kernel separately (through hr_stage_mlp / hr_stage_samples) and in front of whole hr_render calls of both execution plans.

The reference is order- and concurrency-independent by construction (nlf/rendering.py:100-150: pure tensor ops per chunk)."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import Golden

pytestmark = pytest.mark.gpu

PATTERNS = [0x00000000, 0x7FA00000, 0x3C003C00, 0xFFFFFFFF, 0x7F800000, 0x42F60000]     # 0, sNaN, half 1.0 pairs, all ones (a NaN), +inf, 123.0
CASES = [('donerf_sphere_small', 'fp32'), ('donerf_sphere_small', 'fp16'), ('immersive_sphere_small', 'fp32'), ('technicolor_z_plane_small', 'fp32'),
         ('neural_3d_z_plane_small', 'fp32'), ('config1_random_z16', 'fp16'), ('donerf_cylinder_small', 'fp32')]


def _setup(case, grid_dtype, precision, n_target=120000):
    from gpu_common import make_render_fn
    g = Golden(case)
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=precision, grid_dtype=grid_dtype, iteration=g.iteration)
    rep = max(1, n_target // g.rays.shape[0])
    rays = torch.from_numpy(np.concatenate([g.rays] * rep + [g.rays[:37]], 0)).cuda()          # ragged tail: the last block is partly empty
    return g, fn, rays


def _describe(a, b, zp_hint=32):
    bad = (a != b).any(-1).nonzero().flatten().cpu().numpy()
    d = float((a - b).abs().max())
    return f'{len(bad)} rays differ (max |d| {d:.3e}); first rays {bad[:8].tolist()}, ray index mod 8: {np.bincount(bad % 8, minlength=8).tolist()}'


def test_the_poison_kernel_reaches_every_cu():
    from gpu_common import Poison
    p = Poison()
    p.touched.zero_()
    p(0x12345678)
    torch.cuda.synchronize()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert int(p.touched.item()) == 8 * cus


@pytest.mark.parametrize('precision', ['f16x3', 'bf16x3'])
@pytest.mark.parametrize('case,grid_dtype', CASES)
def test_each_kernel_of_the_two_kernel_plan_after_poison(case, grid_dtype, precision):
    """K1 -> [poison] -> sample kernel, and [poison] -> K1 -> sample kernel, through the stage entry points."""
    from gpu_common import Poison
    from hyperreel_amd import lib as hlib
    g, fn, rays = _setup(case, grid_dtype, precision)
    m = fn.model
    m.set_execution(frame_kernel=False)
    L = hlib.load()
    h = m.native()
    n = min(rays.shape[0], 65536)                 # one workspace chunk of every family (Z = 64: 65 536 rays)
    rays = rays[:n].contiguous()
    m.render(rays)                                    # sizes the workspace, runs any lazy set-up
    torch.cuda.synchronize()
    poison = Poison()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def stage_mlp():
        hlib.check(L.hr_stage_mlp(h, ctypes.c_void_p(rays.data_ptr()), n, stream), 'hr_stage_mlp')

    def stage_samples(out):
        hlib.check(L.hr_stage_samples(h, ctypes.c_void_p(rays.data_ptr()), n, ctypes.c_void_p(out.data_ptr()), stream), 'hr_stage_samples')

    stage_mlp()
    ref = torch.full((n, 3), float('nan'), device='cuda')
    stage_samples(ref)
    torch.cuda.synchronize()
    assert torch.equal(ref, m.render(rays)['rgb'])
    for where in ('before_samples', 'before_mlp'):
        for pat in PATTERNS:
            for rep in range(2):
                out = torch.full((n, 3), float('nan'), device='cuda')
                if where == 'before_mlp':
                    poison(pat)
                stage_mlp()
                if where == 'before_samples':
                    poison(pat)
                stage_samples(out)
                torch.cuda.synchronize()
                assert torch.equal(out, ref), f'{case} {grid_dtype} {precision}: poison {pat:#010x} {where}: {_describe(out, ref)}'


@pytest.mark.parametrize('plan', [False, True, 2])
@pytest.mark.parametrize('case,grid_dtype', CASES)
def test_whole_renders_after_poison(case, grid_dtype, plan):
    from gpu_common import Poison
    g, fn, rays = _setup(case, grid_dtype, 'f16x3')
    m = fn.model
    m.set_execution(frame_kernel=plan)
    ref = m.render(rays)['rgb'].clone()
    torch.cuda.synchronize()
    poison = Poison()
    for pat in PATTERNS:
        out = torch.full_like(ref, float('nan'))
        poison(pat)
        m.render(rays, out=out)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), f'{case} {grid_dtype} plan {plan} (frame kernel active: {m.frame_kernel_active()}): poison {pat:#010x}: {_describe(out, ref)}'


@pytest.mark.parametrize('case,grid_dtype,precision', [('config1_random_z16', 'fp16', 'f16x3'), ('donerf_sphere_small', 'fp16', 'f16x3'), ('immersive_sphere_small', 'fp32', 'auto')])
def test_two_models_on_two_streams_render_what_each_renders_alone(case, grid_dtype, precision):
    """What VERDICT r4 item 2 observed, as a test: the MLP kernel of one stream runs beside the sample kernel of the other on the same CUs
    (tools/concurrent_streams_stress.py found 0 - 6 differing images in 40 such rounds before round 5's build change, hyperreel_amd/build.py;
    0 in 480 since).  60 rounds x 2 models x 3 renders; every word of every image is the one the model renders alone."""
    from gpu_common import make_render_fn
    g = Golden(case)
    rep = max(1, 160000 // g.rays.shape[0])
    rays = torch.from_numpy(np.concatenate([g.rays] * rep + [g.rays[:37]], 0)).cuda()
    fns = [make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=precision, grid_dtype=grid_dtype, iteration=g.iteration) for _ in range(2)]
    ref = fns[0].model.render(rays)['rgb'].clone()
    assert torch.equal(fns[1].model.render(rays)['rgb'], ref)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.empty_like(ref), torch.empty_like(ref)]
    for it in range(60):
        for f, s, o in zip(fns, streams, outs):
            with torch.cuda.stream(s):
                for _ in range(3):
                    f.model.render(rays, out=o)
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            assert torch.equal(o, ref), f'{case} round {it} model {i}: {_describe(o, ref)}'
