"""`-m gpu`: hr_render_frame -- one frame of a keyframe net, every ray at the same time: the time planes are read as the keyframe ROW that
time quantises to (2 line taps instead of 4 time-plane taps).  Against hr_render on the same rays: equal up to the ~1e-7 weight the
general path gives the neighbouring row; static nets and both execution plans unchanged."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GRIDS = {'technicolor_z_plane': [44, 36, 20], 'neural_3d_z_plane': [40, 30, 26], 'immersive_sphere': [36, 40, 32], 'donerf_sphere': [32, 32, 32]}


def _fn(model, **kw):
    from gpu_common import make_render_fn
    from hyperreel_amd import config as C, scenes
    cfg, ds = C.model_config(model), C.dataset_scalars(model)
    sd = scenes.make_state_dict(cfg, ds, GRIDS[model], 17, 'dense', 1.0)
    return make_render_fn(cfg, ds, sd, **kw), ds


@pytest.mark.parametrize('plan', [False, 2])
@pytest.mark.parametrize('model', ['technicolor_z_plane', 'neural_3d_z_plane', 'immersive_sphere'])
def test_a_frame_at_one_time_reads_the_keyframe_row(model, plan):
    from hyperreel_amd import scenes
    fn, ds = _fn(model)
    fn.model.set_execution(frame_kernel=plan)
    for frame in (0, 7, 18, 49):
        rays = torch.from_numpy(np.ascontiguousarray(scenes.benchmark_rays(model, 96, 64, frame=frame), np.float32)).cuda()
        t = float(rays[0, -1])
        assert float(rays[:, -1].min()) == t == float(rays[:, -1].max())
        general = fn.model.render(rays)['rgb'].clone()
        frame_img = fn.model.render(rays, frame_time=t)['rgb'].clone()
        torch.cuda.synchronize()
        assert float((general - frame_img).abs().max()) <= 5e-6, (model, frame)
        again = fn.model.render(rays)['rgb']                       # the statement holds for that call only
        assert torch.equal(again, general)


def test_static_nets_ignore_the_frame_time():
    from hyperreel_amd import scenes
    fn, _ = _fn('donerf_sphere')
    rays = torch.from_numpy(np.ascontiguousarray(scenes.benchmark_rays('donerf_sphere', 64, 64, frame=3), np.float32)).cuda()
    assert torch.equal(fn.model.render(rays)['rgb'], fn.model.render(rays, frame_time=0.25)['rgb'])


@pytest.mark.parametrize('grid_dtype', ['fp32', 'fp16'])
def test_render_camera_of_a_keyframe_net_takes_the_frame_path(grid_dtype):
    from hyperreel_amd import scenes
    fn, _ = _fn('immersive_sphere', grid_dtype=grid_dtype)
    pose = scenes.look_at_pose((0.3, 0.0, 0.0), (1.0, 0.1, 0.05))
    K = np.array([[80.0, 0, 48.0], [0, 80.0, 32.0], [0, 0, 1]], np.float32)
    t = 18 / 49.0
    img = fn.model.render_camera(pose, K, 96, 64, time=t)
    rays = fn.model.generate_rays(pose, K, 96, 64, t)
    ref = fn.model.render(rays)['rgb']
    torch.cuda.synchronize()
    d = float((img - ref).abs().max())
    assert d <= 5e-6
    if grid_dtype == 'fp32':
        assert d > 0.0           # the re-associated blend: an image equal bit for bit would mean the general path was taken (float16 texels do take it)


# ---------------------------------------------------------------------------------------------------------
# hr_render_frame against what the REFERENCE rendered (TensorVMKeyframeTime.forward, nlf/nets/tensorf_dynamic.py:287-371,645-839): the
# three full-size fixtures (shipped grids, 32 768 / 256 rays of the 800x800 benchmark frame + random and degenerate rays) and the three
# small ones.  A fixture's rays are grouped by their time -- one big group (the frame) and a few hundred single rays -- and every group
# goes through hr_render_frame with its time: the same bar hr_render is held to (tests/test_gpu_parity.py).
FRAME_GOLDENS = ['technicolor_full', 'neural_3d_full', 'immersive_full', 'technicolor_z_plane_small', 'immersive_sphere_small',
                 'neural_3d_z_plane_small']


@pytest.mark.parametrize('plan', [True, 2])
@pytest.mark.parametrize('precision', ['fp32', 'f16x3', 'bf16x3'])
@pytest.mark.parametrize('case', FRAME_GOLDENS)
def test_hr_render_frame_matches_the_reference_goldens(case, precision, plan):
    from gpu_common import make_render_fn
    from helpers import Golden
    if precision == 'fp32' and plan != True:
        pytest.skip('the exact-fp32 MLP has one plan')
    g = Golden(case)
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=precision)
    fn.model.set_execution(frame_kernel=plan)
    rays = torch.from_numpy(g.rays).cuda()
    times = g.rays[:, -1]
    out = torch.full((rays.shape[0], 3), float('nan'), dtype=torch.float32, device='cuda')
    groups = 0
    for t in np.unique(times):
        idx = torch.from_numpy(np.nonzero(times == t)[0]).cuda()
        out[idx] = fn.model.render(rays[idx].contiguous(), frame_time=float(t))['rgb']
        groups += 1
    torch.cuda.synchronize()
    assert groups > 100 and not torch.isnan(out).any()
    err = np.abs(out.cpu().numpy() - g.rgb).max(-1)
    assert float(err.max()) <= 1e-4, f'{case} / {precision} / plan {plan}: {int((err > 1e-4).sum())} rays over the bar, worst {err.max():.3e}'
