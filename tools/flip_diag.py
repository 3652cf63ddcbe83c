"""Which decision of the sample stage goes the other way on the rays that a cheaper MLP arithmetic moves by more than 1e-4?
python tools/flip_diag.py [model] [precision].  For each such ray of the 800x800 frame: valid samples (distance > 0) under the exact-fp32 MLP
and under the tested arithmetic, the smallest |distance - near| / |distance - far| margin among its samples under the exact MLP, and the
smallest gap between neighbouring sorted distances.  Measurement aid (GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import config as C, scenes
from hyperreel_amd.render import build_render_fn
name = sys.argv[1] if len(sys.argv) > 1 else 'neural_3d_z_plane'
prec = sys.argv[2] if len(sys.argv) > 2 else 'f16f8'
cfg, ds = C.model_config(name), C.dataset_scalars(name)
sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
grid = [int(v) for v in sd['model.color_model.net.gridSize']]
rays = torch.from_numpy(scenes.benchmark_rays(name, 800, 800, frame=7)).cuda()


def make(p):
    fn = build_render_fn(cfg, dataset=ds, grid_size=grid, mlp_precision=p, frame_kernel=False)
    fn.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return fn.model


ref, tst = make('fp32'), make(prec)
a, b = ref.render(rays)['rgb'], tst.render(rays)['rgb']
err = (a - b).abs().amax(-1)
bad = torch.nonzero(err > 1e-4).flatten()
hc = ref._hc
print(f'{name} {prec}: {bad.numel()} of {rays.shape[0]} rays over 1e-4 (worst {float(err.max()):.3e}); near {hc.near:.6g} far {hc.far:.6g} isect_mask_off {hc.isect_mask_off}')
if bad.numel():
    r = rays[bad].contiguous()
    fa = ref.render(r, want=('distances', 'render_weights', 'points'))
    fb = tst.render(r, want=('distances', 'render_weights', 'points'))
    for i in range(min(bad.numel(), 40)):
        da, db = fa['distances'][i], fb['distances'][i]
        va, vb = int((da > 0).sum()), int((db > 0).sum())
        live = da[da > 0]
        gaps = (live[1:] - live[:-1]).abs()
        wa, wb = int((fa['render_weights'][i] > 0).sum()), int((fb['render_weights'][i] > 0).sum())
        print(f'  ray {int(bad[i])}: err {float(err[bad[i]]):.3e}  samples past the near/far mask {va} -> {vb}, with weight > 0 {wa} -> {wb};'
              f'  min gap between sorted distances {float(gaps.min()) if gaps.numel() else float("nan"):.3e};'
              f'  smallest positive distance {float(da[da > 0].min()):.3e} -> {float(db[db > 0].min()):.3e};'
              f'  largest |d distance| among common ranks {float((da[-min(va, vb):] - db[-min(va, vb):]).abs().max()) if min(va, vb) else 0.0:.3e}')
