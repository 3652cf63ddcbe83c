"""Raw-head error of the split-precision MLP modes against the exact fp32-MFMA mode, on rays of the benchmark frame:
python tools/head_error.py <model> [n_rays] [extra modes].  Measurement aid (GPU)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import config as C, scenes
from hyperreel_amd.render import build_render_fn
if os.environ.get('HR_LIB'):
    from hyperreel_amd import lib as _hl
    _hl.LIB_PATH = os.path.abspath(os.environ['HR_LIB'])

name = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
cfg, ds = C.model_config(name), C.dataset_scalars(name)
sd = scenes.make_state_dict(cfg, ds, [64, 64, 64], seed=7, density='dense', app_scale=1.0)
rays = scenes.benchmark_rays(name, 800, 800, frame=7)
r = torch.from_numpy(np.ascontiguousarray(rays[np.random.default_rng(0).choice(rays.shape[0], n, replace=False)])).cuda()
heads = {}
EXTRA = tuple(sys.argv[3:])
for prec in ('fp32', 'bf16x3', 'f16x3', 'f16x2') + EXTRA:
    fn = build_render_fn(cfg, dataset=ds, grid_size=[64, 64, 64], mlp_precision=prec)
    fn.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    heads[prec] = fn.model.render(r, want=('head',))['head'].double()
ref = heads['fp32']; scale = ref.abs().max()
for prec in ('bf16x3', 'f16x3', 'f16x2') + EXTRA:
    d = (heads[prec] - ref).abs()
    live = ref != 0
    print(f'{name} {prec}: max |d head| / max|head| = {float(d.max() / scale):.3e}   rms = {float(d[live].pow(2).mean().sqrt() / scale):.3e}   max|head| = {float(scale):.3f}')
