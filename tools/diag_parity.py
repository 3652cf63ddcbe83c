"""Where does the HIP image differ from the oracle on the benchmark frame?  python tools/diag_parity.py <model> [n_rays]"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from hyperreel_amd import config as C, scenes
from hyperreel_amd.render import build_render_fn
from hyperreel_oracle import HyperReelOracle

name = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
cfg, ds = C.model_config(name), C.dataset_scalars(name)
sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
grid = [int(v) for v in sd['model.color_model.net.gridSize']]
rays = scenes.benchmark_rays(name, 800, 800, frame=7)
idx = np.random.default_rng(0).choice(rays.shape[0], n, replace=False)
r = np.ascontiguousarray(rays[idx])
for prec in ('bf16x3', 'fp32'):
    fn = build_render_fn(cfg, dataset=ds, grid_size=grid, mlp_precision=prec)
    fn.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    out = fn.model.render(torch.from_numpy(r).cuda(), want=('distances', 'render_weights', 'sigma'))
    got = {k: v.cpu().numpy() for k, v in out.items()}
    ref = HyperReelOracle(cfg, ds, sd).render(r, keep='all')
    e = np.abs(got['rgb'] - ref['rgb']).max(-1)
    bad = np.argsort(-e)[:5]
    print(prec, 'rgb linf', e.max(), 'rays over 1e-4:', int((e > 1e-4).sum()))
    Z = ref['distances'].shape[1]
    for b in bad:
        dr = ref['distances'][b].reshape(Z); dg = got['distances'][b]
        wr = ref['render_weights'][b]; wg = got['render_weights'][b]
        k = int(np.argmax(np.abs(dr - dg)))
        print('  ray', int(idx[b]), 'err %.2e' % e[b], 'max |d dist| %.3e at k=%d (ref %.6f got %.6f)' % (np.abs(dr - dg).max(), k, dr[k], dg[k]),
              'max |d w| %.3e' % np.abs(wr - wg).max(), 'zeros ref/got', int((dr == 0).sum()), int((dg == 0).sum()))
