"""Narrowing a rare single-ray difference of the co-resident pair (config1_random_z16, f16x3, float16 texels).  GPU box."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import Golden
from gpu_common import make_render_fn
import argparse
ap = argparse.ArgumentParser(); ap.add_argument('--lib', default=''); A = ap.parse_args()
if A.lib:
    from hyperreel_amd import lib as _hl
    _hl.LIB_PATH = os.path.abspath(A.lib)
case, prec, gd = 'config1_random_z16', 'f16x3', 'fp16'
g = Golden(case)
fn = make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=prec, grid_dtype=gd)
rays = torch.from_numpy(np.concatenate([g.rays] * 40 + [g.rays[:37]], 0)).cuda()
fn.model.set_execution(frame_kernel=False)
two = fn.model.render(rays)['rgb'].clone()
torch.cuda.synchronize()
# (0) is the two-kernel plan itself reproducible, alone and beside a busy stream?
side = torch.cuda.Stream()
junk = torch.empty(64 << 20, device='cuda')
bad = 0
for it in range(40):
    if it >= 20:
        with torch.cuda.stream(side):
            for _ in range(20):
                junk.add_(1.0)
    out = fn.model.render(rays)['rgb']
    torch.cuda.synchronize()
    bad += int(not torch.equal(out, two))
print('two-kernel plan repeated (20 alone, 20 beside a busy stream): differing runs', bad, flush=True)
for name, duo in (('default', {}), ('c=2', {'consumers': 2}), ('c=1', {'consumers': 1}), ('consumer only', {'mode': 3}), ('consumer only c=2', {'mode': 3, 'consumers': 2}), ('consumer only c=1', {'mode': 3, 'consumers': 1}), ('serial', {'mode': 1})):
    fn.model.set_execution(frame_kernel='duo', duo={'consumers': 0, 'mlp_waves': 0, 'mode': 0, **duo})
    bad, where = 0, []
    for it in range(40):
        out = torch.full_like(two, float('nan'))
        fn.model.render(rays, out=out)
        torch.cuda.synchronize()
        d = (out != two).any(-1)
        if bool(d.any()):
            bad += 1
            idx = torch.nonzero(d)[:, 0]
            i = int(idx[0])
            where.append((i // 64, i % 64, [f'{float(v):.9g}' for v in out[i]], [f'{float(v):.9g}' for v in two[i]]))
    print(f'duo {name}: bad runs {bad} / 40', where[:4], flush=True)
