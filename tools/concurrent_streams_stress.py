"""Two models rendering on two streams at the same time (two-kernel plan: MLP kernels of one beside sample kernels of the other on the
same CUs): is every image still the one the model renders alone?  GPU box."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import Golden
from gpu_common import make_render_fn
if os.environ.get('HR_LIB'):          # a measurement build (tools/build_variant.py)
    from hyperreel_amd import lib as _hl
    _hl.LIB_PATH = os.path.abspath(os.environ['HR_LIB'])
for case, gd in (('immersive_sphere_small', 'fp32'), ('donerf_sphere_small', 'fp16'), ('config1_random_z16', 'fp16')):
    g = Golden(case)
    rep = max(1, 160000 // g.rays.shape[0])
    rays = torch.from_numpy(np.concatenate([g.rays] * rep + [g.rays[:37]], 0)).cuda()
    for plan in (False, True):
        fns = [make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision='f16x3', grid_dtype=gd) for _ in range(2)]
        for f in fns:
            f.model.set_execution(frame_kernel=plan)
        ref = fns[0].model.render(rays)['rgb'].clone()
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [torch.empty_like(ref), torch.empty_like(ref)]
        bad = 0
        for it in range(40):
            for f, s, o in zip(fns, streams, outs):
                with torch.cuda.stream(s):
                    for _ in range(3):
                        f.model.render(rays, out=o)
            torch.cuda.synchronize()
            bad += int(not (torch.equal(outs[0], ref) and torch.equal(outs[1], ref)))
            for i, o in enumerate(outs):
                if not torch.equal(o, ref):
                    rr = (o != ref).any(-1).nonzero().flatten().cpu().numpy()
                    print(f'    iter {it} model {i}: {len(rr)} rays differ: {rr[:8].tolist()} (ray % 8 = {(rr[:8] % 8).tolist()}), max |d| {float((o - ref).abs().max()):.3e}, '
                          f'got {o[rr[0]].cpu().numpy().tolist()} alone {ref[rr[0]].cpu().numpy().tolist()}', flush=True)
        print(case, gd, 'plan', 'frame_kernel' if fns[0].model.frame_kernel_active() else 'two_kernels', 'runs with a differing image:', bad, '/ 40', flush=True)
