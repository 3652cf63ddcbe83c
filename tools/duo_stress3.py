"""Which configurations show the rare single-ray difference of the co-resident pair?  40 runs each at the most sensitive setting found
(two sample blocks per CU beside the producer).  GPU box."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import Golden
from gpu_common import make_render_fn
for case in ('config1_random_z16', 'donerf_sphere_small', 'technicolor_z_plane_small', 'neural_3d_z_plane_small', 'immersive_sphere_small'):
    for prec, gd in (('auto', 'fp32'), ('auto', 'fp16'), ('bf16x3', 'fp16')):
        g = Golden(case)
        fn = make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=prec, grid_dtype=gd)
        rep = max(1, 160000 // g.rays.shape[0])
        rays = torch.from_numpy(np.concatenate([g.rays] * rep + [g.rays[:37]], 0)).cuda()
        fn.model.set_execution(frame_kernel=False)
        two = fn.model.render(rays)['rgb'].clone()
        torch.cuda.synchronize()
        res = {}
        for name, duo in (('c=2', {'consumers': 2}), ('c=0', {'consumers': 0})):
            fn.model.set_execution(frame_kernel='duo', duo={'mlp_waves': 0, 'mode': 0, **duo})
            bad, rows = 0, set()
            for it in range(40):
                out = torch.full_like(two, float('nan'))
                fn.model.render(rays, out=out)
                torch.cuda.synchronize()
                d = (out != two).any(-1)
                if bool(d.any()):
                    bad += 1
                    rows.update((torch.nonzero(d)[:, 0] % 64).cpu().tolist())
            res[name] = (bad, sorted(rows)[:12])
        print(case, fn.model.mlp_precision_active(), gd, rays.shape[0], res, flush=True)
