"""Repeated renders through the co-resident pair against the two-kernel image; prints WHERE rays differ (tile, row in tile, block).
python tools/duo_stress.py [case ...]   Measurement / debugging aid (GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import Golden
from gpu_common import make_render_fn
cases = sys.argv[1:] or ['config1_random_z16', 'donerf_sphere_small', 'technicolor_z_plane_small', 'neural_3d_z_plane_small']
for case in cases:
    for prec in ('f16x3', 'bf16x3'):
        for gd in ('fp32', 'fp16'):
            g = Golden(case)
            fn = make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=prec, grid_dtype=gd)
            rays = torch.from_numpy(np.concatenate([g.rays] * 40 + [g.rays[:37]], 0)).cuda()
            fn.model.set_execution(frame_kernel=False)
            two = fn.model.render(rays)['rgb'].clone()
            torch.cuda.synchronize()
            Z = g.cfg['embedding']['embeddings']['ray_prediction_0']['z_channels']
            zp = 8
            while zp < Z: zp *= 2
            rpb = 256 // zp
            bad_runs = 0
            for mode in (0, 1):
                fn.model.set_execution(frame_kernel='duo', duo={'mode': mode})
                for it in range(25):
                    out = torch.full_like(two, float('nan'))
                    fn.model.render(rays, out=out)
                    torch.cuda.synchronize()
                    d = (out != two).any(-1) | torch.isnan(out).any(-1)
                    if bool(d.any()):
                        idx = torch.nonzero(d)[:, 0].cpu().numpy()
                        bad_runs += 1
                        print(f'  {case} {prec} {gd} mode {mode} run {it}: {idx.size} rays differ; tiles {sorted(set((idx // 64).tolist()))[:8]} rows-in-tile {sorted(set((idx % 64).tolist()))[:20]} '
                              f'blocks-in-tile {sorted(set(((idx % 64) // rpb).tolist()))} nan {int(torch.isnan(out).any(-1).sum())} max|d| {float((out - two).abs().nan_to_num(9).max()):.3e}', flush=True)
            print(f'{case} {prec} {gd}: rays {rays.shape[0]}, RPB {rpb}, bad runs {bad_runs} / 50, fault {fn.model.plan_faulted()}', flush=True)
