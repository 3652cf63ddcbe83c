"""How wide must the decision bands of the verified fast path be, and how many rays would they flag?  
For each family's full 800x800 frame: render the intermediates (hr_render_fields) with the exact arithmetic's stand-in (f16x3) and with the fast
arithmetic (f16f8 by default); report
  * the distribution of |d distance|, |d point|, |d weight| between the two over samples whose discrete decisions agree  -> the band must cover it;
  * rays whose rgb differs by more than 1e-4 (the flips) and the smallest margin-multiple that would have flagged each of them;
  * for a ladder of band widths: the fraction of rays with at least one sample inside a band of near / far, an aabb face or the weight threshold.
GPU box; measurement aid."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import config as C, scenes
from hyperreel_amd.render import build_render_fn

ap = argparse.ArgumentParser()
ap.add_argument('--fast', default='f16f8')
ap.add_argument('--models', default='donerf_sphere,technicolor_z_plane,neural_3d_z_plane,immersive_sphere')
ap.add_argument('--out', default='')
args = ap.parse_args()
WANT = ('distances', 'points', 'render_weights')
res = {}
for name in args.models.split(','):
    cfg, ds = C.model_config(name), C.dataset_scalars(name)
    sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
    grid = [int(v) for v in sd['model.color_model.net.gridSize']]
    rays = torch.from_numpy(scenes.benchmark_rays(name, 800, 800, frame=7)).cuda()
    outs = {}
    for prec in ('f16x3', args.fast):
        f = build_render_fn(cfg, dataset=ds, grid_size=grid, mlp_precision=prec)
        f.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        o = {}
        for r0 in range(0, rays.shape[0], 160000):           # fields of a whole frame at once are several GB
            part = f.model.render(rays[r0:r0 + 160000].contiguous(), want=WANT)
            for k, v in part.items():
                o.setdefault(k, []).append(v)
        outs[prec] = {k: torch.cat(v, 0) for k, v in o.items()}
        hc = f.model._hc
        near, far, wthr = float(hc.near), float(hc.far), float(hc.weight_thresh)
        aabb = [float(hc.aabb[i]) for i in range(6)]
        del f
        torch.cuda.empty_cache()
    a, b = outs['f16x3'], outs[args.fast]
    live_a, live_b = a['distances'] > 0, b['distances'] > 0
    same = live_a == live_b
    dd = (a['distances'] - b['distances']).abs()[same & live_a]
    dp = (a['points'] - b['points']).abs().amax(-1)[same & live_a]
    dw = (a['render_weights'] - b['render_weights']).abs()[same & live_a]
    drgb = (a['rgb'] - b['rgb']).abs().amax(-1)
    q = lambda t: [float(torch.quantile(t[:2000000].float(), x)) for x in (0.5, 0.99, 0.9999)] + [float(t.max())]
    r = {'near': near, 'far': far, 'weight_thresh': wthr, 'samples_mask_disagree': int((~same).sum()), 'rays_over_1e-4': int((drgb > 1e-4).sum()),
         'rgb_linf': float(drgb.max()), 'd_distance_p50_p99_p9999_max': q(dd), 'd_point_p50_p99_p9999_max': q(dp), 'd_weight_p50_p99_p9999_max': q(dw), 'bands': {}}
    d = a['distances']; p = a['points']; w = a['render_weights']
    face = torch.minimum((p - torch.tensor(aabb[:3], device='cuda')).abs().amin(-1), (p - torch.tensor(aabb[3:], device='cuda')).abs().amin(-1))
    scale_d = max(abs(near), abs(far) if np.isfinite(far) else 0.0, 1.0)
    for eps in (1e-5, 3e-5, 1e-4, 3e-4, 1e-3):
        # masked samples carry distance 0: count the live side of near / far and double it (the masked side is as populated)
        nf = live_a & (((d - near).abs() < eps * scale_d) | ((d - far).abs() < eps * scale_d))
        bx = live_a & (face < eps * max(max(abs(x) for x in aabb), 1.0))
        wt = (w - wthr).abs() < eps * 0.1
        any_ray = (nf | bx | wt).any(-1)
        r['bands'][str(eps)] = {'near_far_rays': float(nf.any(-1).float().mean()) * 2, 'aabb_rays': float(bx.any(-1).float().mean()) * 2, 'weight_rays': float(wt.any(-1).float().mean()),
                                'any_rays_lower_bound': float(any_ray.float().mean())}
    # the flipped rays: how close was their closest decision?
    bad = (drgb > 1e-4).nonzero().flatten()[:16]
    r['flips'] = [{'ray': int(i), 'drgb': float(drgb[i]), 'min_near_far_margin': float(torch.minimum((d[i] - near).abs(), (d[i] - (far if np.isfinite(far) else 1e30)).abs())[live_a[i]].min()) if bool(live_a[i].any()) else None,
                   'min_face_margin': float(face[i][live_a[i]].min()) if bool(live_a[i].any()) else None, 'min_weight_margin': float((w[i] - wthr).abs().min()),
                   'mask_disagree_samples': int((live_a[i] != live_b[i]).sum())} for i in bad]
    res[name] = r
    print(name, json.dumps(r), flush=True)
if args.out:
    json.dump(res, open(args.out, 'w'), indent=1)
