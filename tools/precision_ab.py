"""The MLP arithmetics side by side on ONE full-size frame: python tools/precision_ab.py [--model M] [--frame-kernel] p1 p2 ...
Every precision renders the benchmark frame through a captured hipGraph (timed in alternation) and is compared with the exact-fp32 MLP's image
and raw head.  Measurement aid (GPU box)."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import config as C, scenes
from hyperreel_amd.render import build_render_fn
import bench as B

ap = argparse.ArgumentParser()
ap.add_argument('--model', default='donerf_sphere')
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--frame-kernel', action='store_true')
ap.add_argument('precisions', nargs='+')
args = ap.parse_args()
cfg, ds = C.model_config(args.model), C.dataset_scalars(args.model)
sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
grid = [int(v) for v in sd['model.color_model.net.gridSize']]
rays = torch.from_numpy(scenes.benchmark_rays(args.model, 800, 800, frame=7)).cuda()
n = rays.shape[0]
sub = rays[torch.from_numpy(np.random.default_rng(0).choice(n, 65536, replace=False)).cuda()].contiguous()


def make(prec):
    fn = build_render_fn(cfg, dataset=ds, grid_size=grid, mlp_precision=prec, frame_kernel=bool(args.frame_kernel))
    fn.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    fn.model.native()
    return fn


ref = make('fp32')
ref_img = ref.model.render(rays)['rgb'].clone() if hasattr(ref.model, 'render') else None
ref_head = ref.model.render(sub, want=('head',))['head'].double()
scale = ref_head.abs().max()
vs = []
for p in args.precisions:
    fn = make(p)
    g, out = B.capture(fn.model, rays)
    g.replay(); torch.cuda.synchronize()
    img = out.clone()
    head = fn.model.render(sub, want=('head',))['head'].double()
    d = (img - ref_img).abs()
    vs.append({'precision': p, 'active': fn.model.mlp_precision_active() if hasattr(fn.model, 'mlp_precision_active') else p, 'g': g, 'fn': fn, 'ms': [],
               'image_linf_vs_fp32_mlp': float(d.max()), 'image_rays_over_1e-4': int((d.amax(-1) > 1e-4).sum()), 'nan': int(torch.isnan(img).sum()),
               'head_linf_rel': float((head - ref_head).abs().max() / scale), 'head_rms_rel': float((head - ref_head).pow(2).mean().sqrt() / scale),
               'overflow': int(fn.model.mlp_overflowed()), 'frame_kernel': fn.model.frame_kernel_active()})
for r in range(args.rounds):
    for v in vs:
        dt = B.timed_frames(v['g'].replay, args.steps, 5, False, None)
        v['ms'].append(dt / args.steps * 1e3)
for v in vs:
    v['ms_best'] = round(min(v['ms']), 4)
    v['ms'] = [round(x, 4) for x in v['ms']]
    v.pop('g'); v.pop('fn')
    print(json.dumps(v))
