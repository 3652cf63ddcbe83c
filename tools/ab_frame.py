"""A/B/C... of measurement builds of the library on ONE frame, in one process: python tools/ab_frame.py [--model M] [--rounds R]
[--two-kernel] [--precision P] name=path.so ...   ("base" = the in-tree library).  Every variant renders the same frame through a
captured hipGraph; the variants are timed in alternation (R rounds) and every image is compared bit for bit with the first variant's.
Measurement aid (GPU box); the product never loads anything from tools/_bin."""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import lib as hl
from hyperreel_amd import config as C, scenes
import bench as B

ap = argparse.ArgumentParser()
ap.add_argument('--model', default='donerf_sphere')
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--two-kernel', action='store_true')
ap.add_argument('--frame-mode', type=int, default=1, help='HR_OPT_FRAME_KERNEL: 1 = where faster, 2 = wherever it fits')
ap.add_argument('--precision', default='auto')
ap.add_argument('--grid-dtype', default='fp32')
ap.add_argument('--sample-waves', type=int, default=0)
ap.add_argument('--out', default='')
ap.add_argument('libs', nargs='+')
args = ap.parse_args()

from hyperreel_amd.render import build_render_fn
cfg, ds = C.model_config(args.model), C.dataset_scalars(args.model)
sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
grid = [int(v) for v in sd['model.color_model.net.gridSize']]
rays = torch.from_numpy(scenes.benchmark_rays(args.model, 800, 800, frame=7)).cuda()
n = rays.shape[0]
variants = []
ref = None
for spec in args.libs:
    name, _, path = spec.partition('=')
    path = hl.LIB_PATH if (name == 'base' and not path) else os.path.abspath(path)
    default_path = hl.LIB_PATH
    hl._lib = None
    hl.LIB_PATH = path
    fn = build_render_fn(cfg, dataset=ds, grid_size=grid, mlp_precision=args.precision, grid_dtype=args.grid_dtype,
                         frame_kernel=(False if args.two_kernel else (2 if args.frame_mode == 2 else True)), sample_waves=args.sample_waves or None)
    fn.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    fn.model.native()
    g, out = B.capture(fn.model, rays)
    g.replay(); torch.cuda.synchronize()
    img = out.clone()
    if ref is None:
        ref = img
    same = bool(torch.equal(img, ref))
    linf = float((img - ref).abs().max())
    variants.append({'name': name, 'graph': g, 'fn': fn, 'lib': hl._lib, 'same': same, 'linf': linf, 'ms': [], 'frame_kernel': fn.model.frame_kernel_active()})
    hl.LIB_PATH = default_path
for r in range(args.rounds):
    for v in variants:
        dt = B.timed_frames(v['graph'].replay, args.steps, 5, False, None)
        v['ms'].append(dt / args.steps * 1e3)
res = []
for v in variants:
    ms = min(v['ms'])
    res.append({'name': v['name'], 'ms_best': round(ms, 4), 'ms_all': [round(x, 4) for x in v['ms']], 'mrays_s': round(n / ms / 1e3, 1),
                'bit_identical_to_first': v['same'], 'linf_vs_first': v['linf'], 'frame_kernel': v['frame_kernel']})
    print(json.dumps(res[-1]), flush=True)
if args.out:
    json.dump({'model': args.model, 'precision': args.precision, 'two_kernel': args.two_kernel, 'results': res}, open(args.out, 'w'), indent=1)
sys.stdout.flush()
os._exit(0)      # the models were created by different copies of the library: skip the destructors
