"""The co-resident pair ("duo" plan) against the other execution plans on ONE frame, in one process:
    python tools/duo_probe.py [--model donerf_sphere] [--configs "c=3,p=4,w=4;c=3,p=2,w=3"] [--rounds 3] [--steps 20] [--frame-time]
Every configuration renders the frame eagerly and through a captured hipGraph, is compared bit for bit with the two-kernel plan's
image, and is timed in alternation with the library's default plan.  c = sample blocks per CU (0: as many as fit), w = MLP wavefronts per producer (4 | 8 | 3 / 6 = four with a three- / six-slot
weight ring), m = measurement mode (1: one stream, 2: producer only, 3: consumer only).  Measurement aid (GPU box)."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import config as C, scenes
from hyperreel_amd.render import build_render_fn
import bench as B

ap = argparse.ArgumentParser()
ap.add_argument('--model', default='donerf_sphere')
ap.add_argument('--configs', default='c=3,p=4,w=4')
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--precision', default='auto')
ap.add_argument('--grid-dtype', default='fp32')
ap.add_argument('--frame-time', action='store_true', help='hr_render_frame (keyframe nets)')
ap.add_argument('--res', type=int, default=800)
ap.add_argument('--out', default='')
ap.add_argument('--lib', default='', help='a measurement build of the library (tools/build_variant.py)')
args = ap.parse_args()
if args.lib:
    from hyperreel_amd import lib as _hl
    _hl.LIB_PATH = os.path.abspath(args.lib)

cfg, ds = C.model_config(args.model), C.dataset_scalars(args.model)
sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
grid = [int(v) for v in sd['model.color_model.net.gridSize']]
rays_np = scenes.benchmark_rays(args.model, args.res, args.res, frame=7)
rays = torch.from_numpy(rays_np).cuda()
n = rays.shape[0]
ft = float(rays_np[0, -1]) if args.frame_time else None


def make(frame_kernel, duo=None):
    fn = build_render_fn(cfg, dataset=ds, grid_size=grid, mlp_precision=args.precision, grid_dtype=args.grid_dtype, frame_kernel=frame_kernel, duo=duo)
    fn.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    fn.model.native()
    return fn


def render(m):
    return m.render(rays, frame_time=ft)['rgb'] if ft is not None else m.render(rays)['rgb']


def duo_times(m):
    import ctypes
    from hyperreel_amd import lib as hl
    tt = (ctypes.c_uint64 * 8)()
    hl.check(hl.load().hr_debug_duo_times(m.native(), tt), 'hr_debug_duo_times')
    t0 = min(x for x in (tt[0], tt[2]) if x) if (tt[0] or tt[2]) else 0
    return {'producer': [round((tt[0] - t0) / 100, 1), round((tt[1] - t0) / 100, 1)] if tt[0] else None,
            'consumer': [round((tt[2] - t0) / 100, 1), round((tt[3] - t0) / 100, 1)] if tt[2] else None,
            'clock_ghz': [round(tt[4] / tt[5] / 10, 3) if tt[5] else None, round(tt[6] / tt[7] / 10, 3) if tt[7] else None]}


base2 = make(False)
ref = render(base2.model).clone()
torch.cuda.synchronize()
variants = []
fdef = make(True)
g, out = B.capture(fdef.model, rays, frame_time=ft)
g.replay(); torch.cuda.synchronize()
variants.append({'name': 'default:' + fdef.model.plan_active(), 'graph': g, 'out': out, 'fn': fdef, 'ms': [], 'eager_equal': bool(torch.equal(render(fdef.model), ref))})
g2, out2 = B.capture(base2.model, rays, frame_time=ft)
variants.append({'name': 'two_kernels', 'graph': g2, 'out': out2, 'fn': base2, 'ms': [], 'eager_equal': True})
for spec in args.configs.split(';'):
    kv = dict(x.split('=') for x in spec.split(','))
    duo = {'consumers': int(kv.get('c', 0)), 'mlp_waves': int(kv.get('w', 0)), 'mode': int(kv.get('m', 0))}
    mode = duo.pop('mode')
    fn = make('duo', duo)
    m = fn.model
    if mode:
        if mode == 3:
            render(m); torch.cuda.synchronize()      # leaves the head and the flags of a complete pair behind
        m.set_execution(duo={'mode': mode})
    img = render(m).clone()
    torch.cuda.synchronize()
    eq = bool(torch.equal(img, ref))
    linf = float((img - ref).abs().max())
    fault = m.plan_faulted()
    times_us = duo_times(m)
    v = {'name': 'duo:' + spec, 'eager_times_us': times_us, 'fn': fn, 'ms': [], 'eager_equal': eq, 'eager_linf': linf, 'fault_eager': fault, 'plan': m.plan_active()}
    print(json.dumps({k: v[k] for k in ('name', 'eager_equal', 'eager_linf', 'fault_eager', 'plan', 'eager_times_us')}), flush=True)
    if fault:
        variants.append(v); continue
    g, out = B.capture(m, rays, frame_time=ft)
    g.replay(); torch.cuda.synchronize()
    v['graph'], v['out'] = g, out
    v['fault_graph'] = m.plan_faulted()
    variants.append(v)
for v in variants:
    v['ms_eager'] = []
for r in range(args.rounds):
    for v in variants:
        if 'graph' not in v or v.get('fault_graph'):
            continue
        dt = B.timed_frames(v['graph'].replay, args.steps, 3, False, None)
        v['ms'].append(dt / args.steps * 1e3)
        m_ = v['fn'].model
        dt = B.timed_frames(lambda: render(m_), args.steps, 3, False, None)      # the same frame as eager launches
        v['ms_eager'].append(dt / args.steps * 1e3)
res = []
for v in variants:
    r = {k: v[k] for k in v if k not in ('graph', 'out', 'fn', 'ms', 'ms_eager')}
    if v['ms_eager']:
        r['ms_eager_best'] = round(min(v['ms_eager']), 4)
    if v['ms']:
        r['ms_best'] = round(min(v['ms']), 4)
        r['ms_all'] = [round(x, 4) for x in v['ms']]
        r['mrays_s'] = round(n / min(v['ms']) / 1e3, 1)
        r['graph_equal'] = bool(torch.equal(v['out'], ref))
        r['fault_after'] = v['fn'].model.plan_faulted()
        if v['name'].startswith('duo'):
            r['steady_times_us'] = duo_times(v['fn'].model)
    res.append(r)
    print(json.dumps(r), flush=True)
if args.out:
    json.dump({'model': args.model, 'rays': n, 'results': res}, open(args.out, 'w'), indent=1)
sys.stdout.flush()
os._exit(0)
