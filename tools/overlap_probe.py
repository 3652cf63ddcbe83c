"""Can the two kernels of the two-kernel plan overlap across chunks?  The MLP kernel of chunk i + 1 on one stream while the sample kernel of
chunk i runs on another (two model handles = two head workspaces, events between the streams), against the same launches on one stream.
    python tools/overlap_probe.py [--model donerf_sphere] [--chunk 131072]
Measurement aid (GPU box)."""
import argparse, ctypes, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import lib as hlib, config as C, scenes
from hyperreel_amd.render import build_render_fn

ap = argparse.ArgumentParser()
ap.add_argument('--model', default='donerf_sphere')
ap.add_argument('--chunk', type=int, default=131072)
ap.add_argument('--steps', type=int, default=30)
args = ap.parse_args()
cfg, ds = C.model_config(args.model), C.dataset_scalars(args.model)
sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
grid = [int(v) for v in sd['model.color_model.net.gridSize']]
rays = torch.from_numpy(scenes.benchmark_rays(args.model, 800, 800, frame=7)).cuda()
n, rd = rays.shape
L = hlib.load()
fns = []
for _ in range(2):
    fn = build_render_fn(cfg, dataset=ds, grid_size=grid, frame_kernel=False)
    fn.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    fns.append(fn)
h = [fn.model.native() for fn in fns]
ref = fns[0].model.render(rays)['rgb'].clone()
out = torch.empty((n, 3), dtype=torch.float32, device='cuda')
chunks = [(o, min(args.chunk, n - o)) for o in range(0, n, args.chunk)]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
P = lambda t, off: ctypes.c_void_p(t.data_ptr() + off)
S = lambda s: ctypes.c_void_p(s.cuda_stream)


def frame_serial():
    s = torch.cuda.current_stream()
    for i, (o, c) in enumerate(chunks):
        hlib.check(L.hr_stage_mlp(h[0], P(rays, o * rd * 4), c, S(s)), 'mlp')
        hlib.check(L.hr_stage_samples(h[0], P(rays, o * rd * 4), c, P(out, o * 12), S(s)), 'smp')


def frame_overlap():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    done = [None, None]          # the sample kernel that last read workspace k
    for i, (o, c) in enumerate(chunks):
        k = i & 1
        if done[k] is not None:
            sa.wait_event(done[k])
        hlib.check(L.hr_stage_mlp(h[k], P(rays, o * rd * 4), c, S(sa)), 'mlp')
        e = torch.cuda.Event(); e.record(sa)
        sb.wait_event(e)
        hlib.check(L.hr_stage_samples(h[k], P(rays, o * rd * 4), c, P(out, o * 12), S(sb)), 'smp')
        d = torch.cuda.Event(); d.record(sb); done[k] = d
    cur.wait_stream(sa); cur.wait_stream(sb)


def ms(f, n_=args.steps):
    for _ in range(5):
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n_):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_ * 1e3


res = {'model': args.model, 'chunk': args.chunk}
frame_serial(); torch.cuda.synchronize(); res['serial_equal'] = bool(torch.equal(out, ref))
out.zero_(); frame_overlap(); torch.cuda.synchronize(); res['overlap_equal'] = bool(torch.equal(out, ref))
res['serial_ms'] = round(ms(frame_serial), 4)
res['overlap_ms'] = round(ms(frame_overlap), 4)
print(json.dumps(res), flush=True)
os._exit(0)
