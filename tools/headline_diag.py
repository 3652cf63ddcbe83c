"""Why did the driver's 20-frame window read 3.08 ms?  python tools/headline_diag.py [--out F.json]

Reproduces bench.py's headline conditions (the first GPU work of the process: build -> capture -> 5 warm-up replays -> 20 timed
replays of the DoNeRF 800x800 frame) with a HIP event pair around EVERY replay, for both execution plans:

  cold_series     per-replay ms of the first 25 replays of the default plan, in order (clock ramp / first-touch shows as a slope)
  ab              200 replays of each plan, interleaved A,B,A,B: {min, p50, p90, p99, max}
  windows         ten consecutive bench.py-style windows (wall clock around 20 replays) of each plan, alternating
  after_idle      a 20-replay window after the GPU sat idle for 0.5 s (does the clock fall back?)

Measurement aid for the GPU box; the product never imports it."""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from hyperreel_amd import config as C, scenes
from hyperreel_amd.render import build_render_fn

ap = argparse.ArgumentParser()
ap.add_argument('--out', default='')
ap.add_argument('--model', default='donerf_sphere')
ap.add_argument('--n', type=int, default=200)
args = ap.parse_args()

t_proc = time.perf_counter()
cfg, ds = C.model_config(args.model), C.dataset_scalars(args.model)
sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
grid = [int(v) for v in sd['model.color_model.net.gridSize']]
rays = torch.from_numpy(scenes.benchmark_rays(args.model, 800, 800, frame=7)).cuda()
n_rays = rays.shape[0]


def make(frame_kernel):
    f = build_render_fn(cfg, dataset=ds, grid_size=grid, frame_kernel=frame_kernel)
    f.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    f.model.native()
    return f


def series(replay, n):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(); replay(); b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    return [round(a.elapsed_time(b), 4) for a, b in ev], round(wall, 4)


def pct(x):
    x = np.sort(np.asarray(x))
    return {'min': float(x[0]), 'p50': float(np.percentile(x, 50)), 'p90': float(np.percentile(x, 90)), 'p99': float(np.percentile(x, 99)),
            'max': float(x[-1]), 'mean': round(float(x.mean()), 4)}


res = {'model': args.model}
fa = make(True)
ga, out_a = B.capture(fa.model, rays)
res['s_before_first_replay'] = round(time.perf_counter() - t_proc, 2)
cold, wall = series(ga.replay, 25)
res['cold_series_default_plan'] = {'ms': cold, 'wall_ms_per_replay': wall, 'frame_kernel': bool(fa.model.frame_kernel_active())}
# bench.py's own window right after, for the record
res['bench_window_default_plan_ms'] = round(B.timed_frames(ga.replay, 20, 5, False, None) / 20 * 1e3, 4)

fb = make(False)
gb, out_b = B.capture(fb.model, rays)
cold_b, wall_b = series(gb.replay, 25)
res['cold_series_two_kernel'] = {'ms': cold_b, 'wall_ms_per_replay': wall_b}
res['bit_identical'] = bool(torch.equal(out_a, out_b))

# interleaved A/B, one event pair per replay
ea, eb = [], []
for i in range(args.n):
    for g, acc in ((ga, ea), (gb, eb)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        acc.append((a, b))
torch.cuda.synchronize()
res['ab'] = {'frame_kernel': pct([a.elapsed_time(b) for a, b in ea]), 'two_kernel': pct([a.elapsed_time(b) for a, b in eb])}

wa, wb = [], []
for i in range(10):
    wa.append(round(B.timed_frames(ga.replay, 20, 5, False, None) / 20 * 1e3, 4))
    wb.append(round(B.timed_frames(gb.replay, 20, 5, False, None) / 20 * 1e3, 4))
res['windows'] = {'frame_kernel': wa, 'two_kernel': wb}

idle = {}
for name, g in (('frame_kernel', ga), ('two_kernel', gb)):
    xs = []
    for _ in range(4):
        time.sleep(0.5)
        s, _w = series(g.replay, 20)
        xs.append(s)
    idle[name] = {'first_replay_ms': [s[0] for s in xs], 'mean_ms': [round(float(np.mean(s)), 4) for s in xs],
                  'no_warmup_window_ms': None}
    time.sleep(0.5)
    idle[name]['no_warmup_window_ms'] = round(B.timed_frames(g.replay, 20, 0, False, None) / 20 * 1e3, 4)
res['after_idle_0.5s'] = idle
line = json.dumps(res)
print(line)
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, 'w').write(line + '\n')
