"""What one rank of an N-GPU strong-scaling run of the 800x800 frame does, measured on ONE GPU: render 1/N of the frame's rays
(contiguous pixel range, as `ShardedFramePipeline` assigns them) as a replayed hipGraph.  The all-gather (7.7 MB in total over
xGMI) is not in it -- no second GPU here; `projected_speedup` = t(1) / t(N) is therefore the ceiling the gather and its
rendezvous can only lower.  python tools/strong_projection.py [--model donerf_sphere]"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import config as C, scenes          # noqa: E402
from hyperreel_amd.render import build_render_fn       # noqa: E402
from hyperreel_amd.parallel import shard_bounds        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--model', default='donerf_sphere')
args = ap.parse_args()
cfg, ds = C.model_config(args.model), C.dataset_scalars(args.model)
sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)          # bench.py's scene
grid = [int(v) for v in sd['model.color_model.net.gridSize']]
fn = build_render_fn(cfg, dataset=ds, grid_size=grid)
fn.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
rays = torch.from_numpy(np.ascontiguousarray(scenes.benchmark_rays(args.model, 800, 800, frame=7))).cuda()
B = rays.shape[0]
out = {'workload': f'{args.model}: one 800x800 frame split over N ranks, the slowest (first) rank\'s share on one MI355X', 'ranks': {}}
t1 = None
for n in (1, 2, 4, 8):
    b = shard_bounds(B, n)
    r = rays[b[0]:b[1]].contiguous()
    rgb = torch.empty((r.shape[0], 3), device='cuda')
    fn.model.reserve(r.shape[0])
    fn.model.render(r, out=rgb); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn.model.render(r, out=rgb)
        torch.cuda.current_stream().synchronize()
        with torch.cuda.graph(g, stream=s):
            fn.model.render(r, out=rgb)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    t1 = t1 or ms
    out['ranks'][str(n)] = {'rays_per_rank': int(r.shape[0]), 'ms_per_frame': round(ms, 4), 'projected_speedup': round(t1 / ms, 2), 'projected_efficiency': round(t1 / ms / n, 3)}
print(json.dumps(out))
