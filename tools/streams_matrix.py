"""The two-stream stress (tools/concurrent_streams_stress.py) over a matrix of settings, one line per cell: which knob moves the rate of
single-ray differences?  python tools/streams_matrix.py [--iters 40]   (HR_LIB selects a measurement build).  GPU box."""
import argparse, itertools, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
if os.environ.get('HR_LIB'):
    from hyperreel_amd import lib as _hl
    _hl.LIB_PATH = os.path.abspath(os.environ['HR_LIB'])
from helpers import Golden
from gpu_common import make_render_fn
ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=40)
ap.add_argument('--case', default='donerf_sphere_small')
ap.add_argument('--one-stream', action='store_true', help='both models on ONE stream (no concurrency at all)')
args = ap.parse_args()
g = Golden(args.case)
rep = max(1, 160000 // g.rays.shape[0])
rays = torch.from_numpy(np.concatenate([g.rays] * rep + [g.rays[:37]], 0)).cuda()
for gd, prec, plan, waves in itertools.product(('fp16', 'fp32'), ('f16x3', 'bf16x3'), (True, False), (8, 4)):
    if not plan and waves == 4:
        continue
    fns = [make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=prec, grid_dtype=gd) for _ in range(2)]
    for f in fns:
        f.model.set_execution(frame_kernel=plan, sample_waves=waves)
    ref = fns[0].model.render(rays)['rgb'].clone()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()] if not args.one_stream else [torch.cuda.Stream()] * 2
    outs = [torch.empty_like(ref), torch.empty_like(ref)]
    bad, seen = 0, {}
    for it in range(args.iters):
        for f, s, o in zip(fns, streams, outs):
            with torch.cuda.stream(s):
                for _ in range(3):
                    f.model.render(rays, out=o)
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                bad += 1
                rr = (o != ref).any(-1).nonzero().flatten().cpu().numpy()
                for r in rr[:4]:
                    seen.setdefault(int(r) % g.rays.shape[0], []).append((int(r) % 8, float((o[r] - ref[r]).abs().max())))
    print(f'{args.case} grid {gd} mlp {prec} plan {"frame" if fns[0].model.frame_kernel_active() else "two"} waves {waves}: {bad} / {2 * args.iters} differ;'
          f' golden rays -> (ray % 8, |d|): { {k: v[:3] for k, v in list(seen.items())[:6]} }', flush=True)
    del fns
