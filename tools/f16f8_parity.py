"""Parity of the experimental f16f8 MLP arithmetic (a measurement build: HR_LIB=tools/_bin/libhr_f16f8.so) on everything the
shipped modes are tested on: every reference-made golden / sweep fixture against the 1e-4 bar, then the rays over the bar on the
full-size frames (tests/test_gpu_parity.py's own helpers).  python tools/f16f8_parity.py [mode]  -> gpurun_out/f16f8_parity.txt"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
if os.environ.get('HR_LIB'):
    from hyperreel_amd import lib as _hl
    _hl.LIB_PATH = os.path.abspath(os.environ['HR_LIB'])
from helpers import Golden, golden_cases, sweep_cases
from gpu_common import make_render_fn, render_np
import test_gpu_parity as T

mode = sys.argv[1] if len(sys.argv) > 1 else 'f16f8'
out = open(os.path.join(ROOT, 'gpurun_out', f'{mode}_parity.txt'), 'w')


def say(*a):
    line = ' '.join(str(x) for x in a)
    print(line, flush=True); out.write(line + '\n'); out.flush()


t0 = time.time()
worst, over_cases, n = 0.0, [], 0
for case in golden_cases() + sweep_cases():
    g = Golden(case)
    try:
        fn = make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=mode, iteration=g.iteration)
    except Exception as e:                                     # configurations the split kernels do not take (hidden width != 256)
        say('skip', case, repr(e)[:80]); continue
    err = np.abs(render_np(fn, g.rays)['rgb'] - g.rgb).max(-1)
    n += 1; worst = max(worst, float(err.max()))
    if err.max() > 1e-4: over_cases.append((case, float(err.max()), int((err > 1e-4).sum())))
say(f'{mode}: {n} fixtures, worst L-inf {worst:.3e}, over the bar: {over_cases}  ({time.time() - t0:.0f} s)')
for model in T.FULL_FRAME_MODELS:
    cfg, ds, sd, rays, idx, ref = T._full_frame(model)
    fn = make_render_fn(cfg, ds, sd, mlp_precision=mode)
    rgb = fn.model.render(torch.from_numpy(rays).cuda())['rgb']
    err = np.abs(rgb[torch.from_numpy(idx).cuda()].cpu().numpy() - ref).max(-1)
    say(f'{mode} full frame {model}: {int((err > 1e-4).sum())} of {idx.size} rays over 1e-4, worst {err.max():.3e}  ({time.time() - t0:.0f} s)')
