#!/usr/bin/env python
"""Per-wave phase timeline of the split-precision MLP kernel (hr_debug_trace_mlp)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hyperreel_amd import config as C, scenes, lib
from hyperreel_amd.render import build_render_fn
if os.environ.get('HR_LIB'):
    lib.LIB_PATH = os.path.abspath(os.environ['HR_LIB'])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
cfg, ds = C.model_config('donerf_sphere'), C.dataset_scalars('donerf')
sd = scenes.make_state_dict(cfg, ds, [64, 64, 64], seed=7)
fn = build_render_fn(cfg, dataset=ds, grid_size=[64, 64, 64], mlp_precision=os.environ.get('HR_PREC', 'auto'), frame_kernel=False)
fn.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
rays = torch.from_numpy(scenes.benchmark_rays('donerf_sphere', 800, 800)[:n]).cuda()
fn.model.reserve(max(n, 4096)); h = fn.model.native(); L = lib.load()
tile = 128 if os.environ.get('HR_MLP_TILE') == '128' else 64
nwg = (n + tile - 1) // tile
tr = torch.zeros((nwg * 4, 64), dtype=torch.int64, device='cuda')
for _ in range(3):
    lib.check(L.hr_debug_trace_mlp(h, ctypes.c_void_p(rays.data_ptr()), n, ctypes.c_void_p(tr.data_ptr()), None), 'trace')
torch.cuda.synchronize()
t = tr.cpu().numpy().astype(np.int64)
ns = int((t[0] != 0).sum())
t = t[:, :ns]
d = np.diff(t, axis=1)
names = ['prologue'] + sum([[f'L{l} gemm', f'L{l} barrier'] + ([f'L{l} first bias loads', f'L{l} epilogue (own, rest)', f'L{l} wait at barrier'] if os.environ.get('HR_FINE') else [f'L{l} epilogue+bar']) for l in range(5)], [])
names += ['last p0 gemm', 'last p0 store', 'last p1 gemm', 'last p1 store']
print(f'{nwg} workgroups of {tile} rays; stamps per wave {ns}; cycles (s_memtime, 100 MHz?) mean/median over waves')
tot = (t[:, -1] - t[:, 0])
print(f'wave lifetime mean {tot.mean():.0f} median {np.median(tot):.0f}  kernel span {t[:, -1].max() - t[:, 0].min()}')
for i in range(d.shape[1]):
    nm = names[i] if i < len(names) else f'phase {i}'
    print(f'  {nm:22s} mean {d[:, i].mean():9.0f}  median {np.median(d[:, i]):9.0f}  max {d[:, i].max():9.0f}   {100 * d[:, i].mean() / tot.mean():5.1f} %')
