#!/usr/bin/env python
"""Per-wave chunk timeline of the register-resident MLP kernel (csrc/mlp_reg_impl.inc, an experiment that only measurement
builds contain): shader cycles between the chunk boundaries of the LAST tile each workgroup processed.
    python tools/build_variant.py reg -DHR_WITH_REG_KERNEL
    HR_LIB=tools/_bin/libhr_reg.so python tools/mlp_reg_trace.py [n_rays] [precision]
(+ -DHR_REG_TRACE_FINE and HR_REG_TRACE_FINE=1: compute / DMA wait / barrier per chunk.)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hyperreel_amd import config as C, scenes, lib
from hyperreel_amd.render import build_render_fn
if os.environ.get('HR_LIB'):
    lib.LIB_PATH = os.path.abspath(os.environ['HR_LIB'])        # a measurement variant (tools/build_variant.py)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
prec = sys.argv[2] if len(sys.argv) > 2 else 'auto'
cfg, ds = C.model_config('donerf_sphere'), C.dataset_scalars('donerf')
sd = scenes.make_state_dict(cfg, ds, [64, 64, 64], seed=7)
fn = build_render_fn(cfg, dataset=ds, grid_size=[64, 64, 64], mlp_precision=prec, frame_kernel=False)
fn.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
rays = torch.from_numpy(scenes.benchmark_rays('donerf_sphere', 800, 800)[:n]).cuda()
fn.model.set_execution(mlp_kernel='registers')
fn.model.reserve(n); h = fn.model.native(); L = lib.load()
assert fn.model.mlp_kernel_active()
nwg = min(256, (n + 127) // 128)
tr = torch.zeros((nwg * 4, 192), dtype=torch.int64, device='cuda')
for _ in range(3):
    lib.check(L.hr_debug_trace_mlp(h, ctypes.c_void_p(rays.data_ptr()), n, ctypes.c_void_p(tr.data_ptr()), None), 'trace')
torch.cuda.synchronize()
t = tr.cpu().numpy().astype(np.int64)
ns = int((t[0] != 0).sum())
t = t[:, :ns]
d = np.diff(t, axis=1)
tot = t[:, -1] - t[:, 0]
print(f'{nwg} workgroups; {ns} stamps per wave; tile lifetime mean {tot.mean():.0f} median {np.median(tot):.0f} cycles')
names = ['prologue', 'L0 c0', 'L0 c1']
for l in range(1, 5):
    for p in range(4):
        names += [f'L{l} p{p} h0', f'L{l} p{p} h1']
names += [f'last p{i // 2} h{i % 2}' for i in range(40)]
if os.environ.get('HR_REG_TRACE_FINE'):
    # stamps: 0 start, 1 prologue, then per chunk (before vmcnt wait, after it, after the barrier)
    dd = d[:, 1:]
    k = dd.shape[1] // 3
    dd = dd[:, :3 * k].reshape(-1, k, 3)
    print('per chunk (mean over waves): compute / wait DMA / barrier')
    for c in range(k):
        print(f'  chunk {c:2d}: {dd[:, c, 0].mean():7.0f} {dd[:, c, 1].mean():7.0f} {dd[:, c, 2].mean():7.0f}')
    print('  totals : %.0f %.0f %.0f' % (dd[:, :, 0].sum(1).mean(), dd[:, :, 1].sum(1).mean(), dd[:, :, 2].sum(1).mean()))
    sys.exit(0)
for i in range(d.shape[1]):
    nm = names[i] if i < len(names) else f'phase {i}'
    print(f'  {nm:14s} mean {d[:, i].mean():8.0f}  median {np.median(d[:, i]):8.0f}  max {d[:, i].max():8.0f}  {100 * d[:, i].mean() / tot.mean():5.1f} %')
