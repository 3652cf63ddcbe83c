"""Which stage differs when two models render on two streams?  (tools/concurrent_streams_stress.py finds the frames.)
Every render also writes the intermediates (hr_render_fields: head, distances, points, sigma, weights); when an image differs from the one
the model renders alone, the first differing field names the kernel: `head` = the MLP kernel, anything later = the sample kernel.
    python tools/streams_diag.py [--iters 60] [--precision f16x3] [--other same|bf16x3|idle]
GPU box; measurement aid."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import Golden
from gpu_common import make_render_fn
import ctypes as C
from hyperreel_amd import lib as hlib
from hyperreel_amd.plan import hr_fields
L = hlib.load()

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=60)
ap.add_argument('--precision', default='f16x3')
ap.add_argument('--cases', default='config1_random_z16:fp16,immersive_sphere_small:fp32,donerf_sphere_small:fp32')
ap.add_argument('--fields', type=int, default=1)
args = ap.parse_args()
WANT = ('head', 'distances', 'points', 'sigma', 'render_weights')

for spec in args.cases.split(','):
    case, gd = spec.split(':')
    g = Golden(case)
    rep = max(1, 160000 // g.rays.shape[0])
    rays = torch.from_numpy(np.concatenate([g.rays] * rep + [g.rays[:37]], 0)).cuda()
    fns = [make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=args.precision, grid_dtype=gd) for _ in range(2)]
    for f in fns:
        f.model.set_execution(frame_kernel=False)
        f.model.native()
    Z = fns[0].model._hc.z_channels
    ref = {k: v.clone() for k, v in fns[0].model.render(rays, want=WANT).items()}
    ref2 = fns[1].model.render(rays, want=WANT)
    torch.cuda.synchronize()
    assert all(torch.equal(ref[k], ref2[k]) for k in ref), 'the two models differ when alone'
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bufs = [{k: torch.empty_like(v) for k, v in ref.items()} for _ in range(2)]
    bad = 0
    for it in range(args.iters):
        outs = [None, None]
        for i, (f, s) in enumerate(zip(fns, streams)):
            with torch.cuda.stream(s):
                for _ in range(3):
                    if args.fields:         # preallocated outputs, the C entry point directly: nothing but the render kernels on the streams
                        o = bufs[i]
                        fl = hr_fields()
                        for k, slot in (('distances', 'distances_dev'), ('points', 'points_dev'), ('sigma', 'sigma_dev'), ('render_weights', 'weights_dev'), ('head', 'head_dev')):
                            setattr(fl, slot, o[k].data_ptr())
                        hlib.check(L.hr_render_fields(f.model.native(), C.c_void_p(rays.data_ptr()), rays.shape[0], C.c_void_p(o['rgb'].data_ptr()), C.byref(fl),
                                                      C.c_void_p(s.cuda_stream)), 'hr_render_fields')
                        outs[i] = o
                    else:
                        outs[i] = {'rgb': f.model.render(rays, out=bufs[i]['rgb'])['rgb']}
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            diff = {k: (o[k] != ref[k]) for k in o}
            if any(bool(d.any()) for d in diff.values()):
                bad += 1
                msg = [f'{case} {gd} iter {it} model {i}:']
                for k in (WANT + ('rgb',) if args.fields else ('rgb',)):
                    d = diff[k].reshape(diff[k].shape[0], -1)
                    rr = d.any(-1).nonzero().flatten().cpu().numpy()
                    if len(rr) == 0:
                        msg.append(f'  {k}: equal')
                        continue
                    cols = d[rr[0]].nonzero().flatten().cpu().numpy()
                    mag = float((o[k] - ref[k]).abs().reshape(d.shape)[rr[0]].max())
                    msg.append(f'  {k}: {len(rr)} rays differ; rays {rr[:6].tolist()} (ray % 64 = {(rr[:6] % 64).tolist()}, rays per sample wavefront = {64 // max(8, 1 << (Z - 1).bit_length())}); '
                               f'first ray: {len(cols)} of {d.shape[1]} entries, columns {cols[:10].tolist()}, max |d| {mag:.3e}')
                    if k == 'head':
                        a = o[k].reshape(d.shape)[rr[0]][cols[:6]].cpu().numpy(); b = ref[k].reshape(d.shape)[rr[0]][cols[:6]].cpu().numpy()
                        msg.append(f'      got {a.tolist()}  alone {b.tolist()}')
                print('\n'.join(msg), flush=True)
    print(f'{case} {gd} {args.precision}: {bad} renders differed in {args.iters} iterations x 2 models', flush=True)
