"""Which side differs when a frame-kernel render differs from itself?  (the open run-to-run difference.)
A -DHR_DEBUG_HSUM build makes the sample stage write, per ray, the XOR of the bits of every head value it READ and of its sorted distances.
The same rays are rendered again and again on one stream; when an image differs from the first, the checksums of the differing ray say whether
the HEAD it read differed (MLP side / hand-over) or the head was the same and the sample stage computed something else from it.
    HR_LIB=tools/_bin/libhr_hsum.so python tools/hsum_bisect.py [--iters 200] [--waves 8] [--plan frame|two]
GPU box; measurement aid."""
import argparse, ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from hyperreel_amd import lib as hl
hl.LIB_PATH = os.path.abspath(os.environ['HR_LIB'])
from helpers import Golden
from gpu_common import make_render_fn
ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=200)
ap.add_argument('--waves', type=int, default=8)
ap.add_argument('--plan', default='frame')
ap.add_argument('--case', default='donerf_sphere_small')
ap.add_argument('--precision', default='f16x3')
ap.add_argument('--lanes-mode', type=int, default=2, help='2: -DHR_DEBUG_HSUM=2 (products), 3: -DHR_DEBUG_HSUM=3 (the eight intermediates)')
ap.add_argument('--lanes', action='store_true', help='a -DHR_DEBUG_HSUM=2 build: per-lane raw values of the six products ro * s, rd * s and of o.o, o.d')
args = ap.parse_args()
g = Golden(args.case)
rep = max(1, 160000 // g.rays.shape[0])
rays = torch.from_numpy(np.concatenate([g.rays] * rep + [g.rays[:37]], 0)).cuda()
fns = [make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=args.precision) for _ in range(2)]
for f in fns:
    f.model.set_execution(frame_kernel=(args.plan == 'frame'), sample_waves=args.waves)
    f.model.native()
L = hl.load()
L.hr_debug_set_hsum.argtypes = [ctypes.c_void_p]
L.hr_debug_set_hsum.restype = None
n = rays.shape[0]
hs = torch.zeros((n, 280 if args.lanes else 16), dtype=torch.int32, device='cuda')
L.hr_debug_set_hsum(ctypes.c_void_p(hs.data_ptr()))
ref = fns[0].model.render(rays)['rgb'].clone()
torch.cuda.synchronize()
ref_hs = hs.clone()
print('plan', 'frame kernel' if fns[0].model.frame_kernel_active() else 'two kernels', 'waves', args.waves, flush=True)
out = torch.empty_like(ref)
bad = 0
mism_total = 0
rows = []
firsts = []
lane_cols = []
for it in range(args.iters):
    for f in fns:
        for _ in range(3):
            hs.zero_()
            f.model.render(rays, out=out)
            torch.cuda.synchronize()
            mism_total += int(hs[:, 6].sum())
            if not torch.equal(out, ref):
                bad += 1
                rr = (out != ref).any(-1).nonzero().flatten()
                r = int(rr[0])
                rows += [int(x) % 64 for x in rr.cpu().numpy()]
                print(f'   head as read BEFORE the distance {"DIFFERS" if int(hs[r, 4]) != int(ref_hs[r, 4]) else "same"}; ray / anchors {"DIFFER" if int(hs[r, 5]) != int(ref_hs[r, 5]) else "same"}')
                print(f'iter {it}: {len(rr)} rays differ; ray {r} (ray % 8 = {r % 8}), |d| {float((out[r] - ref[r]).abs().max()):.3e}; head checksum {"DIFFERS" if int(hs[r, 0]) != int(ref_hs[r, 0]) else "same"}, '
                      f'distance before the sort {"DIFFERS" if int(hs[r, 2]) != int(ref_hs[r, 2]) else "same"}, after the sort {"DIFFERS" if int(hs[r, 3]) != int(ref_hs[r, 3]) else "same"}, '
                      f'final distance {"DIFFERS" if int(hs[r, 1]) != int(ref_hs[r, 1]) else "same"}', flush=True)
                if args.lanes:
                    got = hs[r, 16:272].view(32, 8).cpu().numpy().view(np.float32); want = ref_hs[r, 16:272].view(32, 8).cpu().numpy().view(np.float32)
                    rr_ = hs[r, 272:278].cpu().numpy().view(np.float32)
                    lab = ['ox', 'oy', 'oz', 'dx', 'dy', 'dz', 'o.o', 'o.d'] if args.lanes_mode == 2 else ['act', 'radius', 'o.o', 'o.d', 'disc', 'sqrt', 't1', 't2']
                    for ln in np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(-1))[0][:4]:
                        bad_cols = np.nonzero(got[ln].view(np.uint32) != want[ln].view(np.uint32))[0]
                        print(f'   lane {ln} of the ray (ro {rr_[:3].tolist()}, rd {rr_[3:].tolist()}): ' + '; '.join(f'{lab[c]} got {got[ln, c]!r} ({got[ln, c].view(np.uint32):#010x}) alone {want[ln, c]!r} ({want[ln, c].view(np.uint32):#010x})' for c in bad_cols), flush=True)
                        print('      all eight, got  :', [float(x) for x in got[ln]]); print('      all eight, alone:', [float(x) for x in want[ln]], flush=True)
                    if args.lanes_mode == 3:        # forensic: which float32 evaluation of o.o / o.d reproduces the value that was got?
                        hc = fns[0].model._hc
                        sc = np.array([float(hc.origin_initial[i]) for i in range(3)], np.float32)
                        o = rr_[:3].astype(np.float32) * sc; d = rr_[3:].astype(np.float32) * sc
                        f32 = np.float32
                        def fma(a, b, c): return f32(np.float64(a) * np.float64(b) + np.float64(c))
                        sq = [f32(x * x) for x in o]; od_ = [f32(a * b) for a, b in zip(o, d)]
                        cands = {}
                        import itertools
                        for perm in itertools.permutations(range(3)):
                            i, j, k_ = perm
                            cands[f'oo ({i}+{j})+{k_}'] = f32(f32(sq[i] + sq[j]) + sq[k_])
                            cands[f'oo fma({i},{i},sq{j})+{k_}'] = f32(fma(o[i], o[i], sq[j]) + sq[k_])
                            cands[f'oo fma({k_},{k_},({i}+{j}))'] = fma(o[k_], o[k_], f32(sq[i] + sq[j]))
                            cands[f'od ({i}+{j})+{k_}'] = f32(f32(od_[i] + od_[j]) + od_[k_])
                            cands[f'od fma({k_},({i}+{j}))'] = fma(o[k_], d[k_], f32(od_[i] + od_[j]))
                        print('      scale', sc.tolist(), 'o', o.tolist(), 'd', d.tolist())
                        for ln in np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(-1))[0][:2]:
                            for c, nm in ((2, 'oo'), (3, 'od')):
                                print(f'      lane {ln} {nm}: got matches', [k for k, v in cands.items() if k.startswith(nm) and v.view(np.uint32) == got[ln, c].view(np.uint32)][:6], '| alone matches', [k for k, v in cands.items() if k.startswith(nm) and v.view(np.uint32) == want[ln, c].view(np.uint32)][:6], flush=True)
                    lane_cols.append(tuple(sorted(set(int(c) for ln in range(32) for c in np.nonzero(got[ln].view(np.uint32) != want[ln].view(np.uint32))[0]))))
                    continue
                names = ['activated head value (sigmoid: exp, division)', 'radius (inverse contraction)', 'o.o', 'o.d', 'discriminant', 'sqrt', 't1 (division)', 't2 (division)']
                first = [nm for i, nm in enumerate(names) if int(hs[r, 8 + i]) != int(ref_hs[r, 8 + i])]
                firsts.append(first[0] if first else 'none of the intermediates')
                print('   intermediates that differ, in evaluation order:', first, flush=True)
if args.lanes: print('which of (ox oy oz dx dy dz o.o o.d) differ, per differing render:', {k: lane_cols.count(k) for k in set(lane_cols)})
print('first differing intermediate:', {k: firsts.count(k) for k in set(firsts)})
print(f'double reads of the head that disagreed (all renders): {mism_total};', end=' ')
print(f'{bad} of {args.iters * 6} renders differed; rows of the 64-ray tile that differed: {sorted(set(rows))}', flush=True)
os._exit(0)
