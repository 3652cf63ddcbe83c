import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from hyperreel_amd import lib as _l
import os as _o
if _o.environ.get("HR_DBG_LIB"): _l.LIB_PATH = _o.environ["HR_DBG_LIB"]
from helpers import Golden
from gpu_common import make_render_fn
g = Golden('donerf_sphere_small')
for prec in ('bf16x3', 'f16x3'):
    fn = make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision=prec)
    rays = torch.from_numpy(np.concatenate([g.rays] * 3, 0)).cuda()
    outs = {}
    for name, fk, w in (('two', False, None), ('f8', True, 8), ('f4', True, 4)):
        fn.model.set_execution(frame_kernel=fk, sample_waves=w)
        outs[name] = fn.model.render(rays)['rgb'].cpu().numpy()
    n = g.rays.shape[0]
    for k, v in outs.items():
        e = np.abs(v[:n] - g.rgb).max(-1)
        print(prec, k, 'vs golden: max %.3e  rays>1e-4: %d' % (e.max(), (e > 1e-4).sum()), ' vs two: max %.3e  differing rays %d first %s' % (
            np.abs(v - outs['two']).max(), (v != outs['two']).any(-1).sum(), np.nonzero((v != outs['two']).any(-1))[0][:8]))
    for kname, v in outs.items():
        print(prec, kname, 'copies equal copy 1:', np.array_equal(v[:n], v[n:2 * n]), np.array_equal(v[:n], v[2 * n:3 * n]),
              'rows differing from copy 1:', np.nonzero((v[n:2 * n] != v[:n]).any(-1))[0][:5], np.nonzero((v[2 * n:] != v[:n]).any(-1))[0][:5])
        if kname == 'f8':
            for r in (449, 641):
                print('   ray', r, 'fused', v[r], 'two', outs['two'][r], 'copy1 fused', v[r % n])
