import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from hyperreel_amd import lib as _l
if os.environ.get("HR_DBG_LIB"): _l.LIB_PATH = os.environ["HR_DBG_LIB"]
from helpers import Golden
from gpu_common import make_render_fn
g = Golden('donerf_sphere_small')
fn = make_render_fn(g.cfg, g.dataset, g.state_dict, mlp_precision='bf16x3')
rays = torch.from_numpy(np.concatenate([g.rays] * 3, 0)).cuda()
fn.model.set_execution(frame_kernel=False)
two = fn.model.render(rays)['rgb'].clone()
fn.model.set_execution(frame_kernel=True, sample_waves=8)
import ctypes as C
L = _l.load(); h = fn.model.native()
for trial in range(3):
    out = torch.full((rays.shape[0], 3), float('nan'), device='cuda')
    _l.check(L.hr_render(h, C.c_void_p(rays.data_ptr()), rays.shape[0], C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'r')
    torch.cuda.synchronize()
    bad = (out != two).any(-1) | torch.isnan(out).any(-1)
    print('trial', trial, 'nan rays', int(torch.isnan(out).any(-1).sum()), 'differing', int(bad.sum()), torch.nonzero(bad)[:12, 0].tolist())
