"""sha1 of the bench frame's pixels (and of its raw head) in a given MLP arithmetic, for comparing library builds bit for bit:
    [HR_LIB=tools/_bin/libhr_x.so] python tools/frame_digest.py [f16f8|f16x3|auto ...]"""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyperreel_amd import lib as hl
if os.environ.get('HR_LIB'):
    hl.LIB_PATH = os.path.abspath(os.environ['HR_LIB'])
from hyperreel_amd import config as C, scenes
from hyperreel_amd.render import build_render_fn

for model in ('donerf_sphere', 'technicolor_z_plane'):
    cfg, ds = C.model_config(model), C.dataset_scalars(model)
    sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
    grid = [int(v) for v in sd['model.color_model.net.gridSize']]
    rays = torch.from_numpy(scenes.benchmark_rays(model, 800, 800, frame=7)).cuda()
    for prec in (sys.argv[1:] or ['f16f8']):
        f = build_render_fn(cfg, dataset=ds, grid_size=grid, mlp_precision=prec)
        f.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        rgb = f.model.render(rays)['rgb']
        head = f.model.render(rays[:4096], want=('head',))['head']
        print(model, prec, 'rgb', hashlib.sha1(rgb.cpu().numpy().tobytes()).hexdigest()[:16], 'head[:4096] (fields path)', hashlib.sha1(head.cpu().numpy().tobytes()).hexdigest()[:16],
              'rgb mean %.9f' % float(rgb.double().mean()))
