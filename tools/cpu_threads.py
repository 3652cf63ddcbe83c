import sys, time, os
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np, torch
from hyperreel_amd import config as C, scenes
from torch_port import TorchPort
cfg, ds = C.model_config('donerf_sphere'), C.dataset_scalars('donerf')
sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
rays = scenes.benchmark_rays('donerf_sphere', 800, 800)[:131072]
tp = TorchPort(cfg, ds, sd)
print('cpus', os.cpu_count())
for thr in (8, 16, 32, 64, 128):
    torch.set_num_threads(thr)
    for chunk in (16384, 65536):
        tp.render(rays[:chunk], chunk=chunk)
        t=time.time(); tp.render(rays, chunk=chunk); dt=time.time()-t
        print(f'threads {thr:4d} chunk {chunk:6d}: {131072/dt/1e6:.4f} Mrays/s', flush=True)
