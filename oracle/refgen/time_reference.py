"""Calibration of bench.py's CPU baseline: the REFERENCE ITSELF (imported through ref_shim.py) timed next to the PyTorch-op
port of its algorithm (oracle/torch_port.py, what bench.py's `cpu_baseline` runs on the GPU box, where /root/reference does
not exist) on the same rays, same weights, same thread count.  Authoring container only.

    python oracle/refgen/time_reference.py [model] [n_rays] -> profiles/r02_cpu_calibration.json

TEST / MEASUREMENT INFRASTRUCTURE ONLY."""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import ref_shim  # noqa: E402
from hyperreel_amd import config as C, scenes  # noqa: E402
from torch_port import TorchPort  # noqa: E402


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else 'donerf_sphere'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    chunk = 16384
    cfg, ds = C.model_config(model), C.dataset_scalars(model)
    sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
    grid = [int(v) for v in sd['model.color_model.net.gridSize']]
    rays = scenes.benchmark_rays(model, 800, 800, frame=7)
    idx = np.sort(np.random.default_rng(0).choice(rays.shape[0], n, replace=False))
    r = np.ascontiguousarray(rays[idx])

    def overrides(c):
        c.color.net.grid_size = ref_shim.to_attr({'start': grid, 'end': grid})
    fn = ref_shim.build_reference(ref_shim.load_model_cfg(model, overrides), ds)
    own = dict(fn.state_dict())
    with torch.no_grad():
        for k, v in sd.items():
            if not k.endswith('gridSize'):
                own[k].copy_(torch.from_numpy(v))
    port = TorchPort(cfg, ds, sd)
    tr = torch.from_numpy(r)
    out = {'model': model, 'grid': grid, 'rays': n, 'chunk': chunk, 'host_cpus': os.cpu_count(), 'runs': []}
    for thr in (8, os.cpu_count()):
        torch.set_num_threads(thr)

        def ref_once():
            return torch.cat([ref_shim.run_reference(fn, tr[i:i + chunk])['rgb'] for i in range(0, n, chunk)], 0)
        ref_once(); port.render(r, chunk=chunk)                      # warm-up
        t = []
        for f in (ref_once, lambda: port.render(r, chunk=chunk)['rgb']):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); y = f(); best = min(best, time.perf_counter() - t0)
            t.append((best, np.asarray(y)))
        linf = float(np.abs(t[0][1] - t[1][1]).max())
        out['runs'].append({'threads': thr, 'reference_mrays_s': n / t[0][0] / 1e6, 'port_mrays_s': n / t[1][0] / 1e6,
                            'port_over_reference': t[0][0] / t[1][0], 'linf_port_vs_reference': linf})
        print(out['runs'][-1], flush=True)
        if thr == os.cpu_count():
            break
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'r02_cpu_calibration.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
