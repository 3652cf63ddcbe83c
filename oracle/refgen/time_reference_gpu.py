"""The REFERENCE ITSELF timed on the MI355X next to the PyTorch-op port of its algorithm (oracle/torch_port.py, what bench.py's
`pytorch_gpu_baseline` runs, because /root/reference does not exist on the GPU box) -- SURVEY 8(d) item 2 / VERDICT r4 item 6: the north
star's ">= 10x the reference PyTorch single-GPU rays/s" needs the reference as the denominator, or a port with a stated calibration.

The reference tree is shipped to ONE gpurun lease as a git-ignored input directory (never committed, deleted afterwards):

    cp -r /root/reference gpurun_in_reference
    gpurun -- 'HR_REF_ROOT=$PWD/gpurun_in_reference HR_REF_DEVICE=cuda python oracle/refgen/time_reference_gpu.py gpurun_out/r05_gpu_calibration.json'

Same frame as bench.py (DoNeRF 800x800, 640 000 rays, grid 600^3, seeded weights), the reference bracketed as nlf/__init__.py:841-850 does
(render_chunked over the whole frame, torch.no_grad, eval mode), at its shipped ray_chunk (16 384) and at one chunk per frame (1 048 576).

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
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'r05_gpu_calibration.json')
    model = sys.argv[2] if len(sys.argv) > 2 else 'donerf_sphere'
    assert ref_shim.ON_CUDA and torch.cuda.is_available(), 'run with HR_REF_DEVICE=cuda on a GPU box'
    cfg, ds = C.model_config(model), C.dataset_scalars(model)
    sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
    grid = [int(v) for v in sd['model.color_model.net.gridSize']]
    rays_np = scenes.benchmark_rays(model, 800, 800, frame=7)
    rays = torch.from_numpy(rays_np).cuda()
    n = rays.shape[0]

    def overrides(c):
        c.color.net.grid_size = ref_shim.to_attr({'start': grid, 'end': grid})
    fn = ref_shim.build_reference(ref_shim.load_model_cfg(model, overrides), ds).cuda()
    own = dict(fn.state_dict())
    with torch.no_grad():
        for k, v in sd.items():
            if not k.endswith('gridSize'):
                own[k].copy_(torch.from_numpy(v))
    port = TorchPort(cfg, ds, sd, device='cuda')

    from hyperreel_amd.render import build_render_fn
    hip = build_render_fn(cfg, dataset=ds, grid_size=grid)
    hip.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    hip_rgb = hip.model.render(rays)['rgb'].clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        hip.model.render(rays)
    torch.cuda.synchronize()
    hip_ms = (time.perf_counter() - t0) / 20 * 1e3

    out = {'model': model, 'grid': grid, 'rays': n, 'device': torch.cuda.get_device_name(0), 'torch': torch.__version__,
           'bracket': 'render_chunked(rays, RenderLightfield.eval(), chunk) under torch.no_grad(), wall clock around 3 frames after one warm-up frame, synchronize on both sides (nlf/__init__.py:841-850)',
           'hip_eager_ms_per_frame': round(hip_ms, 4), 'runs': []}
    for chunk in (16384, 1048576):
        def ref_once():
            return ref_shim.run_reference(fn, rays, chunk=chunk)['rgb']

        def port_once():
            return port.render(rays, chunk=chunk)['rgb']
        res = {}
        for name, f in (('reference', ref_once), ('port', port_once)):
            y = f()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                y = f()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            res[name] = (best, y)
        run = {'chunk': chunk, 'reference_mrays_s': n / res['reference'][0] / 1e6, 'port_mrays_s': n / res['port'][0] / 1e6,
               'port_over_reference': res['reference'][0] / res['port'][0],
               'linf_port_vs_reference': float((res['reference'][1] - res['port'][1]).abs().max()),
               'linf_hip_vs_reference': float((res['reference'][1] - hip_rgb).abs().max()),
               'rays_over_1e-4_hip_vs_reference': int(((res['reference'][1] - hip_rgb).abs().amax(-1) > 1e-4).sum())}
        out['runs'].append(run)
        print(run, flush=True)
    best_ref = max(r['reference_mrays_s'] for r in out['runs'])
    out['reference_best_mrays_s'] = best_ref
    out['hip_eager_over_reference'] = n / hip_ms / 1e3 / best_ref
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(out, open(out_path, 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
