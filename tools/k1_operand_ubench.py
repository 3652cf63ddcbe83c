"""What bounds K1 (hr_mlp_f16f8_kernel)?  The product kernel against builds of the SAME kernel -- same tiling, same buffer-load ring,
same LDS reads, same epilogues -- with one ingredient of the inner loop taken out (csrc/mlp_split_core.inc, HR_K1_UBENCH_*):

    nomfma      operands arrive (loads + waits as in the product), no matrix instruction is issued  -> the operand supply alone
    wfixed      every weight load of a layer reads the layer's first k-step: vector-L1 hits, no L2 -> L1 traffic
    nowload     no weight loads after each layer's prologue                                          -> matrix pipe + LDS reads + epilogue
    noxload     no LDS reads of the rays' operands in the loop
    nomfma_nox  neither MFMAs nor LDS reads: the weight stream + epilogue
    nomfma_now  neither MFMAs nor weight loads: the LDS reads + epilogue

python tools/k1_operand_ubench.py   (after the variants were built: see the recipe at the bottom of profiles/r06_k1_operand_ubench.txt).
Times hr_stage_mlp (K1 alone, 131 072 rays per launch, HIP events on the launch stream) for the DoNeRF model; each variant in its own
process (one library per process).  The variants compute garbage; only their time means anything."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = ['product', 'k1_nomfma', 'k1_wfixed', 'k1_nowload', 'k1_noxload', 'k1_nomfma_nox', 'k1_nomfma_now']


def child(lib):
    sys.path.insert(0, ROOT)
    import ctypes
    import numpy as np
    import torch
    from hyperreel_amd import lib as hl
    if lib != 'product':
        hl.LIB_PATH = os.path.join(ROOT, 'tools', '_bin', f'libhr_{lib}.so')
        assert os.path.exists(hl.LIB_PATH), hl.LIB_PATH
    from hyperreel_amd import config as C, scenes
    from hyperreel_amd.render import build_render_fn
    cfg, ds = C.model_config('donerf_sphere'), C.dataset_scalars('donerf_sphere')
    sd = scenes.make_state_dict(cfg, ds, [64, 64, 64], seed=7, density='dense', app_scale=1.0)
    f = build_render_fn(cfg, dataset=ds, grid_size=[64, 64, 64], mlp_precision='f16f8')
    f.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    h = f.model.native()
    rays = torch.from_numpy(scenes.benchmark_rays('donerf_sphere', 800, 800, frame=7)[:131072 * 4]).cuda()
    L = hl.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        for o in range(4):
            hl.check(L.hr_stage_mlp(h, ctypes.c_void_p(rays.data_ptr() + o * 131072 * rays.shape[1] * 4), 131072, st), 'hr_stage_mlp')
    for _ in range(60):          # the clock ramp of an idle GPU
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 4)
    ts.sort()
    print(json.dumps({'variant': lib, 'ms_per_launch_min': round(ts[0], 4), 'p50': round(ts[len(ts) // 2], 4)}))


if __name__ == '__main__':
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for rnd in range(2):
            for v in VARIANTS:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), v], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                print(rnd, next((ln for ln in r.stdout.splitlines() if ln.startswith('{')), f'{v}: failed rc {r.returncode}'), flush=True)
