"""Who waits for whom inside the frame kernel?  Needs the tuning build: python tools/build_variant.py tuning -DHR_TUNING
python tools/frame_stats.py [sample_waves] [mode]   mode 0: normal, 1: MLP only (sample waves idle), 2: sample only"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import lib
lib.LIB_PATH = os.environ.get('HR_LIB') or os.path.join(ROOT, 'tools', '_bin', 'libhr_tuning.so')
from hyperreel_amd import config as C, scenes
from hyperreel_amd.render import build_render_fn

waves = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
prec = sys.argv[3] if len(sys.argv) > 3 else 'auto'
cfg, ds = C.model_config('donerf_sphere'), C.dataset_scalars('donerf_sphere')
sd = scenes.make_state_dict(cfg, ds, None, seed=7, density='dense', app_scale=1.0)
grid = [int(v) for v in sd['model.color_model.net.gridSize']]
fn = build_render_fn(cfg, dataset=ds, grid_size=grid, sample_waves=waves, mlp_precision=prec)
fn.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
rays = torch.from_numpy(scenes.benchmark_rays('donerf_sphere', 800, 800, frame=7)).cuda()
L = lib.load()
setter = getattr(ctypes.CDLL(lib.LIB_PATH), {'auto': 'hr_tuning_set_f16x3', 'f16x3': 'hr_tuning_set_f16x3', 'bf16x3': 'hr_tuning_set_bf16x3', 'f16x2': 'hr_tuning_set_f16x2'}[prec])
setter.argtypes = [ctypes.c_int, ctypes.c_void_p]
phases = getattr(ctypes.CDLL(lib.LIB_PATH), setter.__name__.replace('_set_', '_phases_'))
phases.argtypes = [ctypes.c_void_p, ctypes.c_int]
stats = torch.zeros((256, 8), dtype=torch.int64, device='cuda')
fn.model.render(rays); torch.cuda.synchronize()
setter(mode, ctypes.c_void_p(stats.data_ptr()))
phases(None, 1)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(10):
    fn.model.render(rays)
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 10
s = stats.cpu().numpy().astype(np.float64)
tiles = s[:, 6].mean()
f = lambda c: s[:, c].mean()
print(f'waves {waves} mode {mode} {prec}: {ms:.3f} ms/frame  tiles/WG {tiles:.1f}  [cycles per tile, 100 MHz counter x?]')
print(f'  MLP   total {f(0)/tiles:9.0f}  wait_first {f(1)/tiles:8.0f}  wait_done {f(2)/tiles:8.0f}  barriers {f(3)/tiles:8.0f}')
print(f'  sample total {f(4)/tiles:9.0f}  wait_ready {f(5)/tiles:8.0f}')
ph = (ctypes.c_ulonglong * 16)()
phases(ph, 0)
names = ['distance', 'sort', 'point+delta', 'taps', 'gather0', 'gather1', 'gather2', 'alpha+scan', 'colour+sum+store', 'ray load', 'decode M']
passes = 10 * 640000 / 2          # wave passes (2 rays each) over the 10 frames
tot = sum(ph[i] for i in range(11))
print('  sample phases, cycles per wave pass: ' + '  '.join(f'{n} {ph[i] / passes:.0f}' for i, n in enumerate(names)) + f'  | sum {tot / passes:.0f}')
