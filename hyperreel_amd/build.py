"""Builds libhyperreel_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m hyperreel_amd.build [--force] [--verbose]

The library links only against the HIP runtime; it has no torch dependency.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, '_build')
LIB_PATH = os.path.join(LIB_DIR, 'libhyperreel_hip.so')
SOURCES = ['api.hip', 'mlp_kernel.hip', 'mlp_bf16x3_kernel.hip', 'mlp_f16x3_kernel.hip', 'mlp_f16x2_kernel.hip', 'sample_kernel.hip', 'pack_kernels.hip', 'train_kernel.hip']
HEADERS = ['hr_kernels.h', 'hr_math.h', 'hr_grid.h', 'hr_train.h', 'hr_mask.h', 'mlp_split_impl.inc', os.path.join('..', '..', 'include', 'hyperreel_hip.h')]

# -ffp-contract=off: the per-sample arithmetic follows the reference operation by
# operation (the reference never fuses a multiply with an add across torch ops); the
# places where fusion is wanted use __builtin_fmaf explicitly.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
         '-fno-math-errno', '-Wall', '-Wno-unused-function']
# 1-ulp hardware exp (+ rcp inside sigmoid/tanh) for VALUES that are never compared against a threshold; divisions, square
# roots and sin/cos keep their IEEE forms so that the reference's exact comparisons fall the same way (csrc/hr_math.h)
if os.environ.get('HR_FAST_EXP', '1') != '0':
    FLAGS.append('-DHR_FAST_EXP')
if os.environ.get('HR_FAST_POST', '1') != '0':       # rcp / sqrt / tanh approximations strictly after the near/far mask
    FLAGS.append('-DHR_FAST_POST')
if os.environ.get('HR_FAST_MATH', '0') == '1':       # measurements only: all three approximation groups
    FLAGS.append('-DHR_FAST_MATH')


def hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    for s in SOURCES:
        obj = os.path.join(LIB_DIR, s.replace('.hip', '.o'))
        cmd = [hipcc(), *FLAGS, *extra_flags, '-c', os.path.join(CSRC, s), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', LIB_PATH]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv or '-v' in sys.argv))
