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
SOURCES = ['api.hip', 'mlp_kernel.hip', 'mlp_bf16x3_kernel.hip', 'mlp_f16x3_kernel.hip', 'mlp_f16x2_kernel.hip', 'mlp_f16f8_kernel.hip', 'fused_bf16x3_kernel.hip', 'fused_f16x3_kernel.hip', 'fused_f16x2_kernel.hip', 'fused_f16f8_kernel.hip',
           'sample_kernel.hip', 'range_kernel.hip', 'band_kernel.hip', 'pack_kernels.hip', 'train_kernel.hip', 'train_det_kernel.hip', 'train_gemm_kernel.hip']
HEADERS = ['hr_kernels.h', 'hr_math.h', 'hr_grid.h', 'hr_train.h', 'hr_mask.h', 'mlp_split_impl.inc', 'mlp_split_core.inc', 'sample_core.inc', 'fused_impl.inc', 'train_kernel.hip', os.path.join('..', '..', 'include', 'hyperreel_hip.h')]

# -ffp-contract=off: the per-sample arithmetic follows the reference operation by
# operation (the reference never fuses a multiply with an add across torch ops); the
# places where fusion is wanted use __builtin_fmaf explicitly.
# -fno-slp-vectorize: no packed-fp32 instructions built by the compiler out of the sample stage's scalar arithmetic.  One of them -- an
# in-place  v_pk_mul_f32 v[2:3], v[52:53], v[2:3] op_sel:[0,1]  in the sphere intersection -- loses its low result in the last 16 lanes of a
# sample wavefront that shares its SIMD with MFMA wavefronts (the frame kernel with 8 sample wavefronts; the MLP kernel of one stream beside the
# sample kernel of another), about once in 1e8 samples: pinned at the assembly level, not reproducible in isolation (DESIGN 4,
# profiles/r05_frame_kernel_difference_bisect.txt).  Without the flag's packed forms: 0 differing renders in 600 + 960 + 480 (was 600, 2-13
# per 80, 0-6 per 40), the same bits everywhere else, no cost (K2 0.697 vs 0.700 ms).  The gather's hand-written packed instructions stay.
# -target-feature -packed-fp32-ops (device side; the host side of hipcc prints "not a recognized feature ... ignoring"): NO packed-fp32
# instruction at all -- the gather's hand-written 2-float arithmetic becomes pairs of 32-bit instructions too.  Its register allocation had
# produced the same class of instruction (in place, the overwritten pair read across halves: 348 in the sample kernel); none was ever seen to
# fail, and removing them is free: a packed-fp32 instruction takes two issue passes like the two it replaces (DoNeRF frame 1.775 / 1.779 ms
# with, 1.768 / 1.784 without, K2 0.689 vs 0.690 ms; profiles/r05_no_packed_fp32_ab.txt).
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-slp-vectorize',
         '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops',
         '-fno-math-errno', '-Wall', '-Wno-unused-function']
# 1-ulp hardware exp (+ rcp inside sigmoid/tanh) for VALUES that are never compared against a threshold; divisions, square
# roots and sin/cos keep their IEEE forms so that the reference's exact comparisons fall the same way (csrc/hr_math.h)
if os.environ.get('HR_FAST_EXP', '1') != '0':
    FLAGS.append('-DHR_FAST_EXP')
if os.environ.get('HR_FAST_POST', '1') != '0':       # rcp / sqrt / tanh approximations strictly after the near/far mask
    FLAGS.append('-DHR_FAST_POST')
if os.environ.get('HR_FAST_MATH', '0') == '1':       # measurements only: all three approximation groups
    FLAGS.append('-DHR_FAST_MATH')


def csrc_hash():
    """sha256 over the names and bytes of everything the library is built from (csrc/, the public header, the flags): the identity of
    the kernels.  tools/make_counters.py stamps it into the PMC counter files and bench.py attaches counters to its line only when it
    equals the running tree's -- counters of another tree are reported as "stale", not quoted."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.listdir(CSRC)) + [os.path.join('..', '..', 'include', 'hyperreel_hip.h')]
    for f in files:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            h.update(os.path.basename(f).encode() + b'\0')
            h.update(open(p, 'rb').read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()[:16]


def hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _deps():
    return [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]


def _obj(src):
    return os.path.join(LIB_DIR, src.replace('.hip', '.o'))


def _flags_stamp(flags):
    """Objects are reused only if they were built with the same flags."""
    path = os.path.join(LIB_DIR, 'flags.txt')
    text = ' '.join(flags)
    same = os.path.exists(path) and open(path).read() == text
    return same, path, text


def _stale_sources(flags_same):
    out = []
    newest_dep = max(os.path.getmtime(d) for d in _deps())
    for s in SOURCES:
        o = _obj(s)
        if not flags_same or not os.path.exists(o) or os.path.getmtime(o) < max(newest_dep, os.path.getmtime(os.path.join(CSRC, s))):
            out.append(s)
    return out


def build(force=False, verbose=False, extra_flags=()):
    """Compiles what is out of date (one hipcc per translation unit, in parallel) and links the library."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB_DIR, exist_ok=True)
    flags = [*FLAGS, *extra_flags]
    same, stamp, text = _flags_stamp(flags)
    todo = list(SOURCES) if force else _stale_sources(same)
    if not todo and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(_obj(s)) for s in SOURCES):
        return LIB_PATH

    def compile_one(s):
        cmd = [hipcc(), *flags, '-c', os.path.join(CSRC, s), '-o', _obj(s)]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        # the device-side target feature is offered to the host compilation too, which says so once per translation unit
        err = '\n'.join(ln for ln in r.stderr.split('\n') if "'-packed-fp32-ops' is not a recognized feature" not in ln).strip()
        if err:
            print(err, file=sys.stderr, flush=True)
        if r.returncode:
            raise subprocess.CalledProcessError(r.returncode, cmd)

    # objects of translation units that are no longer part of the library do not travel with the tree
    keep = {os.path.basename(_obj(s)) for s in SOURCES}
    for f in os.listdir(LIB_DIR):
        if f.endswith('.o') and f not in keep:
            os.remove(os.path.join(LIB_DIR, f))
    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4) or 1) as ex:
        list(ex.map(compile_one, todo))
    with open(stamp, 'w') as f:
        f.write(text)
    cmd = [hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', *[_obj(s) for s in SOURCES], '-o', LIB_PATH]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv or '-v' in sys.argv))
