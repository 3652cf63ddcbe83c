"""The shipped ARTIFACT contains no packed-fp32 instruction.

One compiler-formed in-place `v_pk_mul_f32 ... op_sel:[0,1]` of the sample stage lost its low result in the last 16 lanes of a wavefront
that shared its SIMD with MFMA wavefronts (profiles/r05_frame_kernel_difference_bisect.txt) -- about once in 1e8 samples, which
the GPU regression tests catch only statistically.  The fix is a pair of build flags (hyperreel_amd/build.py: -fno-slp-vectorize,
-target-feature -packed-fp32-ops); a toolchain that ignores the hidden target feature, or a variant build that overrides FLAGS, would
re-admit the instruction silently.  So the code objects themselves are inspected: every gfx950 bundle of every library the tree builds is
extracted (llvm-objdump --offloading), disassembled, and searched.  CPU only (hipcc cross-compiles; nothing runs)."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
PACKED_F32 = re.compile(r'\bv_pk_(mul|add|fma)_f32\b')


def _libraries():
    from hyperreel_amd import build as B
    libs = [B.build()]
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from gpu_common import build_poison
    libs.append(build_poison())
    libs += sorted(glob.glob(os.path.join(ROOT, 'tools', '_bin', 'libhr_*.so')))          # measurement variants (tools/build_variant.py), when built
    return libs


def _disassemble(lib, tmp):
    """{bundle name: disassembly text} of the gfx950 code objects inside `lib` (extracted next to a COPY of it under tmp)."""
    d = os.path.join(tmp, os.path.basename(lib) + '.d')
    os.makedirs(d)
    copy = shutil.copy(lib, d)
    subprocess.run([OBJDUMP, '--offloading', copy], check=True, stdout=subprocess.DEVNULL, cwd=d)
    out = {}
    for f in sorted(glob.glob(copy + '.*gfx950*')):
        out[os.path.basename(f)] = subprocess.run([OBJDUMP, '-d', '--mcpu=gfx950', f], check=True, stdout=subprocess.PIPE, text=True).stdout
    return out


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain not found')
def test_no_packed_fp32_instruction_in_any_built_library(tmp_path):
    seen_mfma = False
    for lib in _libraries():
        if os.path.basename(lib).startswith('libhr_') and os.environ.get('HR_GUARD_VARIANTS', '1') == '0':
            continue
        code = _disassemble(lib, str(tmp_path))
        assert code, f'{lib}: no gfx950 code object found (did the bundle format change?)'
        for name, text in code.items():
            assert 's_endpgm' in text, f'{name}: the disassembly holds no kernel'
            hits = PACKED_F32.findall(text)
            lines = [ln.strip() for ln in text.splitlines() if PACKED_F32.search(ln)][:3]
            assert not hits, f'{lib} / {name}: {len(hits)} packed-fp32 instructions, e.g. {lines} -- the build flags of hyperreel_amd/build.py no longer hold'
            seen_mfma = seen_mfma or 'v_mfma_f32_32x32x16_f16' in text
    assert seen_mfma, 'the product library holds no v_mfma_f32_32x32x16_f16: this is not the disassembly of the MLP kernels'


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain not found')
def test_the_guard_sees_a_packed_instruction_when_there_is_one(tmp_path):
    """The search itself: the same tiny kernel built with and without the flags -- with them 0, without them at least one v_pk_*_f32."""
    from hyperreel_amd import build as B
    src = tmp_path / 'pk.hip'
    src.write_text('#include <hip/hip_runtime.h>\n'
                   'typedef float f2 __attribute__((ext_vector_type(2)));\n'
                   '__global__ void k(const f2* a, const f2* b, f2* c) { int i = threadIdx.x; c[i] = a[i] * b[i] + a[i]; }\n')
    counts = {}
    for tag, flags in (('plain', ['--offload-arch=gfx950', '-O3']), ('guarded', [f for f in B.FLAGS if not f.startswith('-D')])):
        lib = tmp_path / f'lib_{tag}.so'
        subprocess.run([B.hipcc(), *flags, '-shared', '-fPIC', str(src), '-o', str(lib)], check=True, stderr=subprocess.DEVNULL)
        text = ''.join(_disassemble(str(lib), str(tmp_path)).values())
        counts[tag] = len(PACKED_F32.findall(text))
    assert counts['plain'] >= 1 and counts['guarded'] == 0, counts
