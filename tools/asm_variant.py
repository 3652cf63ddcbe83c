"""A measurement build whose DEVICE code is patched at the assembly level: python tools/asm_variant.py <name> <base-variant> <mode> [-D...]
Re-compiles the translation units that contain the sample stage (fused_f16x3_kernel.hip, sample_kernel.hip) to assembly, inserts wait states
according to <mode>, assembles, links, bundles and compiles the host side around the patched code object; every other object comes from
tools/_bin/obj_<base-variant> (tools/build_variant.py).  -> tools/_bin/libhr_<name>.so
Modes (hipcc inserts what the ISA manual lists; these add MORE, to find which distance the hardware needs beside co-issued MFMA wavefronts):
  trans:N     s_nop N after every transcendental (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos)
  divfmas:N   s_nop N in front of every v_div_fmas (reads the VCC its v_div_scale wrote)
  cndmask:N   s_nop N in front of every v_cndmask
Measurement aid; the product build is hyperreel_amd/build.py."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperreel_amd import build as B
LL = '/opt/rocm/lib/llvm/bin'
PATCHED = ['fused_f16x3_kernel.hip'] if os.environ.get('HR_ASM_FUSED_ONLY') else ['fused_f16x3_kernel.hip', 'sample_kernel.hip']
TRANS = re.compile(r'^\s+v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag|_legacy)?_(f32|f16)')


def patch_region(text, kind, n):
    """padsample / padmlp: s_nop n in front of EVERY instruction of one role of the frame kernels -- inside each kernel function, everything from
    the first s_setprio on is taken as the MLP role, everything before it as the sample role (block placement keeps the two apart)"""
    lines = text.split('\n')
    out, cnt = [], 0
    i = 0
    while i < len(lines):
        m = re.match(r'^(_Z\w+):', lines[i])
        if not m:
            out.append(lines[i]); i += 1; continue
        j = i + 1
        while j < len(lines) and not lines[j].startswith('.Lfunc_end'):
            j += 1
        body = lines[i:j]
        # block placement puts the sample role first; the MLP role starts at its first s_setprio
        last = min((t for t, ln in enumerate(body) if re.match(r'^\s+s_setprio', ln)), default=-1)
        for t, ln in enumerate(body):
            is_inst = re.match(r'^\s+[vs]_|^\s+ds_|^\s+global_|^\s+buffer_|^\s+flat_|^\s+scratch_', ln) is not None
            in_mlp = t >= last
            if is_inst and last >= 0 and ((kind == 'padmlp' and in_mlp) or (kind == 'padsample' and not in_mlp)) and not re.match(r'^\s+s_(nop|endpgm|branch|cbranch|waitcnt|barrier|sleep|setprio)', ln):
                out.append(f'\ts_nop {n}'); cnt += 1
            out.append(ln)
        i = j
    return '\n'.join(out), cnt


def patch(text, mode):
    kind, _, n = mode.partition(':')
    n = int(n or 3)
    if kind in ('padsample', 'padmlp'):
        return patch_region(text, kind, n)
    out, cnt = [], 0
    for ln in text.split('\n'):
        if kind == 'divfmas' and re.match(r'^\s+v_div_fmas', ln):
            out.append(f'\ts_nop {n}'); cnt += 1
        if kind == 'cndmask' and re.match(r'^\s+v_cndmask', ln):
            out.append(f'\ts_nop {n}'); cnt += 1
        out.append(ln)
        if kind == 'trans' and TRANS.match(ln):
            out.append(f'\ts_nop {n}'); cnt += 1
    return '\n'.join(out), cnt


def main(name, base, mode, extra):
    out = os.path.join(ROOT, 'tools', '_bin')
    objd = os.path.join(out, 'obj_' + name)
    os.makedirs(objd, exist_ok=True)
    flags = [*B.FLAGS, *extra]
    objs = []
    for s in B.SOURCES:
        if s not in PATCHED:
            objs.append(os.path.join(out, 'obj_' + base, s.replace('.hip', '.o')))
            continue
        src = os.path.join(B.CSRC, s)
        stem = os.path.join(objd, s.replace('.hip', ''))
        subprocess.run([B.hipcc(), *flags, '-S', '--cuda-device-only', src, '-o', stem + '.s'], check=True, stderr=subprocess.DEVNULL)
        text, cnt = patch(open(stem + '.s').read(), mode)
        open(stem + '_p.s', 'w').write(text)
        print(s, 'patched sites:', cnt, flush=True)
        subprocess.run([f'{LL}/clang', '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', stem + '_p.s', '-o', stem + '_dev.o'], check=True)
        subprocess.run([f'{LL}/lld', '-flavor', 'gnu', '-m', 'elf64_amdgpu', '--no-undefined', '-shared', '-o', stem + '.out', stem + '_dev.o'], check=True)
        subprocess.run([f'{LL}/clang-offload-bundler', '-type=o', '-bundle-align=4096', '-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950',
                        '-input=/dev/null', '-input=' + stem + '.out', '-output=' + stem + '.hipfb'], check=True)
        subprocess.run([B.hipcc(), *flags, '--cuda-host-only', '-c', src, '-o', stem + '.o', '-Xclang', '-fcuda-include-gpubinary', '-Xclang', stem + '.hipfb'],
                       check=True, stderr=subprocess.DEVNULL)
        objs.append(stem + '.o')
    lib = os.path.join(out, f'libhr_{name}.so')
    subprocess.run([B.hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', lib], check=True)
    print(lib)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:])
