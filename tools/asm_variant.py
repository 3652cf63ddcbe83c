"""A measurement build whose DEVICE code is patched at the assembly level: python tools/asm_variant.py <name> <base-variant> <mode> [-D...]
Re-compiles the translation units that contain the sample stage (fused_f16x3_kernel.hip, sample_kernel.hip) to assembly, inserts wait states
according to <mode>, assembles, links, bundles and compiles the host side around the patched code object; every other object comes from
tools/_bin/obj_<base-variant> (tools/build_variant.py).  -> tools/_bin/libhr_<name>.so
Modes (hipcc inserts what the ISA manual lists; these add MORE, to find which distance the hardware needs beside co-issued MFMA wavefronts):
  trans:N     s_nop N after every transcendental (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos)
  divfmas:N   s_nop N in front of every v_div_fmas (reads the VCC its v_div_scale wrote)
  cndmask:N   s_nop N in front of every v_cndmask
  readlane:N  s_nop N after every v_readlane / v_readfirstlane (VALU writes an SGPR: the restores of spilled SGPRs)
  vcmp:N      s_nop N after every v_cmp
  sgprwar:N   s_nop N in front of every instruction that writes an SGPR a vector instruction read within the last HR_ASM_WINDOW (4) instructions
  nopat:F@L:N s_nop N in front of body line L of the kernels whose mangled name starts with F
  scalarpk[:dist|:rest|:at=F@a-b]  every v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 of the sample role replaced by two 32-bit instructions (same roundings)
  padsample:N / padmlp:N  s_nop N in front of every instruction of one role of the frame kernel
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


def sgprs(tok):
    """SGPR numbers named by an operand token ('s12', 's[12:15]', 'vcc' -> {'vcc'})"""
    tok = tok.strip()
    m = re.fullmatch(r's(\d+)', tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r's\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    if tok in ('vcc', 'vcc_lo', 'vcc_hi'):
        return {'vcc'}
    return set()


def patch_sgpr_war(text, n, window):
    """sgprwar:N -- write-after-read on scalar registers: when an instruction WRITES an SGPR (a scalar ALU result, v_readlane / v_readfirstlane, a
    v_cmp mask) that a vector instruction READ as an operand within the last `window` instructions, s_nop N in front of the writer.  (No such
    rule exists in the ISA manual or in the compiler; the question is whether a vector instruction delayed beside co-issued MFMA wavefronts can
    still be reading the register for its later lanes.)"""
    out, cnt = [], 0
    last_read = {}            # sgpr -> index of the last VALU instruction that read it
    idx = 0
    for ln in text.split('\n'):
        m = re.match(r'^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+|flat_\w+)\s*(.*?)\s*(;.*)?$', ln)
        if not m or ln.lstrip().startswith(('.', ';')):
            if re.match(r'^[\w.$]+:', ln):
                last_read.clear()             # a label: control flow joins, start over
            out.append(ln)
            continue
        op, rest = m.group(1), m.group(2)
        ops = [t for t in re.split(r',\s*', rest)] if rest else []
        writes, reads = set(), set()
        if op.startswith('s_'):
            if not re.match(r's_(nop|waitcnt|barrier|branch|cbranch|endpgm|sleep|setprio|cmp|bitcmp|store|dcache|icache|sendmsg|sethalt|trap|setkill)', op) and ops:
                writes = sgprs(ops[0])
                if re.match(r's_(and|or|andn2|orn2|xor|nand|nor|xnor)_saveexec', op):
                    pass
        elif op.startswith('v_'):
            if re.match(r'v_read(first)?lane_b32', op):
                writes = sgprs(ops[0]); ops = ops[1:]
            elif re.match(r'v_cmpx?_', op):
                if op.endswith('_e32') or (ops and not sgprs(ops[0]) and not ops[0].startswith('s')):
                    writes = {'vcc'}
                else:
                    writes = sgprs(ops[0]); ops = ops[1:]
            elif re.match(r'v_(div_scale|add_co|sub_co|subrev_co|addc_co|subb_co|subbrev_co|mad_u64_u32|mad_i64_i32)', op) and len(ops) > 1:
                writes = sgprs(ops[1]); ops = [ops[0]] + ops[2:]
            for t in ops[1:] if not re.match(r'v_cmpx?_', op) else ops:
                reads |= sgprs(t)
            if re.match(r'v_(div_fmas|cndmask_b32_e32|addc_co_u32_e32|subb_co_u32_e32)', op):
                reads.add('vcc')
        hot = [r for r in writes if r in last_read and idx - last_read[r] <= window]
        if hot:
            out.append(f'\ts_nop {n}'); cnt += 1
        out.append(ln)
        if op.startswith('v_'):
            for r in reads:
                last_read[r] = idx
        idx += 1
    return '\n'.join(out), cnt


PK = re.compile(r'^\s+v_pk_(mul|add|fma)_f32\s+(.*?)\s*(;.*)?$')


def _pk_operand(tok, sel):
    """element `sel` (0 / 1) of a packed-fp32 operand as a 32-bit operand string; None = cannot be expressed"""
    tok = tok.strip()
    m = re.fullmatch(r'([vs])\[(\d+):(\d+)\]', tok)
    if m:
        return f'{m.group(1)}{int(m.group(2)) + sel}'
    if re.fullmatch(r'-?\d+(\.\d+)?|0x[0-9a-fA-F]+', tok):          # inline constant / literal: only its low half is the constant
        return tok if sel == 0 else None
    return None


def scalarize_pk(line):
    """v_pk_{mul,add,fma}_f32 D, A, B[, C] with op_sel / op_sel_hi / neg_lo / neg_hi  ->  two 32-bit instructions with the same roundings.
    Returns the replacement lines, or None where it cannot be done without a temporary register."""
    m = PK.match(line)
    if not m:
        return None
    op, rest = m.group(1), m.group(2)
    mods = dict((k, [int(x) for x in v.split(',')]) for k, v in re.findall(r'(op_sel_hi|op_sel|neg_lo|neg_hi):\[([\d,]+)\]', rest))
    rest = re.sub(r'\s*(op_sel_hi|op_sel|neg_lo|neg_hi):\[[\d,]+\]', '', rest)
    toks = [t.strip() for t in rest.split(',')]
    nsrc = 3 if op == 'fma' else 2
    if len(toks) != nsrc + 1:
        return None
    dm = re.fullmatch(r'v\[(\d+):(\d+)\]', toks[0])
    if not dm:
        return None
    dst = [f'v{int(dm.group(1))}', f'v{int(dm.group(1)) + 1}']
    sel = [mods.get('op_sel', [0] * nsrc), mods.get('op_sel_hi', [1] * nsrc)]
    neg = [mods.get('neg_lo', [0] * nsrc), mods.get('neg_hi', [0] * nsrc)]
    halves = []
    for h in (0, 1):
        srcs = []
        for i in range(nsrc):
            o = _pk_operand(toks[1 + i], sel[h][i] if i < len(sel[h]) else (h if sel[h] is None else 0))
            if o is None:
                return None
            srcs.append(('-' if (i < len(neg[h]) and neg[h][i]) else '') + o)
        halves.append(srcs)
    if sum(1 for x in set(o.lstrip('-') for hs in halves for o in hs) if x.startswith('s')) > 1:
        return None                      # more than one scalar register through the constant bus
    mn = {'mul': 'v_mul_f32_e64', 'add': 'v_add_f32_e64', 'fma': 'v_fma_f32'}[op]
    ins = [f'\t{mn} {dst[h]}, ' + ', '.join(halves[h]) for h in (0, 1)]
    reads = [set(o.lstrip('-') for o in halves[h]) for h in (0, 1)]
    if dst[0] not in reads[1]:
        return ins
    if dst[1] not in reads[0]:
        return [ins[1], ins[0]]
    return None


def patch_scalarize(text, where=''):
    """scalarpk -- every packed-fp32 multiply / add / fma of the SAMPLE role of the frame kernels (everything in front of a kernel's first
    s_setprio) as two 32-bit instructions.  where = 'dist' / 'rest' (-DHR_PHASE_MARK builds): only those between the phase markers 102 (frame kernel: 10) and 0
    (head activation, intersection, near / far mask) / only the others"""
    lines = text.split('\n')
    out, done, kept = [], 0, 0
    in_dist = False
    i = 0
    while i < len(lines):
        if not re.match(r'^(_Z\w+):', lines[i]):
            out.append(lines[i]); i += 1; continue
        j = i + 1
        while j < len(lines) and not lines[j].startswith('.Lfunc_end'):
            j += 1
        body = lines[i:j]
        first = min((t for t, ln in enumerate(body) if re.match(r'^\s+s_setprio', ln)), default=-1)
        for t, ln in enumerate(body):
            pm = re.search(r'; HRPHASE (\d+)', ln)
            if pm:
                in_dist = pm.group(1) in ('102', '10')          # stand-alone kernel: 102, frame kernel: 10 = the marker in front of the sample body
            if where.startswith('at='):          # at=FUNCTION_PREFIX@first-last: only body lines first..last of the kernels whose mangled name starts so
                fn, _, rng = where[3:].partition('@')
                lo, _, hi = rng.partition('-')
                take = body[0].startswith(fn) and int(lo) <= t <= int(hi)
            else:
                take = not where or (where == 'dist') == in_dist
            if 0 <= t < first and PK.match(ln) and take:
                r = scalarize_pk(ln)
                if r is None:
                    kept += 1; out.append(ln)
                else:
                    done += 1; out.extend(r)
            else:
                out.append(ln)
        i = j
    print('packed instructions left as they were (would need a temporary):', kept, flush=True)
    return '\n'.join(out), done


def patch_nop_at(text, spec):
    """nopat:FUNCTION_PREFIX@line:N -- s_nop N in front of ONE instruction: body line `line` of the kernels whose mangled name starts with the prefix"""
    fn, _, rest = spec.partition('@')
    line, _, n = rest.partition(':')
    lines = text.split('\n')
    out, cnt, i = [], 0, 0
    while i < len(lines):
        if not re.match(r'^(_Z\w+):', lines[i]):
            out.append(lines[i]); i += 1; continue
        j = i + 1
        while j < len(lines) and not lines[j].startswith('.Lfunc_end'):
            j += 1
        for t, ln in enumerate(lines[i:j]):
            if lines[i].startswith(fn) and t == int(line):
                out.append(f'\ts_nop {int(n or 1)}'); cnt += 1
            out.append(ln)
        i = j
    return '\n'.join(out), cnt


def patch(text, mode):
    kind, _, n = mode.partition(':')
    if kind == 'nopat':
        return patch_nop_at(text, n)
    if kind == 'scalarpk':
        return patch_scalarize(text, n)
    n = int(n or 3)
    if kind == 'scalarpk':
        return patch_scalarize(text, mode.partition(':')[2])
    if kind in ('padsample', 'padmlp'):
        return patch_region(text, kind, n)
    if kind == 'sgprwar':
        return patch_sgpr_war(text, n, int(os.environ.get('HR_ASM_WINDOW', '4')))
    out, cnt = [], 0
    for ln in text.split('\n'):
        if kind == 'divfmas' and re.match(r'^\s+v_div_fmas', ln):
            out.append(f'\ts_nop {n}'); cnt += 1
        if kind == 'cndmask' and re.match(r'^\s+v_cndmask', ln):
            out.append(f'\ts_nop {n}'); cnt += 1
        if kind == 'salumask' and re.match(r'^\s+s_(and|or|andn2|orn2|xor|and_saveexec|or_saveexec|andn2_saveexec|mov)_b64|^\s+s_cbranch_(vcc|exec)', ln):     # scalar readers of lane masks
            out.append(f'\ts_nop {n}'); cnt += 1
        out.append(ln)
        if kind == 'trans' and TRANS.match(ln):
            out.append(f'\ts_nop {n}'); cnt += 1
        if kind == 'readlane' and re.match(r'^\s+v_read(first)?lane_b32', ln):     # a VALU instruction that writes an SGPR (SGPR-spill restores)
            out.append(f'\ts_nop {n}'); cnt += 1
        if kind == 'vcmp' and re.match(r'^\s+v_cmp', ln):                            # ... a lane mask
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
