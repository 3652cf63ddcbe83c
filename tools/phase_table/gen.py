"""Per-phase instruction table of the sample stage for ONE model, from the disassembly.

    python tools/phase_table/gen.py [model] [--fp16]   ->  table on stdout (+ tools/_bin/phase_<model>.s)

The kernels read the configuration at run time (uniform branches); to count what a given model actually executes, this
tool bakes that model's hr_config and plane descriptors into a wrapper kernel as compile-time constants, lets the
compiler fold the branches, and counts the instructions between phase markers (`; HRPHASE n` comments emitted by the
HR_SPH hooks of sample_core.inc under -DHR_PHASE_MARK).  Counts are per wavefront pass = per sample slot."""
import collections, ctypes, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hyperreel_amd import build as B, config as C, plan


def emit(obj, path, out):
    for name, typ in obj._fields_:
        v = getattr(obj, name)
        p = f'{path}.{name}'
        if isinstance(v, ctypes.Structure):
            emit(v, p, out)
        elif isinstance(v, ctypes.Array):
            for i, e in enumerate(v):
                if isinstance(e, ctypes.Structure):
                    emit(e, f'{p}[{i}]', out)
                else:
                    out.append(f'    {p}[{i}] = {lit(e)};')
        else:
            out.append(f'    {p} = {lit(v)};')


def agg(obj):
    """aggregate initializer of a ctypes struct / array, in declaration order"""
    if isinstance(obj, ctypes.Structure):
        return '{' + ', '.join(agg(getattr(obj, n)) for n, _ in obj._fields_) + '}'
    if isinstance(obj, ctypes.Array):
        return '{' + ', '.join(agg(e) for e in obj) + '}'
    return lit(obj)


def lit(v):
    if isinstance(v, float):
        if v != v: return 'NAN'
        if v in (float('inf'), float('-inf')): return ('-' if v < 0 else '') + 'INFINITY'
        return f'{float.hex(v)}f'
    return str(int(v))


def main():
    model = next((a for a in sys.argv[1:] if not a.startswith('-')), 'donerf_sphere')
    half = '--fp16' in sys.argv
    cfg, ds = C.model_config(model), C.dataset_scalars(model)
    from hyperreel_amd import scenes
    sd = scenes.make_state_dict(cfg, ds, [64, 64, 64], seed=1)
    grid = [int(v) for v in sd['model.color_model.net.gridSize']]
    hc = plan.compile_config(cfg, ds, grid, grid_dtype='fp16' if half else 'fp32')
    # what analyse_live_columns (csrc/api.hip) does to preds_per_z / offsets is not replicated: dead columns only change P
    lines = []
    emit(hc, 'c', lines)
    Z = hc.z_channels
    ZP = 8
    while ZP < Z: ZP *= 2
    # plane descriptors as hr_model_finalize builds them
    MAT = [(0, 1), (0, 2), (1, 2)]; VEC = [2, 1, 0]
    pl = []
    app_off = real_off = 0
    for j in range(3):
        nd, na = hc.n_den[j], hc.n_app[j]
        if hc.video and nd == 0: na = 0
        cd4, ca4 = (nd + 3) // 4, (na + 3) // 4
        tex = 4 * (cd4 + ca4)
        if half: tex = (tex + 7) & ~7
        bw, bh = (hc.grid[VEC[j]] if hc.video else 1), (hc.num_keyframes if hc.video else hc.grid[VEC[j]])
        pl.append(f'    b.planes[{j}].tex = {tex}; b.planes[{j}].aw = {hc.grid[MAT[j][0]]}; b.planes[{j}].ah = {hc.grid[MAT[j][1]]}; '
                  f'b.planes[{j}].bw = {bw}; b.planes[{j}].bh = {bh}; b.planes[{j}].cd4 = {cd4}; b.planes[{j}].ca4 = {ca4}; '
                  f'b.planes[{j}].app_off = {app_off}; b.planes[{j}].app_real = {na}; b.planes[{j}].app_real_off = {real_off}; '
                  f'b.planes[{j}].a = a.planes[{j}].a; b.planes[{j}].b = a.planes[{j}].b;')
        app_off += 4 * ca4; real_off += na
    nd_, na_ = list(hc.n_den)[:3], list(hc.n_app)[:3]
    pclass = 1 if (nd_ == [8, 4, 4] and na_ == [8, 4, 4]) else (2 if (nd_ == [8, 0, 0] and na_[0] == 8) else 0)
    src = f'''#define HR_PHASE_MARK
#include "{B.CSRC}/sample_core.inc"
__global__ __launch_bounds__(256, 3) void phase_kernel(const HrSampleArgs a)
{{
    HrSampleArgs b = HrSampleArgs();
    b.rays = a.rays; b.rgb = a.rgb; b.basis = a.basis; b.n_rays = a.n_rays; b.nq = {(Z * hc.preds_per_z + 3) // 4}; b.cfg_dev = nullptr; b.head = nullptr;
    static constexpr hr_config c = {agg(hc)};
{chr(10).join(pl)}
    b.ca_total = {app_off}; b.n_basis_cols = {real_off}; b.rows_per_ray = 1; b.rows_out = nullptr; b.color_table = nullptr;
    b.fields = hr_fields();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, rib = tid / {ZP}, k = tid % {ZP};
    const int64_t ray = (int64_t)blockIdx.x * {256 // ZP} + rib;
    const bool ray_ok = ray < b.n_rays;
    __builtin_amdgcn_sched_barrier(0); asm volatile("; HRPHASE 100" ::: "memory");
    // the ray record as the stand-alone kernel makes it (sample_kernel.hip): one lane per ray of the workgroup -- executed by wavefront 0 only
    __shared__ __attribute__((aligned(16))) float s_ray[{256 // ZP} * HR_RAY_RECORD];
    if (tid < {256 // ZP}) {{
        const int64_t ray_r = (int64_t)blockIdx.x * {256 // ZP} + tid;
        HrRayLane R = hr_load_ray(c, b, ray_r, ray_r < b.n_rays);
        hr_ray_constants(c, R);
        hr_store_ray_record(R, s_ray + tid * HR_RAY_RECORD);
    }}
    __builtin_amdgcn_sched_barrier(0); asm volatile("; HRPHASE 101" ::: "memory");
    float* M = lds + 4096 + rib * 3 * b.ca_total;
    {{
        HrRayLane V = hr_load_ray(c, b, 0, false);
        if (c.shading == HR_SHADING_SH && ray_ok) {{ const float* r = b.rays + ray * c.ray_dim; V.vd[0] = r[3]; V.vd[1] = r[4]; V.vd[2] = r[5]; }}
        if (c.shading == HR_SHADING_SH || rib == 0) hr_fill_decode<{ZP}>(c, b, V, k, M);
    }}
    __shared__ __attribute__((aligned(16))) float s_ones[HR_GATHER_ONES];
    hr_gather_ones_init(s_ones);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0); asm volatile("; HRPHASE 102" ::: "memory");
    const HrRayLane L = hr_read_ray_record(s_ray + rib * HR_RAY_RECORD);
    __builtin_assume(b.rows_per_ray == 1);
    hr_sample_body<{ZP}, {'true' if half else 'false'}, HR_PHASE_PIPE, HR_PHASE_NB, {pclass}>(c, b, L, ray, ray_ok, k, lds + rib * b.nq * 4, b.nq * 4, M, s_ones, nullptr);
    __builtin_amdgcn_sched_barrier(0); asm volatile("; HRPHASE 199" ::: "memory");
}}
'''
    out = os.path.join(ROOT, 'tools', '_bin')
    os.makedirs(out, exist_ok=True)
    cu = os.path.join(out, f'phase_{model}.hip')
    open(cu, 'w').write(src)
    asm = os.path.join(out, f'phase_{model}.s')
    nb = 4 if hc.video else 2
    subprocess.run([B.hipcc(), *B.FLAGS, f'-DHR_PHASE_PIPE=1', f'-DHR_PHASE_NB={nb}', '-S', '--cuda-device-only', cu, '-o', asm], check=True)
    text = open(asm).read()
    body = text[text.index('_Z12phase_kernel12HrSampleArgs:'):text.index('.Lfunc_end0')]
    names = {100: 'ray record (wavefront 0)', 101: 'decode matrix (ray 0)', 102: '(record read)', 0: 'distance', 1: 'sort', 2: 'point+delta', 3: 'valid+taps', 4: 'gather plane 0',
             5: 'gather plane 1', 6: 'gather plane 2', 7: 'alpha+transmittance', 8: 'colour+sum+store', 199: 'end'}
    cur, counts = None, collections.OrderedDict()
    for ln in body.split('\n'):
        m = re.search(r'; HRPHASE (\d+)', ln)
        if m:
            cur = int(m.group(1)); counts.setdefault(cur, collections.Counter()); continue
        t = ln.strip()
        if cur is None or not t or t.startswith(('.', ';')) or t.endswith(':'): continue
        op = t.split()[0]
        c = counts[cur]
        c['all'] += 1
        if op.startswith('v_'):
            c['valu'] += 1
            if op.startswith('v_pk_'): c['pk'] += 1
            if 'dpp' in t: c['dpp'] += 1
            if op.split('_')[1] in ('exp', 'log', 'rcp', 'rsq', 'sqrt', 'sin', 'cos'): c['trans'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        elif op.startswith('ds_'): c['lds'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['vmem'] += 1
    # a phase's instructions are those AFTER its marker: marker n closes phase n (HR_SPH(n) is placed at the END of phase n)
    order = [100, 101, 102, 0, 1, 2, 3, 4, 5, 6, 7, 8]
    print(f'{model} ({"fp16" if half else "fp32"} texels), Z={Z}: instructions per wavefront pass (static count of the folded code; loops counted once)')
    print(f'{"phase":24s} {"VALU":>6s} {"(packed":>8s} {"dpp":>5s} {"trans)":>7s} {"SALU":>6s} {"LDS":>5s} {"VMEM":>5s}')
    keys = list(counts.keys())
    tot = collections.Counter()
    # instructions between marker[i] and marker[i+1] belong to the phase that marker[i+1] closes, except the 10x markers which open
    seq = keys
    for i, kmark in enumerate(seq):
        c = counts[kmark]
        nxt = seq[i + 1] if i + 1 < len(seq) else None
        label = names.get(kmark if kmark >= 100 else (nxt if nxt is not None else kmark), str(kmark))
        if kmark in (100, 101): label = names[kmark]
        elif kmark == 102: label = names.get(seq[i + 1], '?') if i + 1 < len(seq) else '?'
        else: label = names.get(seq[i + 1], 'tail') if i + 1 < len(seq) and seq[i + 1] < 100 else 'tail (fields, rows)'
        if kmark == 199: continue
        print(f'{label:24s} {c["valu"]:6d} {c["pk"]:8d} {c["dpp"]:5d} {c["trans"]:7d} {c["salu"]:6d} {c["lds"]:5d} {c["vmem"]:5d}')
        tot.update(c)
    print(f'{"total":24s} {tot["valu"]:6d} {tot["pk"]:8d} {tot["dpp"]:5d} {tot["trans"]:7d} {tot["salu"]:6d} {tot["lds"]:5d} {tot["vmem"]:5d}')


if __name__ == '__main__':
    main()
