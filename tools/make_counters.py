"""gpurun_out/pmc_<tag>_{1..5}.txt (tools/pmc.sh) -> a counters JSON that bench.py attaches to its roofline objects.
python tools/make_counters.py <tag> <out.json> <model> <mlp_precision> <grid_dtype> <rays_per_launch> <gx> <gy> <gz>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, out, model, prec, gdt, rpl = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6])
grid = [int(v) for v in sys.argv[7:10]]
vals = {}
for i in range(1, 6):
    p = os.path.join(ROOT, 'gpurun_out', f'pmc_{tag}_{i}.txt')
    if not os.path.exists(p):
        continue
    k = None
    for ln in open(p):
        m = re.match(r'^(?:void )?(hr_\w+)', ln)
        if m and 'dispatches' in ln:
            k = m.group(1); vals.setdefault(k, {}); continue
        m = re.match(r'^\s+(\w+)\s+total\s+\S+\s+per-dispatch\s+(\S+)', ln)
        if m and k:
            vals[k][m.group(1)] = float(m.group(2))
res = {'_about': 'per-launch PMC counters of the render kernels (tools/pmc.sh: separate rocprofv3 --pmc passes, --kernel-trace only; raw pass outputs in '
                 'profiles/r03_*_pmc_pass*.txt).  traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: both counters are in KB, and FETCH_SIZE under-counts '
                 'wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated; Infinity-Cache hits are included in '
                 'these fabric-side counters.  SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE cycles.',
       'workload': {'model': model, 'rays_per_launch': rpl, 'grid': grid, 'mlp_precision': prec, 'grid_dtype': gdt},
       'csrc_hash': __import__('hyperreel_amd.build', fromlist=['csrc_hash']).csrc_hash()}
for k, v in vals.items():
    if not k.startswith(('hr_mlp', 'hr_sample', 'hr_frame')):
        continue
    e = {kk: vv for kk, vv in v.items()}
    if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        e['traffic_bytes'] = int((2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024)
    cyc = v.get('GRBM_GUI_ACTIVE')
    if cyc:
        simd_cycles = cyc / 8.0 * 1024           # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
        if 'SQ_INSTS_VALU' in v:
            e['valu_insts'] = v['SQ_INSTS_VALU']
            e['valu_busy_frac'] = round(v['SQ_INSTS_VALU'] * 4.0 / simd_cycles, 4)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
            e['mfma_busy_frac'] = round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles, 4)
        if 'TA_TA_BUSY_sum' in v:
            e['ta_busy_frac'] = round(v['TA_TA_BUSY_sum'] / (cyc / 8.0 * 256), 4)
    if 'SQ_INSTS_VALU' in v and 'SQ_WAVES' in v and v['SQ_WAVES'] > 0:
        e['valu_insts_per_wave'] = round(v['SQ_INSTS_VALU'] / v['SQ_WAVES'], 1)
    if 'SQ_LDS_BANK_CONFLICT' in v and v.get('SQ_LDS_IDX_ACTIVE'):
        e['lds_conflict_frac'] = round(v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE'], 4)
    if 'TCC_HIT_sum' in v and (v['TCC_HIT_sum'] + v.get('TCC_MISS_sum', 0)) > 0:
        e['l2_hit_rate'] = round(v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']), 4)
    lim = []
    if 'valu_busy_frac' in e: lim.append(f"VALU issue {100 * e['valu_busy_frac']:.0f} %")
    if 'mfma_busy_frac' in e and e['mfma_busy_frac'] > 0: lim.append(f"matrix pipe busy {100 * e['mfma_busy_frac']:.0f} %")
    if 'ta_busy_frac' in e: lim.append(f"address unit busy {100 * e['ta_busy_frac']:.0f} %")
    if lim: e['limiter'] = ', '.join(lim) + ' of the kernel\'s cycles'
    res[k] = e
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ('traffic_bytes', 'valu_busy_frac', 'mfma_busy_frac', 'ta_busy_frac', 'valu_insts_per_wave', 'lds_conflict_frac', 'l2_hit_rate')}
                  for k, v in res.items() if isinstance(v, dict) and k.startswith('hr_')}, indent=1))
