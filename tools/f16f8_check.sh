#!/bin/bash
# One-call device check of the experimental f16f8 arithmetic: tools/f16f8_check.sh [v1|v2]  (GPU box, ~1.5 min)
#   builds nothing -- run `python tools/build_variant.py f16f8 -DHR_WITH_F16F8` (v1) or
#   `python tools/build_variant.py f16f8v2 -DHR_WITH_F16F8 -DHR_F16F8_V2` (v2) first; the .so travels with the snapshot.
set -u
V=${1:-v2}
R=${GRAFT_REPO_ROOT:-/root/repo}
LIB=$R/tools/_bin/libhr_f16f8$([ "$V" = v2 ] && echo v2).so
OUT=$R/gpurun_out/f16f8_$V.txt
mkdir -p $R/gpurun_out; : > $OUT
cd $R
echo "# head error vs the exact fp32-MFMA chain" | tee -a $OUT
HR_LIB=$LIB timeout 120 python tools/head_error.py donerf_sphere 65536 f16f8 2>&1 | grep "head" | tee -a $OUT
echo "# fixtures + full-size frames" | tee -a $OUT
HR_LIB=$LIB timeout 200 python tools/f16f8_parity.py f16f8 2>&1 | grep "f16f8" | tee -a $OUT
echo "# bench: two kernels, frame kernel with 8 and 4 sample wavefronts (v2 only), next to f16x3 / f16x2 on the same box" | tee -a $OUT
for args in "--mlp-precision f16x3 --no-frame-kernel" "--mlp-precision f16f8 --no-frame-kernel" "--mlp-precision f16x2 --no-frame-kernel" \
            "--mlp-precision f16x3" "--mlp-precision f16f8" "--mlp-precision f16f8 --sample-waves 4" "--mlp-precision f16x2"; do
  timeout 100 python bench.py --lib $LIB $args --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$args ->', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d.get('stage_ms'), d['config']['execution'][:22], 'parity', d.get('parity_vs_oracle_linf'), 'over', d.get('parity_rays_over_1e-4'))" | tee -a $OUT
done
