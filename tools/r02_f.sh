#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES"
for m in 0 1 3 7 15 31 63 127; do
  echo "== dbg $m"; HR_SAMPLE_DBG=$m bash tools/pmc_one.sh f_$m "$C" --no-frame-kernel --lib $PWD/tools/_bin/libhr_tuning.so 2>&1 | grep -A6 "hr_sample_kernel" | grep "VALU\|SALU\|LDS\|VMEM"
done
