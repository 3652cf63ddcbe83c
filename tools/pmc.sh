#!/bin/bash
# PMC passes over a short bench run (counters in separate passes; kernel-trace only).
# usage: tools/pmc.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>_<pass>.txt
#        PMC_CMD="python tools/train_bench.py --model immersive_sphere --steps 3" tools/pmc.sh <tag>   (another command, from the repo root)
set -u
tag=$1; shift
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp
i=0
for ctrs in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "GRBM_GUI_ACTIVE FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
  "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" ; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_$i -o p -- bash -c "cd $GRAFT_REPO_ROOT && ${PMC_CMD:-python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-stage-timing --no-extras $*}" > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$ctrs" <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i.txt
import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'], k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in acc:
    if 'hr_' not in k: continue
    print(k, 'dispatches', cnt[k])
    for c, v in acc[k].items():
        print(f'   {c:32s} total {v:.4g}  per-dispatch {v / cnt[k]:.4g}')
PY
  else
    tail -5 /tmp/pmc_$i.log > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i.txt
  fi
done
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_*.txt
