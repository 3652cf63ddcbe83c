#!/bin/bash
# one PMC pass over a short bench run: tools/pmc_one.sh <tag> "<counters>" [bench args...] -> gpurun_out/pmc_<tag>.txt
set -u
tag=$1; ctrs=$2; shift; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp; rm -rf /tmp/pmc_$tag
timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_$tag -o p -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-stage-timing --no-extras "$@" > /tmp/pmc_$tag.log 2>&1
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then
python - "$f" <<'PY' > $R/gpurun_out/pmc_$tag.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][:70]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'], k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in acc:
    if 'hr_' not in k: continue
    print(k, 'dispatches', cnt[k])
    for c, v in acc[k].items():
        print(f'   {c:32s} total {v:.5g}  per-dispatch {v / cnt[k]:.5g}')
PY
else tail -5 /tmp/pmc_$tag.log > $R/gpurun_out/pmc_$tag.txt; fi
cat $R/gpurun_out/pmc_$tag.txt
