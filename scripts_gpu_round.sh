#!/bin/bash
# One GPU session: tests, smoke, bench (both MLP precisions), rocprof kernel trace.  Logs -> gpurun_out/
set -u
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q --maxfail=60 2>&1 | tail -70 ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -20 ) > gpurun_out/smoke.log 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -3 ) > gpurun_out/bench.log 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 --mlp-precision fp32 --cpu-sample 0 2>&1 | tail -3 ) > gpurun_out/bench_fp32.log 2>&1
cd /tmp && ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o rp -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-stage-timing 2>&1 | grep -v simple_timer | tail -5 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find /tmp/prof -name "*stats*.csv"); do cp $f gpurun_out/prof/; done
for f in $(find /tmp/prof -name "*kernel_trace*.csv"); do head -200 $f > gpurun_out/prof/$(basename $f); done
tail -40 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log; cat gpurun_out/bench.log; cat gpurun_out/bench_fp32.log; cat gpurun_out/prof/rp_kernel_stats.csv
