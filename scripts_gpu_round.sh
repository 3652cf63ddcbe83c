#!/bin/bash
# One GPU session: tests, smoke, bench, rocprof kernel trace.  Logs -> gpurun_out/
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q --maxfail=60 -x 2>&1 | tail -60 ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -20 ) > gpurun_out/smoke.log 2>&1
( timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -20 ) > gpurun_out/bench.log 2>&1
cd /tmp && ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-stage-timing 2>&1 | tail -5 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof && find /tmp/prof -type f | head -20 >> gpurun_out/rocprof.log
for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp $f gpurun_out/prof/; done
for f in $(find /tmp/prof -name "*kernel_trace*.csv"); do head -400 $f > gpurun_out/prof/$(basename $f); done
tail -30 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log; cat gpurun_out/bench.log
