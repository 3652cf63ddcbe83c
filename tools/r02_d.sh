#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r02d; O=gpurun_out/r02d
export PYTHONUNBUFFERED=1
for w in 8 4; do for m in 0 2; do timeout 120 python tools/frame_stats.py $w $m >> $O/stats.txt 2>>$O/err.log; done; done
for v in "--sample-waves 8" "--sample-waves 4" "--no-frame-kernel"; do
  timeout 300 python bench.py --steps 30 --warmup 10 --cpu-sample 0 --no-stage-timing $v 2>>$O/err.log | sed -e 's/"config".*//' >> $O/stats.txt
done
timeout 900 python -m pytest tests/test_gpu_frame_kernel.py -x -q -p no:cacheprovider > $O/frame_tests.log 2>&1; echo "frame tests rc=$?" >> $O/stats.txt
cat $O/stats.txt; tail -3 $O/frame_tests.log; grep -v amdgpu.ids $O/err.log | tail -5
