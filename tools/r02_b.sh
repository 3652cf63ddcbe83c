#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r02b; O=gpurun_out/r02b
export PYTHONUNBUFFERED=1
for w in 8 4; do for m in 0 1 2; do timeout 120 python tools/frame_stats.py $w $m >> $O/stats.txt 2>>$O/err.log; done; done
timeout 120 python tools/frame_stats.py 8 0 bf16x3 >> $O/stats.txt 2>>$O/err.log
timeout 900 python -m pytest tests/test_gpu_frame_kernel.py -x -q -p no:cacheprovider > $O/frame_tests.log 2>&1; echo "frame tests rc=$?" >> $O/stats.txt
cat $O/stats.txt; tail -3 $O/frame_tests.log; tail -5 $O/err.log
