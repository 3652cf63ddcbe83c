#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r02c; O=gpurun_out/r02c
export PYTHONUNBUFFERED=1
for m in 0 2; do timeout 120 python tools/frame_stats.py 8 $m >> $O/stats.txt 2>>$O/err.log; done
cat $O/stats.txt; tail -3 $O/err.log
