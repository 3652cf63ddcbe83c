#!/bin/bash
# round-2 GPU call A: frame-kernel correctness, first timings, head error of the MLP modes, full-size flip counts
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_frame_kernel.py -x -q -p no:cacheprovider > $O/frame_tests.log 2>&1; echo "frame tests rc=$?" | tee -a $O/summary.txt
for v in "--sample-waves 8" "--sample-waves 4" "--no-frame-kernel"; do
  timeout 300 python bench.py --steps 30 --warmup 10 --cpu-sample 0 --no-stage-timing $v > $O/bench_$(echo $v | tr -d ' -').json 2>$O/bench_err.log; echo "bench $v rc=$?" | tee -a $O/summary.txt
done
for m in donerf_sphere neural_3d_z_plane immersive_sphere; do timeout 300 python tools/head_error.py $m >> $O/head_error.txt 2>&1; done
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "full_size" > $O/full_size.log 2>&1; echo "full size rc=$?" | tee -a $O/summary.txt
tail -3 $O/frame_tests.log; grep -h '"value"' $O/bench_*.json | sed -e 's/"config".*//' ; cat $O/head_error.txt; tail -15 $O/full_size.log
