#!/bin/bash
# tools/build_variants.sh name1 "<flags1>" name2 "<flags2>" ... : measurement builds into tools/_bin (in parallel, here; the .so travel with gpurun)
cd "$(dirname "$0")/.."
pids=()
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  ( python tools/build_variant.py "$n" $f > /tmp/bv_$n.log 2>&1 || { echo "FAILED $n"; tail -5 /tmp/bv_$n.log; } ) &
  pids+=($!)
done
wait "${pids[@]}"
ls -la tools/_bin/*.so
