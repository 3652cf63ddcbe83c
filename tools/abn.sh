#!/bin/bash
# A/B/C...: tools/abn.sh "<hipcc flags 1>" "<hipcc flags 2>" ... -- [bench args]
# builds every variant on the GPU box (hipcc is there), then alternates bench runs (3 rounds).
flags=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do flags+=("$1"); shift; done
[ "$1" == "--" ] && shift
i=0
for fl in "${flags[@]}"; do
  python - "$fl" "$i" <<'PY'
import sys, shutil
sys.path.insert(0, '.')
from hyperreel_amd import build
p = build.build(force=True, extra_flags=sys.argv[1].split())
shutil.copy(p, f'/tmp/lib{sys.argv[2]}.so')
PY
  i=$((i+1))
done
for r in 1 2 3; do
  i=0
  for fl in "${flags[@]}"; do
    cp /tmp/lib$i.so hyperreel_amd/_build/libhyperreel_hip.so; touch hyperreel_amd/_build/libhyperreel_hip.so
    echo -n "[$fl] "
    python bench.py --cpu-sample 0 $* 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('stage_ms'))"
    i=$((i+1))
  done
done
cp /tmp/lib0.so hyperreel_amd/_build/libhyperreel_hip.so
