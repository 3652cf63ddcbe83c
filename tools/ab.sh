#!/bin/bash
# A/B two builds of the library: tools/ab.sh "<extra hipcc flags A>" "<extra hipcc flags B>" [bench args]
# builds both variants on the GPU box (hipcc is there), then alternates runs.
fa="$1"; fb="$2"; shift 2
python - "$fa" "$fb" <<'PY'
import sys, shutil
sys.path.insert(0, '.')
from hyperreel_amd import build
for tag, fl in (('A', sys.argv[1]), ('B', sys.argv[2])):
    p = build.build(force=True, extra_flags=fl.split())
    shutil.copy(p, f'/tmp/lib{tag}.so')
PY
for i in 1 2 3; do
  for t in A B; do cp /tmp/lib$t.so hyperreel_amd/_build/libhyperreel_hip.so; touch hyperreel_amd/_build/libhyperreel_hip.so; echo -n "$t: "; python tools/sweep.py "--chunk 131072 $*" ; done
done
