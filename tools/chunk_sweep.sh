#!/bin/bash
# headline frame time against the rays per workspace chunk (0 = the library's default, 131 072): tools/chunk_sweep.sh 0 160000 ...
for c in "$@"; do
  python bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 --no-stage-timing --chunk $c 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chunk $c', d['value'], d['ms_per_step'], d['windows_ms_per_step'])"
done
