for r in 1 2 3; do for l in product tools/_bin/libhr_k2old.so; do
  if [ "$l" = product ]; then a=""; else a="--lib $l"; fi
  python bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 --frame-kernel $a 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('frame-kernel $l', d['value'], d['ms_per_step'])"
done; done
