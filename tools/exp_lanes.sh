#!/bin/bash
# throughput against the number of hypothesis lanes (and hypotheses per step) - is 4 still the right default?
for cfg in "4 8" "6 12" "8 8" "8 16" "2 8" "1 8"; do
  set -- $cfg
  python bench.py --steps 6 --warmup 2 --lanes $1 --hyp-per-gpu $2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lanes $1 hyp $2: %.2f passes/s  (%.1f ms/step)' % (d['value'], d['ms_per_step']), flush=True)"
done
