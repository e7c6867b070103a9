#!/bin/bash
# Sweep of the two-level preconditioner's degrees on the benchmark (8 lanes), interleaved repeats on ONE box:
#   tools/sweep_precond.sh > gpurun_out/sweep_precond.txt
# Columns: coarse-degree coarse-ratio smooth-degree | passes/s | mean fine iterations.
for rep in 1 2; do
for cfg in "22 350 3" "18 350 3" "20 350 3" "16 250 3" "18 250 3" "22 350 2" "22 350 4" "26 450 3" "20 300 3"; do
set -- $cfg
python3 bench.py --no-cpu-baseline --no-solo --amortised-cycle 1 --coarse-degree $1 --coarse-ratio $2 --smooth-degree $3 2> /tmp/sweep_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['config']
import re
e=c['eigensolver']; m=re.search(r'mean iterations ([0-9.]+)',e); k=re.search(r'mean ([0-9.]+) corner',e)
print('$cfg |', round(d['value'],2), '| fine', m.group(1), 'corner', k.group(1) if k else '-')
" || { tail -3 /tmp/sweep_err.txt; }
done
done
