#!/bin/bash
# knob sweep of the bench (GPU box): one line per setting
run() {
  echo -n "$* => "
  python bench.py --no-cpu-baseline --steps 4 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d['config']['eigensolver']
print(round(d['value'],2), 'passes/s;', e[e.find('mean iterations'):])"
}
run
run --smooth-degree 1
run --smooth-degree 3
run --block 72
run --block 88
run --coarse-degree 20 --coarse-ratio 300
run --coarse-degree 36 --coarse-ratio 800
run --nested-tol 3e-3
run --nested-tol 3e-2
run --lanes 3
run --lanes 6
run --hyp-per-gpu 16 --lanes 4
