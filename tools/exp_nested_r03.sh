#!/bin/bash
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --amortised-cycle 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d['config']['eigensolver']
print('%-60s %.2f passes/s  %s' % (' '.join(sys.argv[1:]), d['value'], e[e.find('mean iterations'):e.find('mean iterations')+78]), flush=True)" "$@"; }
run --nested-tol 5e-3
run --nested-tol 3e-3
run --nested-tol 2e-3 --nested-maxit 12
run --nested-tol 1e-3 --nested-maxit 12
run --nested-tol 5e-4 --nested-maxit 16
run --nested-tol 3e-3 --block 76
run --nested-tol 3e-3 --smooth-degree 2
run --nested-tol 3e-3 --coarse-degree 26 --coarse-ratio 450
run --nested-tol 3e-3
