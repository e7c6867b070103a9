#!/bin/bash
# nested start / tolerance sweep of the bench (GPU box)
for args in "--tol 1e-5" "--tol 1e-5 --nested-tol 1e-2" "--tol 1e-5 --nested-tol 1e-3" "--tol 1e-5 --nested-tol 3e-2 --nested-maxit 4" "--nested-tol 1e-3"; do
  echo "== $args"
  python bench.py --no-cpu-baseline --steps 3 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],2), 'passes/s;', d['config']['eigensolver'][-150:], d['config']['convergence_gate'][-60:])"
done
