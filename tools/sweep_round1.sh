#!/bin/bash
# Parameter sweep of the two-level preconditioner / block width on the benchmark workload (passes/s, mean iterations).
run() { python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value'],2), d['config']['eigensolver'].split('mean iterations')[-1])"; }
run
run --smooth-degree 2
run --coarse-degree 16 --coarse-ratio 200
run --coarse-degree 20 --coarse-ratio 300
run --coarse-degree 32 --coarse-ratio 700
run --smooth-degree 4
run --smooth-degree 2 --coarse-degree 32 --coarse-ratio 700
run --rr-refresh 16
run --block 72
run --block 76
run
