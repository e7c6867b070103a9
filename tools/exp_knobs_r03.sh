#!/bin/bash
# re-tune the V-cycle and the block width for the round-3 kernels (fine MFMA term 145 us, corner-level term 17.7 us)
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --amortised-cycle 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d['config']['eigensolver']
print('%-60s %.2f passes/s  %s' % (' '.join(sys.argv[1:]), d['value'], e[e.find('mean iterations'):e.find('mean iterations')+70]), flush=True)" "$@"; }
run
run --block 72
run --block 76
run --block 88
run --smooth-degree 4
run --smooth-degree 2
run --coarse-degree 26 --coarse-ratio 450
run --coarse-degree 30 --coarse-ratio 600
run --coarse-degree 18 --coarse-ratio 250
run --smooth-degree 4 --coarse-degree 26 --coarse-ratio 450
run --nested-tol 3e-3
run --nested-tol 3e-2
run
