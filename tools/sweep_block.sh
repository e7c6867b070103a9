#!/bin/bash
# eigensolver block size (guard vectors beyond the wanted modes) against passes/s: tools/sweep_block.sh 80 72 76 88
for b in "$@"; do
  python3 bench.py --no-cpu-baseline --block $b --steps 6 2>/dev/null > /tmp/sweep_block_$b.json || { echo "block $b: bench failed"; tail -3 /tmp/sweep_block_$b.json; continue; }
  python3 - $b <<'PY'
import json, sys
b = sys.argv[1]
d = json.loads(open(f"/tmp/sweep_block_{b}.json").read().strip().splitlines()[-1])
print(f"block {b}: {d['value']:.2f} passes/s, {d['ms_per_step']:.1f} ms per step of {d['config'].get('hypotheses_per_step', '?')} hypotheses")
PY
done
