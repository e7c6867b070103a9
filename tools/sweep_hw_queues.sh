#!/bin/bash
# HIP streams are multiplexed onto GPU_MAX_HW_QUEUES hardware queues (ROCclr default 4): passes/s of the default bench against it.
# tools/sweep_hw_queues.sh default 8 default 8 2 16
for q in "$@"; do
  if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  python3 bench.py --no-cpu-baseline --steps 8 2>/dev/null > /tmp/sweep_q.json || { echo "queues $q: bench failed"; continue; }
  python3 - "$q" <<'PY'
import json, sys
d = json.loads(open("/tmp/sweep_q.json").read().strip().splitlines()[-1])
print(f"GPU_MAX_HW_QUEUES {sys.argv[1]:>7s}: {d['value']:.2f} passes/s, {d['ms_per_step']:.1f} ms per step, loss {d.get('loss_sum_last_step')}")
PY
done
