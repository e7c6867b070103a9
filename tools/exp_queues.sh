#!/bin/bash
# hardware queues against throughput and lane fairness (8 hypothesis lanes = 8 HIP streams; default 4 queues)
for rep in 1 2; do
for q in 4 8; do
  echo -n "GPU_MAX_HW_QUEUES=$q lanes 8 => "
  GPU_MAX_HW_QUEUES=$q python3 bench.py --no-cpu-baseline --steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2))"
done
done
GPU_MAX_HW_QUEUES=8 python3 tools/lane_tail.py 2>&1 | tail -n 1
