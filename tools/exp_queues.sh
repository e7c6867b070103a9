#!/bin/bash
for q in 2 4 8 16; do
  echo -n "GPU_MAX_HW_QUEUES=$q lanes 8 => "
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --steps 4 --lanes 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2))"
done
echo -n "HIP_FORCE_DEV_KERNARG=1 lanes 8 => "
HIP_FORCE_DEV_KERNARG=1 python bench.py --no-cpu-baseline --steps 4 --lanes 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2))"
