#!/bin/bash
# N bench processes sharing the GPU: usage exp_two_procs.sh NPROC [bench args]
np=${1:-2}; shift
pids=()
for i in $(seq 1 $np); do
  python bench.py --no-cpu-baseline --steps 6 "$@" > /tmp/b$i.json 2>/dev/null &
  pids+=($!)
done
wait "${pids[@]}"
python - "$np" <<'PY'
import json,sys
n=int(sys.argv[1])
v=[json.loads(open(f'/tmp/b{i}.json').read().strip().splitlines()[-1])['value'] for i in range(1,n+1)]
print(n, 'processes:', [round(x,2) for x in v], 'sum', round(sum(v),2))
PY
