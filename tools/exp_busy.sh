#!/bin/bash
export TMPDIR=/tmp
rm -rf /tmp/prof_busy
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_busy -o bench -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 1 > /tmp/busy.log 2>&1
python3 tools/gpu_busy.py /tmp/prof_busy 0.5
python3 tools/gpu_busy.py /tmp/prof_busy 0.3
tail -c 300 /tmp/busy.log | head -c 0
python3 - <<'PY'
import json
for l in open('/tmp/busy.log'):
    if l.startswith('{'):
        d=json.loads(l); print('profiled bench value', d['value'])
PY
