#!/bin/bash
# two bench processes sharing the GPU: is one process host-bound?
python bench.py --no-cpu-baseline --steps 6 > /tmp/b1.json 2>/dev/null &
p1=$!
python bench.py --no-cpu-baseline --steps 6 > /tmp/b2.json 2>/dev/null &
p2=$!
wait $p1 $p2
python - <<'PY'
import json
v=[json.loads(open(f).read().strip().splitlines()[-1])['value'] for f in ('/tmp/b1.json','/tmp/b2.json')]
print('two processes:', v, 'sum', sum(v))
PY
python bench.py --no-cpu-baseline --steps 6 2>/dev/null | python -c "
import json,sys; print('one process:', json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"
