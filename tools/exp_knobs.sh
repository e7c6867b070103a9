#!/bin/bash
# knob sweep of the bench (GPU box): one line per setting
run() {
  echo -n "$* => "
  python bench.py --no-cpu-baseline --steps 4 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d['config']['eigensolver']
print(round(d['value'],2), 'passes/s;', e[e.find('mean iterations'):][:60])"
}
for a in "$@"; do run $a; done
