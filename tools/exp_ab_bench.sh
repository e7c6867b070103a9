#!/bin/bash
# A/B of library builds at the bench level on one box: tools/exp_ab_bench.sh libA.so - libA.so - ...  ("-" = the product build)
for l in "$@"; do
  if [ "$l" = "-" ]; then r=$(python3 bench.py --no-cpu-baseline --steps 6 2>/dev/null)
  else r=$(DS_EXP_LIB=$PWD/diffsound_amd/csrc/$l python3 bench.py --no-cpu-baseline --steps 6 2>/dev/null); fi
  echo "$l $(echo "$r" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value'],2),'passes/s; term',round(r['avg_launch_ms']*1e3,1),'us; K W',round(r['lobpcg_spmm']['avg_launch_ms']*1e3,1),'us =',round(r['lobpcg_spmm']['frac_of_stream'],3),'of STREAM')")"
done
