#!/bin/bash
# (round 3; the A/B library tools/libds_nopk.so was `make -C diffsound_amd/csrc BUILD=build_nopk LIB=../../tools/libds_nopk.so` of the
# tree BEFORE plain FMAs became the only form - with `EXTRA="-DDS_PLAIN_FMA=1 -Xclang -target-feature -Xclang -packed-fp32-ops"` - loaded
# through DS_EXP_LIB; kept as the record of how profiles/r03_no_packed_fp32_ab.txt was taken)
# A/B: the production library against the build without packed FP32 (tools/libds_nopk.so: -DDS_PLAIN_FMA=1 and the
# compiler's packed-fp32-ops feature off)
for lib in "" "tools/libds_nopk.so"; do
  export DS_EXP_LIB=$lib
  echo "== lib: ${lib:-production}"
  python tools/mb_mfma_term.py 26 80 2>&1 | grep "term"
  python tools/mb_gram_mix.py 2>&1 | tail -n 12
  for i in 1 2; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --amortised-cycle 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('bench: %.2f passes/s; K W %.1f us; fused term solo %.1f us; loss %.16g' % (d['value'], r['lobpcg_spmm']['avg_launch_ms']*1e3, r['avg_launch_ms']*1e3, d['loss_sum_last_step']), flush=True)"
  done
done
export DS_EXP_LIB=tools/libds_nopk.so
echo "== interference matrix with the no-packed-FP32 build as the production victim"
timeout -k 10 400 python tools/mfma_interference.py 3 2>&1 | grep "production"
