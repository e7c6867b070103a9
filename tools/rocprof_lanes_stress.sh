#!/bin/bash
# about one in ten profiled 8-lane runs of round 4 ended in a SIGSEGV inside the runtime's launch path under
# rocprofv3's hooks.  N profiled 8-lane runs on one lease with python's faulthandler on (a crash prints every thread's Python
# stack); summary -> gpurun_out/<tag>_rocprof_lanes_stress.txt.     tools/rocprof_lanes_stress.sh <tag> [N]
tag=${1:-rXX}; n=${2:-12}
export TMPDIR=/tmp
out=gpurun_out/${tag}_rocprof_lanes_stress.txt
: > $out
fail=0
for i in $(seq 1 $n); do
  rm -rf /tmp/prof_s
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o b -- python3 -X faulthandler bench.py --no-cpu-baseline --no-solo --amortised-cycle 0 --steps 6 > /tmp/stress_$i.out 2> /tmp/stress_$i.err
  rc=$?
  val=$(grep -o '"value": [0-9.]*' /tmp/stress_$i.out | head -1)
  echo "run $i: rc=$rc $val" | tee -a $out
  if [ $rc -ne 0 ]; then fail=$((fail+1)); echo "---- stderr of run $i (tail) ----" >> $out; grep -v amdgpu.ids /tmp/stress_$i.err | tail -n 60 >> $out; fi
done
echo "$fail of $n profiled 8-lane runs failed" | tee -a $out
