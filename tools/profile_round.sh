#!/bin/bash
# Round evidence: bench JSON (with the CPU baseline) + rocprofv3 kernel stats of the default run and of a one-lane run.
# Run on the GPU box from the repo root; summaries land in gpurun_out/ (copy them to profiles/ afterwards).
# usage: tools/profile_round.sh r02 [extra bench args]
set -e
tag=${1:-rXX}; shift || true
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py "$@" > gpurun_out/${tag}_bench_n1.json 2> gpurun_out/${tag}_bench_stderr.log
tail -c 400 gpurun_out/${tag}_bench_n1.json; echo
rm -rf /tmp/prof_b /tmp/prof_l1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_bench_prof.log 2>&1
python3 tools/summarize_prof.py /tmp/prof_b gpurun_out/${tag}_bench_kernel_stats.csv --top 45
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l1 -o bench -- python3 bench.py --no-cpu-baseline --lanes 1 --hyp-per-gpu 2 --steps 4 --warmup 1 "$@" > gpurun_out/${tag}_bench_prof_l1.log 2>&1
python3 tools/summarize_prof.py /tmp/prof_l1 gpurun_out/${tag}_bench_lanes1_kernel_stats.csv --top 45
