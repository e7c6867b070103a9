#!/bin/bash
# Round evidence: bench JSON (with the CPU baseline) + rocprofv3 kernel stats of the default run and of a one-lane run.
# Run on the GPU box from the repo root; summaries land in gpurun_out/ (copy them to profiles/ afterwards).
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py > gpurun_out/r01_bench_n1.json 2> gpurun_out/bench_stderr.log
tail -c 600 gpurun_out/r01_bench_n1.json; echo
rm -rf /tmp/prof_b /tmp/prof_l1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
python3 tools/summarize_prof.py /tmp/prof_b gpurun_out/r01_bench_kernel_stats.csv --top 45
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l1 -o bench -- python3 bench.py --no-cpu-baseline --lanes 1 --hyp-per-gpu 2 > gpurun_out/bench_prof_l1.log 2>&1
python3 tools/summarize_prof.py /tmp/prof_l1 gpurun_out/r01_bench_lanes1_kernel_stats.csv --top 45
