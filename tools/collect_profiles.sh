#!/bin/bash
# Round evidence on the GPU box (run from the repo root): tools/collect_profiles.sh r02
#   <tag>_bench_n1.json                 bench.py default run (with the CPU baseline)
#   <tag>_bench_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the default run (4 lanes)
#   <tag>_bench_lanes1_kernel_stats.csv the same with one hypothesis at a time (clean per-kernel durations)
#   <tag>_spmm_pmc_<kind>.json          FETCH_SIZE / WRITE_SIZE per launch of the SpMM forms (separate passes): fused term fp32 / bf16 / mfma,
#                                       K W (kx), [K W | M W] (km), the fused residual (resid)
#   <tag>_gram_mix.txt                  Gram / mix timings at the solver's shapes
#   <tag>_rr_pmc_<shape>.json, <tag>_gram_mix_mfma_util.json   matrix-pipe counters of the Rayleigh-Ritz kernels per shape and the
#                                       derived utilisation (-> profiles/gram_mix_mfma_util.json, quoted by bench.py)
#   <tag>_symbolic_phase_timing.txt     tools/time_lift.py: ord-2 lifting + symbolic phase per topology, warm
#   <tag>_c5_bench.json / _c5_kernel_stats.csv   bench.py --workload c5 (configs[4]) under rocprofv3 --kernel-trace --stats
#   <tag>_mb_kx.txt / _mb_corner.txt / _m32_diag.txt   the eigensolver's own products alone, the corner-node level, per-wave cycles
# Copy what is to be judged into profiles/ - in particular <tag>_spmm_pmc_bytes_per_launch.json -> profiles/spmm_pmc_bytes_per_launch.json
# and <tag>_gram_mix_mfma_util.json -> profiles/gram_mix_mfma_util.json (the two records bench.py quotes, keyed by source hashes: collect
# part 2 AFTER the last change to the kernel sources / modal_ops.py, then part 1, so that the bench line carries them).
# A gpurun call is capped at 20 minutes: tools/collect_profiles.sh <tag> <part>, part = 1 (the bench line), 1b (rocprofv3 kernel
# tables of the 8-lane / one-lane / kernel-alone runs + busy fraction), 2 (PMC passes, microbenchmarks, diagnostics),
# 3 (configs[4]) or all.
set -e -o pipefail  # (a failing tool must not leave its traceback behind as evidence)
tag=${1:-rXX}
part=${2:-all}
case "$part" in 1|1b|2|3|all) ;; *) echo "collect_profiles.sh: unknown part '$part' (1, 1b, 2, 3 or all)" >&2; exit 2;; esac
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
if [ $part = 1 ] || [ $part = all ]; then
python3 bench.py > $out/${tag}_bench_n1.json 2> $out/${tag}_bench_stderr.log
echo "bench done"; tail -c 300 $out/${tag}_bench_n1.json; echo
fi
if [ $part = 1b ] || [ $part = all ]; then
rm -rf /tmp/prof_b /tmp/prof_l1
# (about one in ten profiled 8-lane runs ends in a SIGSEGV inside the runtime's launch path under the profiler's hooks -
# profiles/README.md, "A note on the profiled runs" - so this step gets one more try)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python3 bench.py --no-cpu-baseline --no-solo --no-api-path --amortised-cycle 1 --steps 8 > $out/${tag}_bench_prof.log 2>&1 || { echo "profiled run failed, once more"; rm -rf /tmp/prof_b; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python3 bench.py --no-cpu-baseline --no-solo --no-api-path --amortised-cycle 1 --steps 8 > $out/${tag}_bench_prof.log 2>&1; }
python3 tools/summarize_prof.py /tmp/prof_b $out/${tag}_bench_kernel_stats.csv --top 45
python3 tools/gpu_busy.py /tmp/prof_b 0.45 0.05 > $out/${tag}_gpu_busy.txt; cat $out/${tag}_gpu_busy.txt  # 55 % .. 95 % of the run: inside the timed steps, without the last step's drain
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l1 -o bench -- python3 bench.py --no-cpu-baseline --no-solo --no-api-path --lanes 1 --hyp-per-gpu 2 --steps 4 --warmup 1 > $out/${tag}_bench_prof_l1.log 2>&1
python3 tools/summarize_prof.py /tmp/prof_l1 $out/${tag}_bench_lanes1_kernel_stats.csv --top 45
# the kernel-alone measurements of the bench line (roofline.avg_launch_ms: hundreds of back-to-back launches after the timed region)
# under the profiler: a short run whose launches of the fused term, K W and the triad are almost all those
rm -rf /tmp/prof_solo
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_solo -o bench -- python3 bench.py --no-cpu-baseline --no-api-path --lanes 1 --hyp-per-gpu 1 --steps 1 --warmup 1 --amortised-cycle 1 > $out/${tag}_bench_prof_solo.log 2>&1
python3 tools/summarize_prof.py /tmp/prof_solo $out/${tag}_bench_solo_kernel_stats.csv --top 12
fi
if [ $part = 2 ] || [ $part = all ]; then
# PMC bytes of the fused term (two passes: the TCC counters do not fit one)
for kind in fp32 bf16 mfma kx km resid; do
  rm -rf /tmp/pmc_b
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --kernel-include-regex "spmm_union" --pmc $c --output-format csv -d /tmp/pmc_b/$c -o p -- python3 tools/mb_cheb_only.py 6 $kind > /tmp/pmc_log_$c.txt 2>&1 || { tail -5 /tmp/pmc_log_$c.txt; exit 1; }
    echo "$kind $c done"
  done
  kname="spmm_union_kernel<"; [ $kind = mfma ] && kname="spmm_union_mfma_kernel"
  python3 tools/pmc_summary.py /tmp/pmc_b "$kname" > $out/${tag}_spmm_pmc_$kind.json; cat $out/${tag}_spmm_pmc_$kind.json
done
python3 tools/pmc_bytes.py $tag > $out/${tag}_spmm_pmc_bytes_per_launch.json   # -> profiles/spmm_pmc_bytes_per_launch.json (keyed by the kernel-source hash)
python3 tools/time_lift.py > $out/${tag}_symbolic_phase_timing.txt 2>&1; cat $out/${tag}_symbolic_phase_timing.txt
# the eigensolver's own products alone on the device (VALU union kernel and the fp32 matrix-core form), the corner-node level,
# and the per-wave cycle breakdown of a diagnostic build (make -C diffsound_amd/csrc BUILD=/tmp/build_diag LIB=libds_m32diag.so
# EXTRA=-DDS_DIAG libds_m32diag.so - built in the container, it travels with the snapshot)
python3 tools/mb_kx.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_mb_kx.txt; cat $out/${tag}_mb_kx.txt
python3 tools/mb_corner_time.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_mb_corner.txt; cat $out/${tag}_mb_corner.txt
python3 tools/mb_solver_spmm.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_mb_solver_spmm.txt; cat $out/${tag}_mb_solver_spmm.txt
if [ -f diffsound_amd/csrc/libds_m32diag.so ]; then
  DS_EXP_LIB=$PWD/diffsound_amd/csrc/libds_m32diag.so python3 tools/m32_diag.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_m32_diag.txt; cat $out/${tag}_m32_diag.txt
fi
python3 tools/mb_gram_mix.py > $out/${tag}_gram_mix.txt 2>&1; cat $out/${tag}_gram_mix.txt
# matrix-pipe counters of the Rayleigh-Ritz kernels, ONE shape per profiled process (the iteration's shapes on the raw basis and the
# explicit route's) -> profiles/gram_mix_mfma_util.json, which bench.py quotes (roofline.rayleigh_ritz.mfma_busy)
for shape in gram_256x160 mix_256x160 gram_256x80 gram_240x80 mix_256x80 mix_240x160; do
  rm -rf /tmp/pmc_g
  timeout -k 10 200 rocprofv3 --kernel-include-regex "gram32_partial|mix_lds" --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_g -o p -- python3 tools/mb_rr_shapes.py $shape > /tmp/pmc_log_g.txt 2>&1 || { tail -5 /tmp/pmc_log_g.txt; exit 1; }
  python3 tools/pmc_summary.py /tmp/pmc_g "gram32_partial_kernel" "mix_lds_kernel" > $out/${tag}_rr_pmc_$shape.json
  echo "$shape counters done"
done
python3 tools/mfma_util.py $tag > $out/${tag}_gram_mix_mfma_util.json; cat $out/${tag}_gram_mix_mfma_util.json
fi
if [ $part = 3 ] || [ $part = all ]; then
# configs[4]: the 1M-tet / 128-mode / fp64 stress with its kernel table
rm -rf /tmp/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o c5 -- python3 bench.py --workload c5 > $out/${tag}_c5_bench.json 2> $out/${tag}_c5_stderr.log
python3 tools/summarize_prof.py /tmp/prof_c5 $out/${tag}_c5_kernel_stats.csv --top 40
fi
echo "all done"
