#!/bin/bash
# HBM bytes per launch of the fused Chebyshev-term SpMM: two separate PMC passes (FETCH_SIZE, WRITE_SIZE) over
# tools/mb_cheb_only.py; summary -> gpurun_out/pmc_bytes.json   (guides/MI355X_MICROARCH.md, HBM section)
set -e
export TMPDIR=/tmp
out=/tmp/pmc_bytes; rm -rf $out; mkdir -p $out gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 150 rocprofv3 --kernel-include-regex "spmm_union" --pmc $c --output-format csv -d $out/$c -o p -- python3 tools/mb_cheb_only.py 6 > $out/log_$c.txt 2>&1 || { tail -5 $out/log_$c.txt; exit 1; }
  echo "$c done $(date +%s)" >> gpurun_out/pmc_progress.txt
done
python3 tools/pmc_summary.py $out "spmm_union_kernel<20, 1" "spmm_union_kernel<20, 0" "spmm_union_kernel" > gpurun_out/pmc_bytes.json
cat gpurun_out/pmc_bytes.json
