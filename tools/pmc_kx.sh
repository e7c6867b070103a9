#!/bin/bash
# PMC passes over the 80-column K X product (tools/mb_kx_only.py); summaries -> gpurun_out/pmc_kx_<tag>.json
# usage: tools/pmc_kx.sh <tag> [env assignments...]
set -e
tag=$1; shift
export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
out=/tmp/pmc_$tag; rm -rf $out; mkdir -p $out gpurun_out
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-include-regex "spmm_" --pmc $set --output-format csv -d $out/p$i -o p -- python tools/mb_kx_only.py 4 > $out/log$i.txt 2>&1 || { tail -5 $out/log$i.txt; exit 1; }
echo "pass $i done $(date +%s)" >> gpurun_out/pmc_progress.txt
done <<SETS
TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum
TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD
SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU
SETS
python tools/pmc_summary.py $out spmm_ > gpurun_out/pmc_kx_$tag.json
