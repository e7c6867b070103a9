#!/bin/bash
# Knock-out builds of the neighbour-union kernel (make BUILD=build_koN LIB=libds_koN.so EXTRA=-DDS_KO=N): which part of the
# K X product is its time?  bits: 1 no FMAs, 2 no coefficient reads from LDS, 4 no panel gathers, 8 no block-value loads
out=gpurun_out/r03_union_knockout.txt
: > $out
echo "== production" >> $out
python3 tools/mb_kx.py >> $out 2>&1
for ko in 1 2 4 8 3 5 7 15; do
  echo "== DS_KO=$ko" >> $out
  DS_EXP_LIB=$PWD/diffsound_amd/csrc/libds_ko$ko.so python3 tools/mb_kx.py >> $out 2>&1
done
grep -v amdgpu.ids $out
