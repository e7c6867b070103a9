#!/bin/bash
# Knock-out builds of gram32_partial_kernel (make BUILD=build_kogN LIB=libds_kogN.so EXTRA=-DDS_KOG=N):
# bits: 1 no fp64 folds inside the loop, 2 no operand loads inside the loop, 4 no MFMAs
out=gpurun_out/r03_gram_knockout.txt
: > $out
echo "== production" >> $out
ONLY=240,80,0 python3 tools/mb_gram_mix.py >> $out 2>&1
for ko in 8 24 12; do
  echo "== DS_KOG=$ko" >> $out
  DS_EXP_LIB=$PWD/diffsound_amd/csrc/libds_kog$ko.so ONLY=240,80,0 python3 tools/mb_gram_mix.py >> $out 2>&1
done
grep -v amdgpu.ids $out
