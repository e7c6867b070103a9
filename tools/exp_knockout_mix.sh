#!/bin/bash
# Knock-out builds of mix_lds_kernel (make BUILD=build_komN LIB=libds_komN.so EXTRA=-DDS_KOM=N):
# bits: 1 no loads of A after the first two k-steps, 2 no LDS reads of the coefficients after them, 4 no MFMAs
out=gpurun_out/r03_mix_knockout.txt
: > $out
echo "== production" >> $out
ONLY=1,1,0 MIX_ONLY=1 python3 tools/mb_gram_mix.py >> $out 2>&1
for ko in 1 2 4 3; do
  echo "== DS_KOM=$ko" >> $out
  DS_EXP_LIB=$PWD/diffsound_amd/csrc/libds_kom$ko.so MIX_ONLY=1 ONLY=1,1,0 python3 tools/mb_gram_mix.py >> $out 2>&1
done
grep -v "amdgpu.ids\|gram" $out
