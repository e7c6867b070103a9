#!/bin/bash
# Knock-out builds of the fp32 matrix-core product kernel (spmm_mfma32.inc; make BUILD=/tmp/build_m32koN LIB=libds_m32koN.so
# EXTRA=-DDS_M32_KO=N): which part of K X is its time?  bits: 1 no MFMAs, 2 no panel gathers, 4 no B-fragment reads,
# 8 no A-fragment chain
out=gpurun_out/r04_mfma32_knockout.txt
: > $out
echo "== production" >> $out
python3 tools/mb_kx.py >> $out 2>&1
for ko in "$@"; do
  echo "== DS_M32_KO=$ko" >> $out
  DS_EXP_LIB=$PWD/diffsound_amd/csrc/libds_m32ko$ko.so python3 tools/mb_kx.py 2>&1 | grep "matrix cores" >> $out
done
grep -v amdgpu.ids $out
