#!/bin/bash
# A/B of library builds on one box: tools/exp_ab_kx.sh libA.so libB.so ...  (paths under diffsound_amd/csrc; "-" = the product build)
for l in "$@"; do
  echo "== $l"
  if [ "$l" = "-" ]; then python3 tools/mb_kx.py 2>&1 | grep -v amdgpu.ids
  else DS_EXP_LIB=$PWD/diffsound_amd/csrc/$l python3 tools/mb_kx.py 2>&1 | grep -v amdgpu.ids; fi
done
