#!/bin/bash
# sweep of the two-level preconditioner parameters on the benchmark mesh (one line per configuration)
for cfg in "3 10 24 200" "2 10 24 200" "4 10 24 200" "3 6 24 200" "3 20 24 200" "3 10 12 100" "3 10 16 100" "3 10 32 400" "3 10 48 800" "2 6 16 100" "4 20 24 200" "3 10 24 400" "3 10 16 200"; do
  set -- $cfg
  echo -n "SD=$1 SR=$2 CD=$3 CR=$4: "
  SD=$1 SR=$2 CD=$3 CR=$4 python tools/time_solver.py 2>&1 | grep "^total"
done
