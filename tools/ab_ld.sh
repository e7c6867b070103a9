for i in 1 2 3; do
  for ld in 0 1; do
    DS_EXP_LD=$ld timeout -k 10 300 python bench.py --no-cpu-baseline --amortised-cycle 0 --steps 20 > gpurun_out/r04_ab_ld_${ld}_$i.json 2>/dev/null
    python -c "import json;d=json.load(open('gpurun_out/r04_ab_ld_${ld}_$i.json'));print('[rows 1 KiB apart: $ld]',round(d['value'],2),'passes/s')"
  done
done
