for i in 1 2 3; do
  for kf in 0 1; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --amortised-cycle 0 --steps 20 --kx-fresh $kf > gpurun_out/r04_ab_kxf_${kf}_$i.json 2>/dev/null
    python -c "import json;d=json.load(open('gpurun_out/r04_ab_kxf_${kf}_$i.json'));print('[kx_fresh $kf] 8 lanes',round(d['value'],2),'passes/s')"
  done
done
for kf in 0 1; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --amortised-cycle 0 --lanes 1 --hyp-per-gpu 1 --steps 20 --kx-fresh $kf > gpurun_out/r04_ab_kxf1_${kf}.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/r04_ab_kxf1_${kf}.json'));print('[kx_fresh $kf] 1 lane ',round(d['value'],2),'passes/s')"
done
