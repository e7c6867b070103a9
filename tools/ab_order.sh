for i in 1 2; do
  DS_EXP_ORDER=abs timeout -k 10 300 python bench.py --no-cpu-baseline --amortised-cycle 0 > gpurun_out/r04_ab_abs_$i.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/r04_ab_abs_$i.json'));print('abs  ',round(d['value'],2),'passes/s  K W alone',round(d['roofline']['lobpcg_spmm']['avg_launch_ms']*1e3,1),'us  term alone',round(d['roofline']['achieved']),'GB/s')"
  timeout -k 10 300 python bench.py --no-cpu-baseline --amortised-cycle 0 > gpurun_out/r04_ab_new_$i.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/r04_ab_new_$i.json'));print('plane',round(d['value'],2),'passes/s  K W alone',round(d['roofline']['lobpcg_spmm']['avg_launch_ms']*1e3,1),'us  term alone',round(d['roofline']['achieved']),'GB/s')"
done
