for i in 1 2; do
  for cap in 116 140; do
    DS_EXP_UNION_CAP=$cap timeout -k 10 200 python tools/mb_kx.py 26 2>&1 | grep "VALU" | sed "s/^/[cap $cap] /"
    DS_EXP_UNION_CAP=$cap timeout -k 10 300 python bench.py --no-cpu-baseline --amortised-cycle 0 > gpurun_out/r04_ab_cap_${cap}_$i.json 2>/dev/null
    python -c "import json;d=json.load(open('gpurun_out/r04_ab_cap_${cap}_$i.json'));r=d['roofline']['lobpcg_spmm'];print('[cap $cap]',round(d['value'],2),'passes/s  K W alone',round(r['avg_launch_ms']*1e3,1),'us =',round(r['frac_of_stream'],3),'of STREAM')"
  done
done
