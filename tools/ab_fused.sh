timeout -k 10 300 python -m pytest tests/test_hip_kernels.py tests/test_modal_gpu.py -x -q -m gpu -k "fused_residual or fresh_products or native_iteration" 2>&1 | tail -3
for i in 1 2 3; do
  for f in 0 1; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --amortised-cycle 0 --steps 20 --fused-residual $f > gpurun_out/r04_ab_fused_${f}_$i.json 2>/dev/null
    python -c "import json;d=json.load(open('gpurun_out/r04_ab_fused_${f}_$i.json'));print('[fused residual $f]',round(d['value'],2),'passes/s  fine its',d['ranks'][0]['fine_iterations'],'loss',d['loss_sum_last_step'])"
  done
done
