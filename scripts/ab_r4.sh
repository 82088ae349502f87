#!/bin/bash
python -m pytest tests -m gpu -q --durations=6 > gpurun_out/full_gpu_tests4.log 2>&1; tail -12 gpurun_out/full_gpu_tests4.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 3 --warmup 1 2>gpurun_out/bench_r4c.err | tail -1 > gpurun_out/bench_r4c.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r4c.json'))
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'achieved', d['roofline']['achieved'], 'executed', d['roofline']['executed_tflops'], 'eager_sum', d['roofline']['unet_forward_ms_eager_sum'])
for k in ('adaptive_loop','adaptive_loop_b1','occupancy','secondary'): print(k, d[k]['value'])
PY
bash scripts/profile_gpu.sh inpaint --steps 2 --warmup 1 > gpurun_out/prof_inpaint.log 2>&1
R=${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/pmc_cmd.sh unet_traffic "FETCH_SIZE" "WRITE_SIZE" -- python $R/scripts/time_unet.py 16 3 --eager --shared > gpurun_out/pmc_unet_traffic.log 2>&1
python scripts/unet_traffic.py gpurun_out/pmc_unet_traffic/summary.txt 4 > gpurun_out/unet_gemm_traffic.txt 2>&1
head -3 gpurun_out/unet_gemm_traffic.txt
