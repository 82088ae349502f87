#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
python bench.py --steps 3 --warmup 1 2>gpurun_out/bench_r4b.err | tail -1 > gpurun_out/bench_r4b.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r4b.json'))
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'achieved', d['roofline']['achieved'], 'executed', d['roofline']['executed_tflops'])
for k in ('adaptive_loop','adaptive_loop_b1','occupancy','secondary'): print(k, d[k]['value'])
print('occ frac', d['occupancy']['roofline']['frac'], 'fused_ms', d['occupancy']['fused_ms'])
PY
bash scripts/profile_gpu.sh inpaint --steps 2 --warmup 1 > gpurun_out/prof_inpaint.log 2>&1
tail -5 gpurun_out/prof_inpaint.log
for i in 1 2; do python scripts/time_vae.py 8 2>&1 | grep encoder; python - <<'PY'
import sys, os, time; sys.path.insert(0, os.getcwd())
import torch
from coma_amd.sd import weights, vae as vae_mod
vae_mod._VaeBase.packed_conv_in = False
from coma_amd.sd.vae import HipAutoencoderKL
vae = HipAutoencoderKL(weights.random_state(weights.vae_shapes(), seed=1), batch=8, device="cuda:0")
vae.enc.x.copy_(torch.randn(vae.enc.x.shape, device="cuda:0").half()); vae.enc.x[:, :, 3:] = 0
run = vae.enc.encode_static
run(); run(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize(); print(f"encoder (K=576 conv_in) B=8: {(time.perf_counter()-t0)/5*1e3:.2f} ms")
PY
done
