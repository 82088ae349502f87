#!/bin/bash
python -m pytest tests/test_sd_ops_gpu.py -m gpu -q -x 2>&1 | tail -5
python -m pytest tests/test_sd_vae_gpu.py tests/test_sd_model_gpu.py -m gpu -q -rP 2>&1 | grep -E "METRIC|passed|failed|Error" | tail -8
for i in 1 2; do
  python scripts/time_vae.py 8 2>&1 | grep -E "decoder|encoder"
  echo -n "phases off: "; python - <<'PY'
import sys, os, time; sys.path.insert(0, os.getcwd())
import torch
from coma_amd.sd import weights, vae as vae_mod
vae_mod._VaeBase.upsample_phases = False
from coma_amd.sd.vae import HipAutoencoderKL
vae = HipAutoencoderKL(weights.random_state(weights.vae_shapes(), seed=1), batch=8, device="cuda:0")
vae.dec.z.copy_(torch.randn(vae.dec.z.shape, device="cuda:0").half()); vae.dec.z[:, :, 4:] = 0
run = vae.dec.decode_static
run(); run(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize(); print(f"decoder (3x3 over the upsampled tensor) B=8: {(time.perf_counter()-t0)/5*1e3:.2f} ms")
PY
  echo -n "UNet phases on:  "; python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1
  echo -n "UNet phases off: "; SD_UPSAMPLE_PHASES=0 python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1
done
python scripts/time_vae.py 8 --profile 2>&1 | grep -E "phase|decoder"
python -m pytest tests/test_sd_unet_gpu.py -m gpu -q -rP 2>&1 | grep -E "passed|failed|Error" | tail -3
