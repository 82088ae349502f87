#!/bin/bash
# round-4 A/B batch (GPU box): small-n conv timing, UNet Winograd levels inside the captured forward
python -m pytest tests/test_sd_ops_gpu.py -m gpu -q -x -k "small_n" 2>&1 | tail -3
python scripts/time_vae.py 8 --profile 2>&1 | grep -E "decoder|small|table"
python scripts/time_vae.py 8 --gemm-conv-out 2>&1 | grep decoder
python scripts/time_vae.py 8 2>&1 | grep decoder
for i in 1 2; do
  for w in 0 16 32; do echo -n "SD_WINOGRAD=$w  "; SD_WINOGRAD=$w python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1; done
done
SD_WINOGRAD=16 python scripts/time_unet.py 16 5 --shared --profile 2>&1 | grep -iE "winograd|z=16"
SD_WINOGRAD=16 python -m pytest tests/test_sd_unet_gpu.py -m gpu -q -rP 2>&1 | grep -E "METRIC|passed|failed" | sort | uniq -c | sort -rn | head -40
