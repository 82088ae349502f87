#!/bin/bash
# round-4 A/B batch (GPU box): Winograd level rule with the z-batched 256x320 tile and the fused GroupNorm transforms
python -m pytest tests/test_sd_ops_gpu.py -m gpu -q -k "winograd" 2>&1 | tail -5
for i in 1 2; do
  for w in 16 32 0; do echo -n "SD_WINOGRAD=$w  "; SD_WINOGRAD=$w python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1; done
done
python scripts/time_unet.py 16 5 --shared --profile 2>&1 | grep -iE "winograd|z=16"
