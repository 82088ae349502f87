#!/bin/bash
# round-4 A/B batch (GPU box): Winograd upsamplers, fused UNet conv_out; then the re-timed parity tests
python -m pytest tests/test_sd_ops_gpu.py -m gpu -q -k "winograd or small_n" 2>&1 | tail -4
for i in 1 2; do
  echo -n "all on          "; python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1
  echo -n "SD_WINOGRAD_UP=0  "; SD_WINOGRAD_UP=0 python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1
  echo -n "SD_FUSE_CONV_OUT=0  "; SD_FUSE_CONV_OUT=0 python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1
done
python scripts/time_unet.py 16 5 --shared --profile 2>&1 | grep -iE "upsampled|small n|eager per-launch"
python -m pytest tests/test_sd_unet_gpu.py tests/test_sd_pipeline_gpu.py tests/test_sd_adaptive_gpu.py -m gpu -q -x --durations=12 -rP 2>&1 | grep -E "METRIC fixed|passed|failed|Error|s call|s setup" | head -30
