#!/bin/bash
# round-4 A/B batch (GPU box): z-batched 256x320 tile for the Winograd plane products inside the captured forward
python -m pytest tests/test_sd_ops_gpu.py -m gpu -q -x -k "winograd or batched" 2>&1 | tail -3
for i in 1 2; do
  for z in 0 1; do echo -n "SD_WINOGRAD=16 SD_GEMM_ZBIG=$z  "; SD_GEMM_ZBIG=$z python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1; done
done
echo -n "SD_WINOGRAD=0  "; SD_WINOGRAD=0 python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1
python scripts/time_unet.py 16 5 --shared --profile 2>&1 | grep -iE "winograd|z=16|groupnorm B=16 hw=(64|256) "
python -m pytest tests/test_sd_unet_gpu.py tests/test_sd_model_gpu.py -m gpu -q -rP 2>&1 | grep -E "passed|failed|Error" | head
