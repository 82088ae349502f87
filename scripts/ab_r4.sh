#!/bin/bash
python -m pytest tests/test_sd_model_gpu.py tests/test_sd_unet_gpu.py -m gpu -q -x 2>&1 | tail -4
for i in 1 2; do for c in 40 20; do echo -n "SD_GN_WINOGRAD_MIN_CG=$c  "; SD_GN_WINOGRAD_MIN_CG=$c python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1; done; done
echo -n "SD_WINOGRAD=16  "; SD_WINOGRAD=16 python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1
