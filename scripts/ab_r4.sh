#!/bin/bash
python -m pytest tests/test_sd_ops_gpu.py -m gpu -q -x -k "winograd" 2>&1 | tail -4
for i in 1 2; do for c in 1 0; do echo -n "SD_GN_TABLE_WINOGRAD=$c  "; SD_GN_TABLE_WINOGRAD=$c python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1; done; done
python -m pytest tests/test_sd_unet_gpu.py tests/test_sd_model_gpu.py -m gpu -q -rP 2>&1 | grep -E "METRIC|passed|failed|Error" | sort | uniq -c | sort -rn | head -8
