#!/bin/bash
python -m pytest tests/test_sd_vae_gpu.py tests/test_sd_model_gpu.py -m gpu -q -rP 2>&1 | grep -E "METRIC|passed|failed|Error" | tail -8
python scripts/time_vae.py 8 2>&1 | grep -E "decoder|encoder"
python -m pytest tests/test_sd_adaptive_gpu.py tests/test_sd_pipeline_gpu.py tests/test_inpaint_cli_gpu.py -m gpu -q -x 2>&1 | tail -3
