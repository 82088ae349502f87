#!/bin/bash
timeout 900 python -m pytest tests/test_sd_ops_gpu.py -x -q -k "three_input" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_sd_vae_gpu.py tests/test_sd_model_gpu.py -x -q 2>&1 | tail -5
for i in 1 2; do
  echo -n "A direct "; python scripts/time_vae.py 8 --profile --all 2>&1 | grep "encoder\|c3\|im2col\|conv_in\|groupnorm(table) B=8 hw=262144 C=128" | tail -4 | tr '\n' ' '; echo
  echo -n "B packed "; python scripts/time_vae.py 8 --profile --all --packed-conv-in 2>&1 | grep "encoder\|c3\|im2col\|conv_in\|groupnorm(table) B=8 hw=262144 C=128" | tail -5 | tr '\n' ' '; echo
done
