#!/bin/bash
python -m pytest tests/test_sd_adaptive_gpu.py -m gpu -q --durations=5 2>&1 | tail -9
for b in 8 4 2; do python scripts/time_vae.py $b 2>&1 | grep -E "decoder|encoder"; done
