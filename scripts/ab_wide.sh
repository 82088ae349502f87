#!/bin/bash
# A B A B of the 4-wave / 8-wave wide-head attention inside the VAE graphs (one box)
timeout 900 python -m pytest tests/test_sd_ops_gpu.py -x -q -k "wide" 2>&1 | tail -5
for i in 1 2; do
  for nw in 4 8; do echo -n "NW=$nw B=8 "; SD_WIDE_NW=$nw python scripts/time_vae.py 8 --profile --all 2>&1 | grep "decoder\|encoder\|attention" | tr '\n' ' '; echo; done
done
for nw in 4 8; do echo -n "NW=$nw B=1 "; SD_WIDE_NW=$nw python scripts/time_vae.py 1 --profile --all 2>&1 | grep "decoder\|encoder\|attention" | tr '\n' ' '; echo; done
for nw in 4 8; do echo -n "NW=$nw B=2 "; SD_WIDE_NW=$nw python scripts/time_vae.py 2 --profile --all 2>&1 | grep "decoder\|encoder\|attention" | tr '\n' ' '; echo; done
for nw in 4 8; do echo -n "NW=$nw B=4 "; SD_WIDE_NW=$nw python scripts/time_vae.py 4 --profile --all 2>&1 | grep "decoder\|encoder\|attention" | tr '\n' ' '; echo; done
