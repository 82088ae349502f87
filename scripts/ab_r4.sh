#!/bin/bash
for i in 1 2; do
  echo -n "base                "; python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1
  echo -n "LN fold from C=640  "; SD_LN_FOLD=1 SD_LN_FOLD_MIN_C=640 python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1
  echo -n "LN fold from C=1280 "; SD_LN_FOLD=1 SD_LN_FOLD_MIN_C=1280 python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1
done
