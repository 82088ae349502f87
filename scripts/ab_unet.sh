#!/bin/bash
# A B A B of two builds of the library on ONE box (box-to-box spread exceeds most effects): scripts/ab_unet.sh <other.so> [time_unet args]
# prints the captured batch-16 forward for the in-tree build (A) and for COMA_HIP_LIB=<other.so> (B), twice each, alternating.
OTHER=$1; shift
ARGS=${*:-16 20 --shared}
for i in 1 2; do
  echo -n "A (in-tree)  "; python scripts/time_unet.py $ARGS 2>&1 | tail -1
  echo -n "B ($OTHER) "; COMA_HIP_LIB=$OTHER python scripts/time_unet.py $ARGS 2>&1 | tail -1
done
