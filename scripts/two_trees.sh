#!/bin/bash
# A/B of two CHECKOUTS of this repo on ONE box (python + library of each tree; used when the ABI differs between the two, so that
# COMA_HIP_LIB cannot swap the library alone):   scripts/two_trees.sh <other-tree> [rounds] [unet|bench|both]
# per round: the captured UNet forwards (scripts/time_unet.py, batch 16 and 2) and / or bench.py --steps 5 --no-secondary for both trees;
# the order of the two trees flips every round (A B, B A, ...) so that "first after idle" and thermal state do not favour one.
OTHER=$1; ROUNDS=${2:-2}; WHAT=${3:-both}
HERE="$(cd "$(dirname "$0")/.." && pwd)"
for i in $(seq $ROUNDS); do
  if [ $((i % 2)) = 1 ]; then ORDER="$HERE $OTHER"; else ORDER="$OTHER $HERE"; fi
  for T in $ORDER; do
    echo "== round $i tree $T"
    if [ $WHAT != bench ]; then
      (cd $T && python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1)
      (cd $T && python scripts/time_unet.py 2 50 2>&1 | tail -1)
    fi
    if [ $WHAT != unet ]; then
      (cd $T && python bench.py --steps 5 --warmup 1 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['value'],3), round(d['ms_per_step'],1), 'gemm avg launch ms', round(d['roofline']['avg_launch_ms']*1e3,2), 'us; eager sum', round(d['roofline']['unet_forward_ms_eager_sum'],2))")
    fi
  done
done
