#!/bin/bash
# A B A B of two CHECKOUTS of this repo on ONE box (python + library of each tree; used when the ABI differs between the two, so that
# COMA_HIP_LIB cannot swap the library alone):   scripts/ab_trees.sh <other-tree> [rounds]
# per round and tree: the captured batch-16 UNet forward (scripts/time_unet.py 16 20 --shared) and bench.py --steps 5 --no-secondary.
OTHER=$1; ROUNDS=${2:-2}
HERE="$(cd "$(dirname "$0")/.." && pwd)"
for i in $(seq $ROUNDS); do
  for T in "$HERE" "$OTHER"; do
    echo "== round $i tree $T"
    (cd $T && python scripts/time_unet.py 16 20 --shared 2>&1 | tail -1)
    (cd $T && python scripts/time_unet.py 2 50 2>&1 | tail -1)
    (cd $T && python bench.py --steps 5 --warmup 1 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))")
  done
done
