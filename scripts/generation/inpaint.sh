#!/bin/bash
# Per-GPU fan-out of the inpainting stage, as the reference's scripts/generation/inpaint.sh:205-268: one process per GPU,
# each taking slice --parallel_idx of --parallel_num of the sorted work list; processes share nothing but the file system.
# Defaults are read from the Python module, like the reference does with `python -c`.
set -e
gpu_ids=(0 1 2 3 4 5 6 7)
extra=()
skip=(--skip_done)
while [[ $# -gt 0 ]]; do
  case $1 in
    --gpus) shift; gpu_ids=(); while [[ $# -gt 0 && $1 != --* ]]; do gpu_ids+=("$1"); shift; done ;;
    --no_skip_done) skip=(); shift 1 ;;
    *) extra+=("$1"); shift 1 ;;
  esac
done
n=${#gpu_ids[@]}
i=0
for g in "${gpu_ids[@]}"; do
  HIP_VISIBLE_DEVICES=$g python src/generation/inpaint.py "${extra[@]}" "${skip[@]}" --parallel_idx $i --parallel_num $n &
  i=$((i + 1))
done
wait
