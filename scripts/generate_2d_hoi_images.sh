#!/bin/bash
# Same flags as the reference's scripts/generate_2d_hoi_images.sh (--gpus --dataset_type --supercategory --category
# --no_skip_done).  Rendering / mask selection / prompt generation are upstream (Blender, out of scope); this runs the
# accelerated stage on the renders they left under results/generation/.
set -e
args=()
gpus=()
while [[ $# -gt 0 ]]; do
  case $1 in
    --gpus) shift; while [[ $# -gt 0 && $1 != --* ]]; do gpus+=("$1"); shift; done ;;
    --dataset_type) shift 2 ;;
    --supercategory) args+=(--supercategories "$2"); shift 2 ;;
    --category) args+=(--categories "$2"); shift 2 ;;
    --no_skip_done) args+=(--no_skip_done); shift 1 ;;
    *) echo "Unknown option: $1"; exit 1 ;;
  esac
done
if [ ${#gpus[@]} -gt 0 ]; then args+=(--gpus "${gpus[@]}"); fi
bash scripts/generation/inpaint.sh "${args[@]}"
