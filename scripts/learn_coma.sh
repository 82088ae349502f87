#!/bin/bash
# Learn ComA for one category on MI355X.  Same flags as the reference's scripts/learn_coma.sh
# (--IoU_threshold_min --inlier_num_threshold_min --dataset_type --supercategory --category --no_skip_done) plus --gpus.
# Only the accelerated stage is run here: filtering / down-sampling are upstream stages of the reference (out of scope,
# SURVEY.md section 2) and must already have written results/coma/{human_postfilterings,asset_downsample}.
# With more than one GPU the samples of every (asset, prompt) are sharded over ranks and the partial ComA states are
# summed with one RCCL all-reduce (src/coma/extract_coma.py).
set -e
gpu_ids=(0)
skip=(--skip_done)
while [[ $# -gt 0 ]]; do
  case $1 in
    --gpus) shift; gpu_ids=(); while [[ $# -gt 0 && $1 != --* ]]; do gpu_ids+=("$1"); shift; done ;;
    --IoU_threshold_min|--inlier_num_threshold_min|--dataset_type) shift 2 ;;   # consumed by the upstream stages
    --supercategory) supercategory="$2"; shift 2 ;;
    --category) category="$2"; shift 2 ;;
    --no_skip_done) skip=(); shift 1 ;;
    *) echo "Unknown option: $1"; exit 1 ;;
  esac
done
n=${#gpu_ids[@]}
export HIP_VISIBLE_DEVICES=$(IFS=,; echo "${gpu_ids[*]}")
launch="python"
if [ "$n" -gt 1 ]; then launch="python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29611"; fi
# the reference passes qual:${category}_object/_human/_occupancy; all three resolve here (constants/coma/qual.py aliases)
for key in "qual:${category}_object" "qual:${category}_human" "qual:${category}_occupancy"; do
  $launch src/coma/extract_coma.py --supercategories "$supercategory" --categories "$category" --hyperparams_key "$key" "${skip[@]}"
done
