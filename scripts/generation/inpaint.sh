#!/bin/bash
# Per-GPU fan-out of the inpainting stage (the reference's scripts/generation/inpaint.sh): one process per GPU, each taking
# slice --parallel_idx of --parallel_num of the sorted work list; processes share nothing but the file system.
# The accepted flags are the reference's explicit set (:72-196) -- anything else is an error, not forwarded -- plus the two
# asset-provisioning additions of src/generation/inpaint.py and its --batch_size (images per pipeline call, default 8).  Unset flags fall back to that module's constants (argparse
# defaults), which is what the reference's `python -c` preamble reads.
set -e
gpu_ids=(0 1 2 3 4 5 6 7)
value_flags=" num_img_per_combination prompts_dir asset_render_dir asset_mask_dir asset_seg_dir save_dir ldm_model_key adaptive_mask_model_type default_cfg_scale default_strength default_ddim_steps default_pointrend_threshold default_enforce_full_mask_ratio default_human_detection_thres negative_prompt seed weights_dir mask_model batch_size "
list_flags=" supercategories categories "
bool_flags=" enable_sam_multitask_output enable_safety_checker use_visualizer verbose "
args=()
skip_done=true
while [[ $# -gt 0 ]]; do
  name=${1#--}
  if [[ $1 == --gpus ]]; then
    shift; gpu_ids=()
    while [[ $# -gt 0 && $1 != --* ]]; do gpu_ids+=("$1"); shift; done
  elif [[ $1 == --no_skip_done ]]; then
    skip_done=false; shift
  elif [[ $1 == --* && $list_flags == *" $name "* ]]; then
    args+=("$1"); shift
    while [[ $# -gt 0 && $1 != --* ]]; do args+=("$1"); shift; done
  elif [[ $1 == --* && $value_flags == *" $name "* ]]; then
    [[ $# -ge 2 ]] || { echo "inpaint.sh: $1 needs a value" >&2; exit 2; }
    args+=("$1" "$2"); shift 2
  elif [[ $1 == --* && $bool_flags == *" $name "* ]]; then
    args+=("$1"); shift
  else
    echo "inpaint.sh: unknown argument '$1'" >&2; exit 2
  fi
done
[[ $skip_done == true ]] && args+=(--skip_done)
n=${#gpu_ids[@]}
i=0
for g in "${gpu_ids[@]}"; do
  HIP_VISIBLE_DEVICES=$g python src/generation/inpaint.py "${args[@]}" --parallel_idx $i --parallel_num $n &
  i=$((i + 1))
done
wait
