#!/bin/bash
# The UNet at the reference's own call shape (one image per call = UNet batch 2): where the 7 ms go and what each existing switch is worth.
#   gpurun -- 'bash scripts/unet_batch2_switches.sh > gpurun_out/ab_batch2.log 2>&1'
cd ${GRAFT_REPO_ROOT:-.}
T="python scripts/time_unet.py 2 50"
for rep in 1 2; do
  echo "== default";                 $T | tail -1
  echo "== --shared (CFG prefix)";   $T --shared | tail -1
  echo "== --no-xfront";             $T --no-xfront | tail -1
  echo "== --no-xchain";             $T --no-xchain | tail -1
  echo "== --no-xtail";              $T --no-xtail | tail -1
  echo "== --no-qkv";                $T --no-qkv | tail -1
  echo "== winograd from batch 2";   $T --wino-min-batch=2 | tail -1
  echo "== GN stats from M=8192";    SD_GN_STATS_MIN_M=8192 $T | tail -1
  echo "== GN stats from M=2048";    SD_GN_STATS_MIN_M=2048 $T | tail -1
  echo "== eager (no graph)";        $T --eager | tail -1
done
echo "== per-class breakdown, batch 2"
python scripts/unet_breakdown.py 2 --shapes
echo "== adaptive loop at one image per call"
python scripts/time_adaptive.py 1
