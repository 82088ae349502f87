#!/bin/bash
# r04: conv_gemm traffic of eager UNet forwards (FETCH_SIZE / WRITE_SIZE in separate passes); B = 1 shape with and without Winograd
R=${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/pmc_cmd.sh unet_traffic "FETCH_SIZE" "WRITE_SIZE" -- python $R/scripts/time_unet.py 16 3 --eager --shared > gpurun_out/pmc_unet_traffic.log 2>&1
python scripts/unet_traffic.py gpurun_out/pmc_unet_traffic/summary.txt 4 > gpurun_out/unet_gemm_traffic.txt 2>&1
cat gpurun_out/unet_gemm_traffic.txt
for i in 1 2; do for w in 32 0; do echo -n "B=2 SD_WINOGRAD=$w  "; SD_WINOGRAD=$w python scripts/time_unet.py 2 30 --shared 2>&1 | tail -1; done; done
