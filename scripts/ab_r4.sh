#!/bin/bash
# r04 profiles: rocprofv3 kernel trace + PMC passes of the bench command, and the conv_gemm traffic of eager UNet forwards
bash scripts/profile_gpu.sh inpaint --steps 2 --warmup 1 > gpurun_out/prof_inpaint.log 2>&1
tail -40 gpurun_out/prof_inpaint.log
bash scripts/pmc_cmd.sh unet_traffic "FETCH_SIZE" "WRITE_SIZE" -- python scripts/time_unet.py 16 3 --eager --shared > gpurun_out/pmc_unet_traffic.log 2>&1
python scripts/unet_traffic.py gpurun_out/pmc_unet_traffic/summary.txt 4 > gpurun_out/unet_gemm_traffic.txt 2>&1
cat gpurun_out/unet_gemm_traffic.txt
