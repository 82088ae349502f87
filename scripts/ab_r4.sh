#!/bin/bash
python -m pytest tests/test_sd_ops_gpu.py -m gpu -q -k "winograd" 2>&1 | tail -3
python -m pytest tests/test_sd_unet_gpu.py tests/test_sd_pipeline_gpu.py tests/test_sd_adaptive_gpu.py -m gpu -q --durations=14 -rP 2>&1 | grep -E "METRIC fixed|passed|failed|Error|s call|s setup" | head -30
python bench.py --steps 3 --warmup 1 2>gpurun_out/bench_r4a.err | tail -1 > gpurun_out/bench_r4a.json; head -c 1500 gpurun_out/bench_r4a.json; echo; tail -3 gpurun_out/bench_r4a.err
