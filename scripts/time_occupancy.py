#!/usr/bin/env python3
"""bench.py's occupancy sub-benchmark alone (config 5 per-GPU share): python scripts/time_occupancy.py"""
import json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
for _ in range(2):
    r = bench.bench_occupancy(types.SimpleNamespace(), torch.device("cuda", 0), 1, 0)
print(json.dumps(r, indent=1))
if "--kernels" in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        bench.bench_occupancy(types.SimpleNamespace(), torch.device("cuda", 0), 1, 0)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
