#!/usr/bin/env python3
"""Where the fused occupancy pass spends its time (config-5 share H=1310, R=128): HIP-event time of the three kernels for
   (a) S = 8: store sweep only; (b) S = 2000, every sample outside the grid along x: record scan only; (c) S = 2000 uniform in the
   grid (the bench workload); (d) S = 2000 clustered (each vertex around its own spot: the test suite's config-5 case)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from utils.coma_occupancy import ComA_Occupancy

dev = "cuda:0"
H, R = 1310, 128
g = torch.Generator(device=dev).manual_seed(7)


def run(q, tag):
    occ = ComA_Occupancy(scale_tolerance=3.0, human_res=H, obj_res=1, normal_res=0, spatial_res=R, device=dev)
    best = 1e9
    for rep in range(4):
        occ.reset()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        occ.accumulate_device(q)
        occ.return_aggregated_spatial_grids()
        b.record()
        torch.cuda.synchronize()
        if rep:
            best = min(best, a.elapsed_time(b))
    print(f"{tag:60s} {best:7.3f} ms", flush=True)
    del occ
    torch.cuda.empty_cache()


S = 2000
uni = (torch.rand([S, H, 3], generator=g, device=dev) * 2.6 - 1.3).contiguous()
run(uni[:8].contiguous(), "(a) S=8: store sweep only")
out = uni.clone()
out[..., 0] = 5.0
run(out, "(b) S=2000, all outside along x: record scan only")
run(uni, "(c) S=2000 uniform (bench workload)")
centre = torch.rand([1, H, 3], generator=g, device=dev) * 1.8 - 0.9
run((centre + 0.08 * torch.randn([S, H, 3], generator=g, device=dev)).contiguous(), "(d) S=2000 clustered per vertex")
