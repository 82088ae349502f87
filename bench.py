#!/usr/bin/env python3
"""bench.py -- throughput of ComA's dense hot path on MI355X (contract: see the round brief).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload contact|occupancy]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  contact   (default): the fused K1-K3 accumulator over S samples of H=10475 SMPL-X vertices x O=180 object
            points x N=250 orientation bins per GPU (BASELINE.json config 4, per-GPU slice), followed -- when
            N_gpus > 1 -- by the RCCL all-reduce(SUM) of the ComA state (2*H*O*N + 3*H*O floats), i.e. one
            complete "learn a ComA from S*N_gpus samples" job per step.  Weak scaling: S per GPU is fixed.
  occupancy : K5 splat of S samples at H=10475, R=128 rows sharded across ranks + K6 reduce + all-reduce(MAX).
Rank 0 prints ONE JSON line.  `value` = vertex-pair contacts/s summed over all GPUs.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: peak FP32 vector
HBM_PEAK_GBS = 8000.0


def flop_per_pair(N):
    """Algorithmic flop per (sample, h, o) pair, SURVEY.md 8d: 14 (distance/proximity) + 2*43 (canonicalise)
    + 13 per bin per grid (5 dot, 2 clip, acos, square, /sigma^2, exp, recip, add)."""
    return 100 + 26 * N


def cpu_baseline(seconds_budget=25.0):
    """The NumPy oracle (a port of the reference's torch-CPU path, same dtype flow) on the host cores, config 1
    shape (H=1000, O=180, N=250).  Bounded sample; NumPy elementwise kernels run on one thread."""
    from oracle import coma_oracle as orc
    from tests.synth import cfg1_samples
    H, O, N = 1000, 180, 250
    samples = cfg1_samples(9, seed=0, H=H, O=O)
    m = orc.ComAOracle(H, O, N, 0.07, 0.03, sigma=0.25, eps=1e-10)
    m.aggregate_sample(**samples[0])           # warm-up
    done, t0 = 0, time.perf_counter()
    for s in samples[1:]:
        m.aggregate_sample(**s)
        done += 1
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    return {"value": done * H * O / dt, "unit": "vertex-pair contacts/s", "cores": 1, "kind": "port",
            "sample": f"{done} samples of config 1 (H=1000,O=180,N=250) through oracle/coma_oracle.py (NumPy, f64 "
                      f"intermediates as in the reference), {dt:.1f} s; host has {os.cpu_count()} logical cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="contact", choices=["contact"])
    ap.add_argument("--samples", type=int, default=64, help="samples per GPU per step")
    ap.add_argument("--human-res", type=int, default=10475)
    ap.add_argument("--obj-res", type=int, default=180)
    ap.add_argument("--normal-res", type=int, default=250)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from utils.coma import ComA
    H, O, N, S = args.human_res, args.obj_res, args.normal_res, args.samples
    coma = ComA(H, O, N, 0, proximity_settings=dict(spatial_grid_size=0.07, spatial_grid_thres=0.03),
                normal_gaussian_sigma=0.25, eps=1e-10, device=dev)

    # synthetic inputs of config 4's shape, generated on the device (resident in HBM before the timed region)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    lo = torch.tensor([-0.3, -0.15, -0.85], device=dev)
    hi = torch.tensor([0.3, 0.15, 0.85], device=dev)
    hv = lo + (hi - lo) * torch.rand([S, H, 3], generator=g, device=dev)
    hn = torch.nn.functional.normalize(torch.randn([S, H, 3], generator=g, device=dev), dim=-1)
    g0 = torch.Generator(device=dev).manual_seed(99)           # the object is the same on every rank
    on = torch.nn.functional.normalize(torch.randn([O, 3], generator=g0, device=dev), dim=-1)
    ov = on * 0.2 + torch.tensor([0.0, -0.15, 0.3], device=dev)

    def step():
        coma.accumulate_device(hv, hn, ov, on)
        if world > 1:
            coma.all_reduce()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        coma.accumulate_device(hv, hn, ov, on)      # same stream as the events (torch current stream)
        ev[i][1].record()
        if world > 1:
            coma.all_reduce()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        pairs_per_step = world * S * H * O
        value = pairs_per_step * args.steps / dt
        flops = S * H * O * flop_per_pair(N)
        achieved = flops / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": "vertex-pair contacts/sec (ComA K1-K3 accumulation; BASELINE metric part 2)",
            "value": value, "unit": "vertex-pair contacts/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ComA contact+orientation accumulation, H={H} SMPL-X verts x O={O} object points x "
                                   f"N={N} bins, {S} samples per GPU per step (BASELINE.json config 4 per-GPU slice)"
                                   + (", + RCCL all-reduce(SUM) of the ComA state every step" if world > 1 else ""),
                       "parallelism": f"samples sharded over {world} GPU(s)"},
            "roofline": {"bound": "fp32_valu", "kernel": "contact_accumulate_kernel", "achieved": achieved,
                         "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_PEAK_TFLOPS,
                         "kernel_ms": kern_ms, "flop_per_pair": flop_per_pair(N),
                         "algorithmic_hbm_bytes": 24 * (S * H + O) + (16 * N + 24) * H * O,
                         "traffic": None},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
