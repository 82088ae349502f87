"""GPU: bench.py's contract -- one JSON line with the required keys at N=1, and the N>1 control flow (barriers, MAX over
ranks, the ComA all-reduce, rank-0-only printing) exercised with two ranks sharing cuda:0 over gloo
(COMA_BENCH_SHARED_DEVICE=1: RCCL refuses two ranks on one device; the driver's multi-GPU run uses RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}
SMALL = ["--steps", "1", "--warmup", "0", "--ddim-steps", "3", "--contact-steps", "1", "--samples", "4", "--human-res", "512",
         "--no-cpu-baseline"]


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{\"metric\"")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line(hip_lib):
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + SMALL, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert KEYS <= set(line) and line["n_gpus"] == 1 and line["value"] > 0
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert "workload" in line["config"] and line["secondary"]["value"] > 0
    assert line["roofline"]["algorithmic_bytes"] > 0                       # so that traffic / algorithmic is on the line
    occ = line["occupancy"]
    assert occ["value"] > 0 and occ["roofline"]["traffic"] > 0 and 0 < occ["roofline"]["frac"] < occ["roofline"]["structure_b_frac"] < 1
    assert line["adaptive_loop"]["value"] > line["adaptive_loop_b1"]["value"] > 0        # the one-image-per-call shape is a stated number
    pr = line["adaptive_loop_pointrend"]                                    # the device PointRend-architecture plug-in, measured in a child process
    assert 0 < pr["value"] < line["adaptive_loop"]["value"] and pr["plugin"] == "pointrend" and "child" in pr["process"]
    seg = pr["segmentation"]
    assert seg["forward_ms"] > 0 and 0 < seg["roofline"]["frac"] < 1 and seg["roofline"]["peak"] == 157.3
    assert seg["detections_per_image_after_the_last_loop"] == seg["detections_cap"] == 4


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_two_ranks_control_flow(hip_lib, launcher):
    """launcher = "self": a plain `python bench.py --gpus 2` re-executes itself under torch.distributed.run."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["COMA_BENCH_SHARED_DEVICE"] = "1"
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29541", "bench.py", "--gpus", "2"] + SMALL
    else:
        cmd = [sys.executable, "bench.py", "--gpus", "2"] + SMALL
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["secondary"]["n_gpus"] == 2
    # all four sections run at N > 1: the contact job ends with ONE all-reduce, occupancy runs row-sharded with its MAX all-reduce
    assert line["secondary"]["final_allreduce_ms"] > 0 and "ONE RCCL all-reduce" in line["secondary"]["config"]["workload"]
    assert line["occupancy"]["n_gpus"] == 2 and line["occupancy"]["value"] > 0 and "all-reduce(MAX)" in line["occupancy"]["config"]["workload"]
    assert line["adaptive_loop"]["n_gpus"] == 2 and line["adaptive_loop"]["value"] > 0
