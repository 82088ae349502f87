#!/usr/bin/env python3
"""Per-launch timing of the person-segmentation plan (coma_amd/seg): python scripts/time_seg.py [batch] [detections] [--eager-only]

Prints the captured forward time, then every launch of the plan with its HIP-event time (eager, 3 repetitions), algorithmic flops
and TF/s, and the totals per launch family.  Seeded random weights, a smooth noise image (bench.py's pointrend_plugin)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    batch = int(args[0]) if args else 8
    forced = int(args[1]) if len(args) > 1 else 4
    from coma_amd.seg import weights as SW
    from coma_amd.seg.predictor import HipPointRendPredictor
    dev = torch.device("cuda:0")
    state = SW.random_state(seed=0, cls_gain=0.2, delta_gain=0.1, person_bias=3.0)
    eager_only = "--eager-only" in sys.argv          # counter passes: rocprofv3 --pmc segfaults under hipGraph replay on this image
    if eager_only:
        from coma_amd.seg.model import HipPointRend
        plan = HipPointRend(state, batch, 512, 512, dev, score_thresh=0.2, keep_masks=False, use_graph=False, detections_per_image=forced)
    else:
        pred = HipPointRendPredictor(pointrend_thres=0.2, device=dev, state=state, detections_per_image=forced)
        plan = pred.pointrend_seg_model.plan(batch, 512, 512)
    g = torch.Generator().manual_seed(9)
    low = torch.rand(batch, 3, 16, 16, generator=g)
    img = (torch.nn.functional.interpolate(low, size=(512, 512), mode="bicubic").clamp(0, 1) * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().to(dev)
    out = plan(img)
    plan(img)
    torch.cuda.synchronize()
    print("detections per image:", out["count"].tolist(), " proposals:", plan.t["prop_count"].tolist())
    for rep in range(0 if eager_only else 3):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            plan(img)
        e.record()
        torch.cuda.synchronize()
        print(f"captured forward, batch {batch}: {a.elapsed_time(e) / 5:.3f} ms  ({a.elapsed_time(e) / 5 / batch:.3f} ms per image)")
    prof = plan.g.profile(reps=3)
    fam = {}
    for tag, fl, ms in prof:
        key = "seg gemm" if tag.startswith("seg gemm") else tag.split(" p")[0] if tag.startswith("seg rpn select") else tag
        f = fam.setdefault(key, [0, 0.0, 0.0])
        f[0] += 1
        f[1] += ms
        f[2] += fl
        print(f"{ms * 1e3:9.1f} us  {fl / 1e9:9.2f} GF  {fl / ms / 1e9 if ms > 0 else 0:7.1f} TF/s  {tag}")
    tot = sum(ms for _, _, ms in prof)
    print(f"eager sum {tot:.3f} ms over {len(prof)} launches; flops {plan.g.flops / 1e9:.1f} GF ({plan.g.flops / batch / 1e9:.1f} per image)")
    for k, (n, ms, fl) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"  {ms:8.3f} ms  {100 * ms / tot:5.1f} %  {n:4d} launches  {fl / ms / 1e9 if ms > 0 else 0:7.1f} TF/s  {k}")


if __name__ == "__main__":
    main()
