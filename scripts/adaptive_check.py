"""Diagnostic (GPU): HIP adaptive loop vs oracle/sd_oracle.AdaptiveLoopRef (torch fp32 evaluated on the device), per re-estimation
IoU / glue exactness / latent error.  `python scripts/adaptive_check.py [B] [strength] [plugin]`.  Feeds tests/test_sd_adaptive_gpu.py."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coma_amd.sd import weights  # noqa: E402
from coma_amd.sd.pipeline import AdaptiveMaskInpaintPipeline, default_adaptive_mask_settings  # noqa: E402
from oracle import sd_oracle as so  # noqa: E402
from tests.adaptive_common import make_inputs, make_plugin, run_hip, run_ref, iou  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
strength = float(sys.argv[2]) if len(sys.argv) > 2 else 0.98
kind = sys.argv[3] if len(sys.argv) > 3 else "block"
HW = int(sys.argv[4]) if len(sys.argv) > 4 else 512
DEV = "cuda:0"
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
inp = make_inputs(B, HW, seed=5)
t0 = time.time()
pipe = AdaptiveMaskInpaintPipeline.from_random(batch_size=B, height=HW, width=HW, device=DEV, seed=0)
hip = run_hip(pipe, inp, make_plugin(kind), strength=strength)
torch.cuda.synchronize()
t1 = time.time()
ref = run_ref(inp, hip["noises"], make_plugin(kind), strength=strength, device=DEV)
torch.cuda.synchronize()
t2 = time.time()
print(f"B={B} strength={strength} plugin={kind}: HIP {t1 - t0:.1f}s (incl. build), oracle-on-device {t2 - t1:.1f}s, "
      f"{len(hip['trace'])} re-estimations")
for h, r in zip(hip["trace"], ref["trace"]):
    assert h["i"] == r["i"]
    glue = so.adapt_mask_ref  # teacher-forced glue: HIP seg through the restatement
    k = inp["settings"].dilate_scheduler(h["i"])
    ious = [iou(h["mask"][b], r["mask"][b]) for b in range(B)]
    seg_ious = [iou(h["seg"][b], r["seg"][b]) for b in range(B)]
    exact = all(np.array_equal(h["mask"][b], glue(h["seg"][b], inp["default_np"][b], k, h["use_default"], inp["thres"]).astype(np.uint8))
                for b in range(B))
    x0e = float((h["x0"].double() - r["x0"].double()).norm() / r["x0"].double().norm())
    mle = float((h["masked_lat"].double() - r["masked_lat"].double()).norm() / r["masked_lat"].double().norm())
    du8 = np.abs(h["image_u8"].astype(np.int32) - r["image_u8"].astype(np.int32))
    print(f" i={h['i']:2d} t={h['t']:3d} k={k:2d} dflt={int(h['use_default'])} mask IoU min {min(ious):.4f} seg IoU min {min(seg_ious):.4f} "
          f"glue-exact {exact} x0 rel {x0e:.3e} masked_lat rel {mle:.3e} u8 maxdiff {du8.max()} mean {du8.mean():.3f} "
          f"area {h['area'].tolist()[:4]}")
def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


fe = [rel(hip["latents"][b], ref["latents"][b]) for b in range(B)]
print("final latents rel-L2 per image (product vs fp32 flow):", ["%.3e" % e for e in fe])
if os.environ.get("FLOW16", "1") != "0":
    # the reference's own dtype flow (fp16 latents / CFG / scheduler step / VAE sample; networks fp32 in both restatements)
    r16 = run_ref(inp, hip["noises"], make_plugin(kind), strength=strength, device=DEV, dtype_flow="fp16")
    print("final latents rel-L2 per image (product vs fp16 flow):", ["%.3e" % rel(hip["latents"][b], r16["latents"][b]) for b in range(B)])
    print("final latents rel-L2 per image (fp16 flow vs fp32 flow):", ["%.3e" % rel(r16["latents"][b], ref["latents"][b]) for b in range(B)])
    x0p = [rel(h["x0"], r["x0"]) for h, r in zip(hip["trace"], r16["trace"])]
    x0f = [rel(a["x0"], r["x0"]) for a, r in zip(r16["trace"], ref["trace"])]
    mi = [min(iou(h["mask"][b], r["mask"][b]) for b in range(B)) for h, r in zip(hip["trace"], r16["trace"])]
    mf = [min(iou(a["mask"][b], r["mask"][b]) for b in range(B)) for a, r in zip(r16["trace"], ref["trace"])]
    print(f"x0 at the re-estimations: product vs fp16 flow {min(x0p):.2e} ... {max(x0p):.2e}; fp16 flow vs fp32 flow {min(x0f):.2e} ... {max(x0f):.2e}")
    print(f"free-running mask IoU (min over steps / images): product vs fp16 flow {min(mi):.4f}; fp16 flow vs fp32 flow {min(mf):.4f}")
