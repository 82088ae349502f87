"""GPU: BASELINE config 3 -- the full adaptive-mask loop (512 x 512, strength 0.98 -> 49 DDIM steps, 21 mask re-estimations)
against oracle/sd_oracle.AdaptiveLoopRef, the fp32 restatement of utils/adaptive_mask_inpainting.py:988-1076, :1111-1157.

How the restatement is evaluated: it is plain torch fp32 functional code; a 49-step loop at 64 x 64 latents takes ~10 minutes per
image on the host cores (12 s per UNet forward), so here it is evaluated by torch's own fp32 kernels on the device (nothing of
libcoma_hip.so is involved) -- `test_restatement_on_device_equals_restatement_on_cpu` ties that evaluation to the CPU one on a small
case.  Both loops consume the same noise draws (recorded from the HIP run) and the same deterministic mask plug-in.

Checked per re-estimation: the mask glue is BIT-EXACT when the restatement's glue (scipy dilation, logical_and, area test) is fed
the HIP loop's own segmentation; the free-running masks agree (IoU); the x0 / masked-image latents track; and the final latents
are within the loop tolerance.  Image b of a batch-8 run is compared with its own batch-1 run.  "parity unpinned" still applies to
the UNet / VAE arithmetic (diffusers absent), as stated in oracle/sd_oracle.py."""
import numpy as np
import pytest
import torch

from oracle import sd_oracle as so
from tests.adaptive_common import iou, make_inputs, make_plugin, run_hip, run_ref, take

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# tolerances (fp16 UNet / VAE against fp32, 49 steps) = 2 x what scripts/adaptive_check.py measures (profiles/r03_notes.md 3,
# profiles/r04_notes.md 3): final latents 2.0e-3 ... 3.5e-3 relative L2 -> 7e-3, x0 at the re-estimations <= 7e-3 -> 1.4e-2, so that a
# kernel regression of 2-3 x shows; free-running mask IoU >= 0.989 -- the plug-in decides per 16 x 16 block, and ONE block flipping
# on a +-1 uint8 difference of the decoded image moves the IoU of a ~30 000-pixel mask by 0.8 %, so the per-step bar is 0.98 (two
# blocks) and the mean over all steps and images 0.995.
MASK_IOU, MASK_IOU_MEAN, FINAL_REL, X0_REL = 0.98, 0.995, 7e-3, 1.4e-2


def _pipe(B, HW=512):
    from coma_amd.sd.pipeline import AdaptiveMaskInpaintPipeline
    return AdaptiveMaskInpaintPipeline.from_random(batch_size=B, height=HW, width=HW, device=DEV, seed=0)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def _check_against_ref(inp, hip, ref, B):
    assert len(hip["trace"]) == len(ref["trace"]) == 21
    ious = []
    for h, r in zip(hip["trace"], ref["trace"]):
        assert (h["i"], h["t"]) == (r["i"], r["t"])
        k = inp["settings"].dilate_scheduler(h["i"])
        for b in range(B):
            # (1) glue, teacher-forced: HIP segmentation -> restated dilation / AND / area test == HIP mask, bit for bit
            exp = so.adapt_mask_ref(h["seg"][b], inp["default_np"][b], k, h["use_default"], inp["thres"]).astype(np.uint8)
            assert np.array_equal(h["mask"][b], exp), (h["i"], b)
            assert int(h["area"][b]) == (0 if h["use_default"] else int(h["seg"][b].sum()))     # forced default: no area pass
            L = inp["HW"] // 8
            assert np.array_equal(h["mask_lat"][b].reshape(L, L), exp[::8, ::8].astype(np.float32))
            # (2) free-running: same mask as the restatement's own loop
            ious.append(iou(h["mask"][b], r["mask"][b]))
            assert ious[-1] >= MASK_IOU, (h["i"], b, ious[-1])
        assert _rel(h["x0"], r["x0"]) <= X0_REL, (h["i"], _rel(h["x0"], r["x0"]))
    assert float(np.mean(ious)) >= MASK_IOU_MEAN, float(np.mean(ious))
    for b in range(B):
        assert _rel(hip["latents"][b], ref["latents"][b]) <= FINAL_REL, (b, _rel(hip["latents"][b], ref["latents"][b]))
    # the adapted masks really differ from the default mask somewhere (the branch under test is live) and stay inside it
    assert any((h["mask"] != inp["default_np"]).any() for h in hip["trace"])
    assert all((h["mask"] <= inp["default_np"]).all() for h in hip["trace"])


def take_hip(hip, idx):
    """The HIP run's results for images `idx` only (every traced tensor has the batch as its leading axis)."""
    out = dict(hip, latents=hip["latents"][idx])
    out["trace"] = [{k: (v[idx] if hasattr(v, "shape") and getattr(v, "ndim", 0) >= 1 and v.shape[0] == len(hip["latents"]) else v)
                     for k, v in h.items()} for h in hip["trace"]]
    return out


def _check_glue_exact(inp, hip, B):
    """Teacher-forced mask glue of EVERY image and re-estimation: HIP segmentation -> restated dilation / AND / area test == HIP mask."""
    for h in hip["trace"]:
        k = inp["settings"].dilate_scheduler(h["i"])
        for b in range(B):
            exp = so.adapt_mask_ref(h["seg"][b], inp["default_np"][b], k, h["use_default"], inp["thres"]).astype(np.uint8)
            assert np.array_equal(h["mask"][b], exp), (h["i"], b)


@pytest.fixture(scope="module")
def fp32_strict():
    old = torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32
    torch.backends.cudnn.allow_tf32 = torch.backends.cuda.matmul.allow_tf32 = False
    yield
    torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32 = old


def test_restatement_on_device_equals_restatement_on_cpu(hip_lib, fp32_strict):
    """128 x 128 image, the last 2 of the 50 timesteps (1 re-estimation; every CPU step costs ~15 s of the suite's budget): AdaptiveLoopRef
    evaluated on the CPU == evaluated by torch's fp32 device kernels, masks identical; and the HIP loop agrees with both."""
    inp = make_inputs(1, HW=128, seed=11, ratio=0.0)
    hip = run_hip(_pipe(1, 128), inp, make_plugin("block"), strength=0.04)
    cpu = run_ref(inp, hip["noises"], make_plugin("block"), strength=0.04, device="cpu")
    dev = run_ref(inp, hip["noises"], make_plugin("block"), strength=0.04, device=DEV)
    assert len(cpu["trace"]) == len(dev["trace"]) == len(hip["trace"]) == 1
    assert _rel(dev["latents"], cpu["latents"]) <= 1e-4
    for c, d, h in zip(cpu["trace"], dev["trace"], hip["trace"]):
        assert iou(c["mask"][0], d["mask"][0]) >= 0.999 and iou(c["mask"][0], h["mask"][0]) >= MASK_IOU
    assert _rel(hip["latents"], cpu["latents"]) <= 3e-2            # 16 x 16 latents: fewer elements per GroupNorm, 1.5e-2 measured


@pytest.mark.parametrize("B", [1, 8])
def test_adaptive_loop_49_steps_matches_restatement(hip_lib, fp32_strict, B):
    inp = make_inputs(B, seed=5)
    pipe = _pipe(B)
    hip = run_hip(pipe, inp, make_plugin("block"), strength=0.98)
    del pipe
    torch.cuda.empty_cache()
    # every image of the batch goes through the bit-exact glue check; the fp32 restatement (25 s per image and loop on the device) runs
    # for two of the eight images of the B = 8 case -- the images of a batch are independent, each has its own noise stream and mask
    idx = list(range(B)) if B == 1 else [0, 6]
    _check_glue_exact(inp, hip, B)
    sub = take(inp, idx)
    ref = run_ref(sub, [n[idx] for n in hip["noises"]], make_plugin("block"), strength=0.98, device=DEV)
    _check_against_ref(sub, take_hip(hip, idx), ref, len(idx))
    if B == 8:
        # image b of the batch-8 run == its own batch-1 run (same generator seed, same inputs; other tile shapes -> fp16 rounding only)
        p1 = _pipe(1)
        for b in (5,):
            one = run_hip(p1, take(inp, [b]), make_plugin("block"), strength=0.98, seeds=[100 + b])
            assert _rel(one["latents"][0], hip["latents"][b]) <= FINAL_REL
            for h1, h8 in zip(one["trace"], hip["trace"]):
                assert iou(h1["mask"][0], h8["mask"][b]) >= MASK_IOU


def test_fixed_mask_loop_10_steps_batch_8(hip_lib, fp32_strict):
    """BASELINE config 2's shape (batch 8, 512 x 512, fixed mask, CFG 11) for 10 DDIM steps against the restatement."""
    inp = make_inputs(8, seed=7)
    pipe = _pipe(8)
    hip = run_hip(pipe, inp, make_plugin("block"), strength=1.0, use_adaptive_mask=False, steps=10)
    del pipe
    torch.cuda.empty_cache()
    idx = [1, 4, 7]                                  # the restatement runs for three of the eight (independent) images
    ref = run_ref(take(inp, idx), [n[idx] for n in hip["noises"]], make_plugin("block"), strength=1.0, use_adaptive_mask=False, steps=10, device=DEV)
    assert not hip["trace"] and hip["last_mask"] is None
    for j, b in enumerate(idx):
        assert _rel(hip["latents"][b], ref["latents"][j]) <= FINAL_REL, (b, _rel(hip["latents"][b], ref["latents"][j]))
    assert bool(torch.isfinite(hip["latents"]).all()) and len({float(hip["latents"][b].abs().sum()) for b in range(8)}) == 8


def test_batched_mask_adapt_matches_numpy(hip_lib):
    """sd_mask_adapt_batched: per-image area test on the device, dilation, AND, nearest 8x down-sample, masked image -- against NumPy
    for an image below the area threshold, one above it, one forced to the default mask; 16-byte writes only touch channels 0-7."""
    from coma_amd.sd import ops
    B, H, W = 3, 64, 96
    rng = np.random.default_rng(0)
    seg = (rng.random((B, H, W)) > 0.97).astype(np.uint8)
    seg[1] = 0
    seg[1, 10, 10] = 1                                   # area 1 < threshold -> default mask
    dflt = np.zeros((B, H, W), np.uint8)
    dflt[:, 8:56, 16:80] = 1
    img = rng.uniform(-1, 1, size=(B, 3, H, W)).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to(DEV)
    for k, force, thres in ((0, False, 5.0), (3, False, 5.0), (20, False, 5.0), (5, True, 0.0), (2, False, 1e9)):
        mask_full = torch.empty(B, H, W, dtype=torch.uint8, device=DEV)
        mask_lat = torch.empty(B, H // 8 * W // 8, dtype=torch.float16, device=DEV)
        masked = torch.full((B * H * W, 64), 7.0, dtype=torch.float16, device=DEV)
        area = torch.empty(B, dtype=torch.int32, device=DEV)
        scratch = torch.empty(B, H, W, dtype=torch.uint8, device=DEV)
        ops.mask_adapt_batched(d(seg), d(dflt), d(img), mask_full, mask_lat, masked, area, scratch, batch=B, H=H, W=W, dilate_iters=k,
                               force_default=force, area_thres=thres, cpad=64)
        assert area.tolist() == ([0] * B if force else seg.reshape(B, -1).sum(-1).tolist())
        for b in range(B):
            ref = so.adapt_mask_ref(seg[b], dflt[b], k, force, thres / (512 * 512)).astype(np.uint8)
            assert np.array_equal(mask_full[b].cpu().numpy(), ref), (k, force, b)
            assert np.array_equal(mask_lat[b].float().cpu().numpy().reshape(H // 8, W // 8), ref[::8, ::8].astype(np.float32))
            m = masked.float().cpu().numpy().reshape(B, H, W, 64)[b]
            exp = np.where(ref[None] > 0, 0.0, img[b]).transpose(1, 2, 0)
            assert np.array_equal(m[:, :, :3], exp.astype(np.float16).astype(np.float32))
            assert float(np.abs(m[:, :, 3:8]).max()) == 0.0 and (m[:, :, 8:] == 7.0).all()     # write_pad=0 leaves channels >= 8 alone


def test_batched_mask_adapt_matches_g20(hip_lib):
    """sd_mask_adapt_batched at 512 x 512 against G20 (tests/golden/make_golden_inpaint.py: scipy.ndimage.binary_dilation, independent of
    the oracle's own dilation): the four segmentations as one batch of 4, k in {0, 1, 5, 20}, area threshold 512 * 512 * 0.005 decided on
    the device -- full-resolution masks bit-equal, latent masks = their nearest 8 x down-sampling."""
    import os
    from coma_amd.sd import ops
    gi = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inpaint_golden.npz"))
    unpack = lambda a: np.unpackbits(a, axis=-1)
    segs, default, ks, thres = unpack(gi["g20_segs"]), unpack(gi["g20_default"]), [int(k) for k in gi["g20_ks"]], float(gi["g20_thres"])
    adapted = unpack(gi["g20_adapted"])
    B, H, W = segs.shape
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    img = torch.zeros(B, 3, H, W, dtype=torch.float32, device=DEV)
    dflt = d(np.broadcast_to(default, (B, H, W)))
    for j, k in enumerate(ks):
        mask_full = torch.empty(B, H, W, dtype=torch.uint8, device=DEV)
        mask_lat = torch.empty(B, H // 8 * W // 8, dtype=torch.float16, device=DEV)
        masked = torch.zeros(B * H * W, 64, dtype=torch.float16, device=DEV)
        area = torch.empty(B, dtype=torch.int32, device=DEV)
        scratch = torch.empty(B, H, W, dtype=torch.uint8, device=DEV)
        ops.mask_adapt_batched(d(segs), dflt, img, mask_full, mask_lat, masked, area, scratch, batch=B, H=H, W=W, dilate_iters=k,
                               force_default=False, area_thres=512 * 512 * thres, cpad=64)
        assert area.tolist() == segs.reshape(B, -1).sum(-1).tolist()
        for b in range(B):
            assert np.array_equal(mask_full[b].cpu().numpy(), adapted[b, j]), (b, k)
            assert np.array_equal(mask_lat[b].float().cpu().numpy().reshape(H // 8, W // 8), adapted[b, j][::8, ::8].astype(np.float32)), (b, k)
