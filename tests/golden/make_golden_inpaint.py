#!/usr/bin/env python3
"""Golden vectors G13-G16 for the pure-Python glue of the inpainting path, produced by the REAL reference
(utils/adaptive_mask_inpainting.py imported on CPU with every third-party module auto-stubbed).
Run from anywhere in the build container:  python tests/golden/make_golden_inpaint.py
The UNet / VAE / DDIM arithmetic itself lives in diffusers (absent) -> not pinned here ("parity unpinned")."""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import PIL.Image
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, n):
        return _Dummy()


class _Meta(type):
    def __getattr__(cls, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Dummy()


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Meta(name, (_Dummy,), {})


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    PREFIXES = ("diffusers", "detectron2", "segment_anything", "supervision", "cv2", "open3d", "trimesh", "easydict",
                "transformers", "packaging")

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.PREFIXES:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        pass


def g20_dilation():
    """G20: the mask-dilation branch of adapt_mask (utils/adaptive_mask_inpainting.py:1123-1157) -- `cv2.dilate(mask, ones((3,3)),
    iterations=k)` AND default mask, or the default mask when `mask.sum() < 512 * 512 * thres`.  cv2 is absent from this image, so the
    expected masks come from an implementation INDEPENDENT of the build's own restatement (oracle/sd_oracle.py uses grey_dilation):
    scipy.ndimage.binary_dilation(structure=ones((3,3)), iterations=k, border_value=0), cross-checked here against a shifted-OR in
    plain NumPy.  Still not cv2: what cv2.dilate does with a 3x3 all-ones kernel, default anchor and default border (the border value
    never wins a max) is restated, not run.  Masks are stored bit-packed."""
    from scipy.ndimage import binary_dilation

    def shifted_or(m, k):
        m = m.astype(bool)
        for _ in range(k):
            p = np.pad(m, 1)
            acc = np.zeros_like(m)
            for dy in range(3):
                for dx in range(3):
                    acc |= p[dy:dy + m.shape[0], dx:dx + m.shape[1]]
            m = acc
        return m

    rng = np.random.default_rng(20)
    H = W = 512
    yy, xx = np.mgrid[0:H, 0:W]
    segs = []
    s = np.zeros((H, W), bool)                                   # 0: blobs touching all four borders and a corner + an interior ellipse
    s[0:3, 100:140] = s[H - 2:H, 300:360] = s[200:260, 0:2] = s[50:90, W - 1:W] = s[H - 1, W - 1] = s[0, 0] = True
    s |= ((yy - 256) / 90.0) ** 2 + ((xx - 250) / 40.0) ** 2 <= 1.0
    segs.append(s)
    segs.append(rng.random((H, W)) > 0.995)                      # 1: salt noise (every pixel grows its own square)
    s = np.zeros((H, W), bool)                                   # 2: a small blob: area 600 < 512 * 512 * 0.005 -> the default-mask branch
    s[300:320, 100:130] = True
    segs.append(s)
    s = (((yy - 200) / 120.0) ** 2 + ((xx - 300) / 70.0) ** 2 <= 1.0) & (rng.random((H, W)) > 0.3)      # 3: a ragged person-sized blob
    segs.append(s)
    segs = np.stack(segs)
    default = np.zeros((H, W), bool)                             # the candidate-box default mask of the harness, off-centre, touching one border
    default[96:512, 140:420] = True
    ks, thres = [0, 1, 5, 20], 0.005
    out = {"g20_segs": np.packbits(segs, axis=-1), "g20_default": np.packbits(default, axis=-1), "g20_ks": np.array(ks), "g20_thres": np.array(thres)}
    exp, dil = [], []
    for s in segs:
        for k in ks:
            d = binary_dilation(s, structure=np.ones((3, 3), bool), iterations=k, border_value=0) if k > 0 else s.copy()
            assert np.array_equal(d, shifted_or(s, k)), "scipy.ndimage.binary_dilation and the shifted-OR disagree"
            dil.append(d)
            exp.append(default.copy() if s.sum() < 512 * 512 * thres else (d & default))
    out["g20_dilated"] = np.packbits(np.stack(dil).reshape(len(segs), len(ks), H, W), axis=-1)
    out["g20_adapted"] = np.packbits(np.stack(exp).reshape(len(segs), len(ks), H, W), axis=-1)
    return out


def main():
    sys.meta_path.insert(0, _Finder())
    root = os.path.dirname(os.path.dirname(HERE))
    sys.path = [p for p in sys.path if os.path.abspath(p or ".") != root]
    sys.path.insert(0, REF)
    os.chdir(REF)                                 # constants/segmentation.py opens a relative json
    import utils.adaptive_mask_inpainting as r
    assert r.__file__.startswith(REF)
    out = {}
    # G13 / G14: the schedules of src/generation/inpaint.py:112-132 (values typed here, classes from the reference)
    n, step = 50, 5
    sched = [20] * step + [10] * step + [5] * step + [4] * step + [3] * step + [2] * step + [1] * step + [0] * (n - 7 * step)
    d = r.MaskDilateScheduler(max_dilate_num=20, num_inference_steps=n, schedule=sched)
    out["g13_dilate"] = np.array([d(i) for i in range(n)])
    d2 = r.MaskDilateScheduler(max_dilate_num=15, num_inference_steps=n)
    out["g13_dilate_default"] = np.array([d2(i) for i in range(n)])
    p = r.ProvokeScheduler(num_inference_steps=n, schedule=list(range(2, 11, 2)) + list(range(12, 41, 2)) + [45], is_zero_indexing=False)
    out["g14_provoke"] = np.array([i for i in range(n) if p(i)])
    pz = r.ProvokeScheduler(num_inference_steps=n, schedule=[0, 3, 49], is_zero_indexing=True)
    out["g14_provoke_zero"] = np.array([i for i in range(n) if pz(i)])
    # G15: boxes
    rng = np.random.default_rng(15)
    segs = []
    for k in range(3):
        s = np.zeros((20, 30), np.uint8)
        y0, x0 = rng.integers(0, 10), rng.integers(0, 15)
        s[y0:y0 + rng.integers(1, 9), x0:x0 + rng.integers(1, 14)] = 1
        segs.append(s)
    out["g15_segs"] = np.stack(segs)
    boxes = [r.seg2bbox(s) for s in segs]
    out["g15_boxes"] = np.stack(boxes)
    out["g15_merged"] = r.merge_bbox(boxes)
    # G16: prepare_mask_and_masked_image on PIL, numpy and tensor inputs
    img = rng.integers(0, 256, size=(24, 32, 3)).astype(np.uint8)
    m = (rng.random((24, 32)) > 0.6)
    out["g16_img_u8"], out["g16_mask_bool"] = img, m
    mk, ms, im = r.prepare_mask_and_masked_image(PIL.Image.fromarray(img), PIL.Image.fromarray((m * 255).astype(np.uint8)), 24, 32,
                                                 return_image=True)
    out["g16_pil_mask"], out["g16_pil_masked"], out["g16_pil_image"] = mk.numpy(), ms.numpy(), im.numpy()
    mk, ms = r.prepare_mask_and_masked_image(img, m.astype(np.float32) * 0.7 + 0.1, 24, 32)     # soft mask -> binarised at 0.5
    out["g16_np_mask"], out["g16_np_masked"] = mk.numpy(), ms.numpy()
    ti = torch.tensor(img.transpose(2, 0, 1)[None].astype(np.float32) / 127.5 - 1.0)
    tm = torch.tensor(m.astype(np.float32))[None, None]
    mk, ms = r.prepare_mask_and_masked_image(ti.clone(), tm.clone(), 24, 32)
    out["g16_pt_mask"], out["g16_pt_masked"] = mk.numpy(), ms.numpy()
    errs = []
    for bad in (lambda: r.prepare_mask_and_masked_image(ti * 2, tm, 24, 32), lambda: r.prepare_mask_and_masked_image(ti, tm * 2, 24, 32),
                lambda: r.prepare_mask_and_masked_image(ti, m, 24, 32), lambda: r.prepare_mask_and_masked_image(None, tm, 24, 32)):
        try:
            bad()
            errs.append("none")
        except Exception as e:   # noqa: BLE001
            errs.append(type(e).__name__)
    out["g16_errors"] = np.array(errs)
    # G19: the mask plug-in family (PointRendPredictor, SAMHumanPredictor*, utils/adaptive_mask_inpainting.py:1182-1454) run for
    # real on a short frame sequence, with the two third-party networks replaced by the deterministic stand-ins of
    # tests/fake_seg_backends.py (the classes' own logic -- category filter, merging, box policies, asset exclusion -- is what
    # is being pinned).  detectron2 / segment_anything are stubbed by the import hook above.
    sys.path.insert(0, os.path.join(root, "tests"))
    import fake_seg_backends as fb

    class _Inst:
        def __init__(self, masks, scores, classes):
            self.pred_masks, self.scores, self.pred_classes = torch.as_tensor(masks), torch.as_tensor(scores), torch.as_tensor(classes)

    def detectron_like(image):
        return {"instances": _Inst(*fb.fake_pointrend(image))}

    r.sam_model_registry = {"vit_h": lambda checkpoint=None: _Dummy()}
    r.SamPredictor = lambda sam: None
    frames = [fb.scene(0), fb.scene(1, n_person=2), fb.scene(2, person=False), fb.scene(3), fb.scene(4, n_person=2), fb.scene(5, person=False),
              fb.scene(6)]
    out["g19_frames"] = np.stack([f[0] for f in frames])
    out["g19_asset_mask"] = frames[0][1]
    cases = [("p_merge", r.PointRendPredictor, dict(merge_mode="merge"), False), ("p_maxconf", r.PointRendPredictor, dict(merge_mode="max-confidence"), False),
             ("ps", r.SAMHumanPredictor, dict(), False), ("ps_multi", r.SAMHumanPredictor, dict(is_sam_multitask_output=True), False),
             ("ps_ae", r.SAMHumanPredictorWithAssetExclusion, dict(), True),
             ("s_db_ae", r.SAMHumanPredictorWithDefaultBboxAssetExclusion, dict(), True),
             ("s_ab_ae", r.SAMHumanPredictorAccumulativeBboxAssetExclusion, dict(is_sam_multitask_output=True), True)]
    for tag, cls, kw, has_asset in cases:
        pred = cls(pointrend_thres=0.2, device="cpu", use_visualizer=False, **kw)
        pred.pointrend_seg_model = detectron_like
        if cls is not r.PointRendPredictor:
            pred.sam_seg_model = fb.FakeSam()
        if has_asset:
            pred.set_presumed_asset_mask(frames[0][1])
        masks, assets, kinds = [], [], []
        for k, (img, _) in enumerate(frames):
            if tag == "p_maxconf" and k in (2, 5):
                continue                         # np.argmax of an empty score list raises in the reference; not a usable case
            res = pred(img)
            if isinstance(res, tuple):          # SAMHumanPredictor's "nobody found" branch returns (mask, vis) -- :1278-1282
                kinds.append("tuple")
                res = {"mask": res[0], "asset_mask": None}
            else:
                kinds.append("dict")
            masks.append(res["mask"])
            assets.append(np.zeros_like(res["mask"]) if res["asset_mask"] is None else res["asset_mask"])
            kinds[-1] += ":none" if res["asset_mask"] is None else ":asset"
        out[f"g19_{tag}_masks"], out[f"g19_{tag}_assets"], out[f"g19_{tag}_kinds"] = np.stack(masks), np.stack(assets), np.array(kinds)
        if hasattr(pred, "initial_human_bbox") and pred.initial_human_bbox is not None:
            out[f"g19_{tag}_final_bbox"] = np.asarray(pred.initial_human_bbox)
    out.update(g20_dilation())
    np.savez_compressed(os.path.join(HERE, "inpaint_golden.npz"), **out)
    print("wrote inpaint_golden.npz", {k: getattr(v, "shape", None) for k, v in out.items()}, errs)


if __name__ == "__main__":
    main()
