"""The reference's PointRend predictor served by the device plan (coma_amd/seg/model.py) instead of detectron2:

  * `HipPointRendBackend(state, threshold)` -- the backend signature of coma_amd/sd/predictors.py: `backend(image_u8_HWC) -> (masks bool
    [N,H,W], scores [N], classes [N])` = what `DefaultPredictor(cfg)(image)["instances"]` holds (utils/adaptive_mask_inpainting.py:1227-1233),
    and `backend.instances(image_bgr)` = the record of the post-inpaint stage (src/generation/segment_human.py:152-166).  Plans are built
    per image size on first use (batch 1) and replayed as hipGraphs afterwards.
  * `HipPointRendPredictor` -- `PointRendPredictor` (utils/adaptive_mask_inpainting.py:1182-1236) as a DEVICE plug-in of the adaptive loop:
    `accepts_device_tensor`, the decoded x0 stays in HBM, `predict_batch(uint8 [B,H,W,3]) -> {"mask": uint8 [B,H,W]}` is one plan replay for the
    B images of a re-estimation, and the merged person mask (merge_mode "merge": np.any over the instances of `cat_id_to_focus`) never leaves
    the device before `sd_mask_adapt_batched` dilates it.

Weights: a detectron2 checkpoint (`model_final_edd263.pkl`, constants/segmentation.py:5 of the reference) through
`weights.load_detectron2_pkl`, or seeded random parameters of the same architecture (`weights.random_state`) when none is provisioned.
"""
from __future__ import annotations

import numpy as np
import torch

from . import weights as W
from .model import HipPointRend


class HipPointRendBackend:
    def __init__(self, state, threshold, device="cuda", cat_id=0, keep_masks=True, detections_per_image=100):
        self.state, self.threshold, self.device, self.cat_id, self.keep_masks = state, float(threshold), torch.device(device), cat_id, keep_masks
        self.detections_per_image = detections_per_image
        self._plans = {}

    @classmethod
    def from_checkpoint(cls, path, threshold, device="cuda", **kw):
        return cls(W.load_detectron2_pkl(path), threshold, device, **kw)

    def plan(self, batch, height, width):
        key = (batch, height, width)
        if key not in self._plans:
            self._plans[key] = HipPointRend(self.state, batch, height, width, self.device, score_thresh=self.threshold, keep_masks=self.keep_masks,
                                            cat_id=self.cat_id, detections_per_image=self.detections_per_image)
        return self._plans[key]

    def _run(self, image):
        img = torch.as_tensor(np.ascontiguousarray(image)) if not isinstance(image, torch.Tensor) else image
        assert img.dim() == 3 and img.shape[2] == 3 and img.dtype == torch.uint8, "uint8 [H, W, 3] image (channel order as handed to DefaultPredictor)"
        plan = self.plan(1, img.shape[0], img.shape[1])
        plan(img[None])
        return plan

    def instances(self, image):
        """-> dict(pred_boxes f32 [n,4], scores f32 [n], pred_classes i64 [n], pred_masks bool [n,H,W], raw=None): detector_postprocess'ed."""
        rec = self._run(image).instances(0)
        rec["raw"] = None
        return rec

    def __call__(self, image):
        rec = self.instances(image)
        return rec["pred_masks"], rec["scores"], rec["pred_classes"]

    def person_masks(self, images_u8):
        """uint8 [B,H,W,3] device tensor -> uint8 [B,H,W] device tensor: np.any over the masks of `cat_id`, one plan replay for the batch."""
        B, H, Wd, _ = images_u8.shape
        return self.plan(B, H, Wd)(images_u8)["person"]


class HipPointRendPredictor:
    """utils/adaptive_mask_inpainting.py:1182-1236 with the detector on the device.  Same constructor arguments and result dictionary."""
    accepts_device_tensor = True

    def __init__(self, cat_id_to_focus=0, pointrend_thres=0.9, device="cuda", use_visualizer=False, merge_mode="merge", *, state=None, weights_pth=None,
                 detections_per_image=100):
        assert merge_mode in ["merge", "max-confidence"], f"'merge_mode': {merge_mode} not implemented."
        if use_visualizer:
            raise NotImplementedError("use_visualizer draws with detectron2's Visualizer; the device plug-in has none")
        if state is None:
            if weights_pth is None:
                raise ValueError("HipPointRendPredictor needs `state` (parameters) or `weights_pth` (a detectron2 .pkl)")
            state = W.load_detectron2_pkl(weights_pth)
        self.cat_id_to_focus, self.merge_mode, self.use_visualizer, self.device = cat_id_to_focus, merge_mode, False, device
        self.pointrend_seg_model = HipPointRendBackend(state, pointrend_thres, device, cat_id=cat_id_to_focus, keep_masks=merge_mode != "merge",
                                                       detections_per_image=detections_per_image)

    def predict_batch(self, images_u8):
        if self.merge_mode != "merge":                     # max-confidence picks ONE instance: per image through the host-side record
            return {"mask": torch.stack([torch.as_tensor(self(im)["mask"]).to(images_u8.device) for im in images_u8]), "vis": None, "asset_mask": None}
        return {"mask": self.pointrend_seg_model.person_masks(images_u8), "vis": None, "asset_mask": None}

    def __call__(self, image):
        on_device = isinstance(image, torch.Tensor) and image.is_cuda
        if self.merge_mode == "merge":
            img = image if isinstance(image, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(image))
            mask = self.pointrend_seg_model.person_masks(img[None].to(self.device))[0]
        else:
            masks, scores, classes = self.pointrend_seg_model(image)
            keep = classes == self.cat_id_to_focus
            mask = torch.as_tensor(masks[keep][np.argmax(scores[keep])].astype(np.uint8))      # raises on no person, as the reference's argmax does
        return {"asset_mask": None, "mask": mask if on_device else mask.cpu().numpy().astype(np.uint8), "vis": None}
