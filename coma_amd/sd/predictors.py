"""Human-mask plug-ins of the adaptive-mask loop: the reference's PointRend / SAM predictor family
(/root/reference/utils/adaptive_mask_inpainting.py:1182-1454) with the same class names, constructor arguments, methods
(`merge_mask`, `set_presumed_asset_mask`, `set_initial_human_bbox`, `reset_initial_human_bbox`) and call contract:

    predictor(image_u8_HWC) -> {"mask": u8 [H,W], "vis": image or None, "asset_mask": u8 [H,W] or None},  .use_visualizer

The segmentation networks themselves are third-party (detectron2's PointRend via the ODISE clone, segment-anything;
INSTALL.md:15-34) and are imported LAZILY: constructing a predictor without them raises a clear ImportError, nothing is
stubbed.  Tests (and anyone with another detector) inject `pointrend_backend` / `sam_backend` instead:
    pointrend_backend(image) -> (masks bool [N,H,W], scores [N], classes [N])
    sam_backend.set_image(image); sam_backend.predict(box=xyxy, multimask_output=bool) -> (masks [K,H,W], scores [K], logits)
The five classes differ only in where the SAM box prompt comes from and in whether the presumed asset is carved out, so
one implementation carries two policy attributes instead of five copies of `__call__`:
    _box_policy   "current"   box of this frame's PointRend person mask                       (SAMHumanPredictor[WithAssetExclusion])
                  "sticky"    first PointRend box (or the one given by set_initial_human_bbox), PointRend skipped afterwards
                  "accumulate" union of all PointRend boxes seen so far
    _exclude_asset            AND NOT the SAM mask prompted with the presumed asset's box
"""
from __future__ import annotations

import numpy as np

from .pipeline import merge_bbox, seg2bbox

# constants/segmentation.py:4-5 and utils/adaptive_mask_inpainting.py:1239-1243 of the reference
COCO_SEG_CONFIG_PTH = "./imports/pointrend/config/pointrend_rcnn_R_50_FPN_3x_coco.yaml"
COCO_SEG_WEIGHTS_PTH = "./imports/pointrend/weights/model_final_edd263.pkl"
SAM_MODEL_PTH_DICT = {"vit_h": "./imports/segment-anything/sam_vit_h_4b8939.pth", "vit_l": "./imports/segment-anything/sam_vit_h_4b8939.pth",
                      "vit_b": "./imports/segment-anything/sam_vit_h_4b8939.pth"}


def pointrend_backend(threshold, device="cuda", config_pth=None, weights_pth=None, state=None):
    """Public name of the detector shared by the adaptive-mask plug-ins and src/generation/segment_human.py: the device plan of
    coma_amd/seg (no detectron2 needed) when the checkpoint is there or parameters are given, detectron2's DefaultPredictor otherwise."""
    import os
    pth = weights_pth or COCO_SEG_WEIGHTS_PTH
    if state is not None or os.path.exists(pth):
        from ..seg.predictor import HipPointRendBackend
        return HipPointRendBackend(state, threshold, device) if state is not None else HipPointRendBackend.from_checkpoint(pth, threshold, device)
    return _detectron2_pointrend(threshold, device, config_pth, weights_pth)


def _detectron2_pointrend(threshold, device, config_pth=None, weights_pth=None):
    """detectron2 DefaultPredictor for PointRend, wrapped to the backend signature.  Raises ImportError when detectron2 is absent."""
    try:
        from detectron2.config import get_cfg
        from detectron2.engine import DefaultPredictor
        from detectron2.projects import point_rend
    except ImportError as e:                                  # pragma: no cover (no detectron2 in the build image)
        raise ImportError("PointRendPredictor needs detectron2 with the PointRend project (INSTALL.md of the reference, ODISE clone); "
                          "install it, pass pointrend_backend=..., or run with --mask_model synthetic") from e
    cfg = get_cfg()
    point_rend.add_pointrend_config(cfg)
    cfg.merge_from_file(config_pth or COCO_SEG_CONFIG_PTH)
    cfg.MODEL.WEIGHTS = weights_pth or COCO_SEG_WEIGHTS_PTH
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = threshold
    cfg.MODEL.DEVICE = device
    model = DefaultPredictor(cfg)

    def run(image):
        inst = model(image)["instances"]
        return (inst.pred_masks.detach().cpu().numpy(), inst.scores.detach().cpu().numpy(), inst.pred_classes.detach().cpu().numpy())

    def instances(image):
        """everything the post-inpaint segmentation stage stores (src/generation/segment_human.py:152-166), plus the raw object"""
        inst = model(image)["instances"]
        return dict(pred_boxes=inst.pred_boxes.tensor.detach().cpu().numpy(), scores=inst.scores.detach().cpu().numpy(),
                    pred_classes=inst.pred_classes.detach().cpu().numpy(), pred_masks=inst.pred_masks.detach().cpu().numpy(), raw=inst)
    run.model, run.instances = model, instances
    return run


def _segment_anything(sam_key, device):
    try:
        from segment_anything import SamPredictor, sam_model_registry
    except ImportError as e:                                  # pragma: no cover
        raise ImportError("SAM predictors need the segment-anything package and its checkpoint (INSTALL.md of the reference); "
                          "install it or pass sam_backend=...") from e
    sam = sam_model_registry[sam_key](checkpoint=SAM_MODEL_PTH_DICT[sam_key])
    sam.to(device)
    return SamPredictor(sam)


class PointRendPredictor:
    """utils/adaptive_mask_inpainting.py:1182-1236."""
    _box_policy = None
    _exclude_asset = False

    def __init__(self, cat_id_to_focus=0, pointrend_thres=0.9, device="cuda", use_visualizer=False, merge_mode="merge", *,
                 pointrend_backend=None):
        self.cat_id_to_focus = cat_id_to_focus
        assert merge_mode in ["merge", "max-confidence"], f"'merge_mode': {merge_mode} not implemented."
        self.merge_mode = merge_mode
        self.use_visualizer = use_visualizer
        self.device = device
        self.pointrend_seg_model = pointrend_backend if pointrend_backend is not None else globals()["pointrend_backend"](pointrend_thres, device)

    # ---- pieces
    def merge_mask(self, masks, scores=None):
        if self.merge_mode == "merge":
            return np.any(masks, axis=0)
        return masks[np.argmax(scores)]                       # "max-confidence"

    def vis_seg_on_img(self, image, mask):                    # pragma: no cover (needs detectron2's Visualizer)
        import torch
        from detectron2.data import MetadataCatalog
        from detectron2.structures import Instances
        from detectron2.utils.visualizer import ColorMode, Visualizer
        mask = torch.as_tensor(mask)
        v = Visualizer(image, MetadataCatalog.get("coco_2017_val"), scale=0.5, instance_mode=ColorMode.IMAGE_BW)
        inst = Instances(image_size=image.shape[:2], pred_masks=mask if mask.dim() == 3 else mask[None])
        return v.draw_instance_predictions(inst.to("cpu")).get_image()

    def _person_mask(self, image):
        masks, scores, classes = self.pointrend_seg_model(image)
        keep = np.asarray(classes) == self.cat_id_to_focus
        return self.merge_mask(np.asarray(masks)[keep], scores=np.asarray(scores)[keep])

    def _result(self, image, mask, asset_mask=None):
        return {"asset_mask": None if asset_mask is None else asset_mask.astype(np.uint8), "mask": mask.astype(np.uint8),
                "vis": self.vis_seg_on_img(image, mask) if self.use_visualizer else None}

    def __call__(self, image, debug=False):
        if self._box_policy is None:                          # PointRend only
            return self._result(image, self._person_mask(image))
        box = self._prompt_box(image)
        if box is None:                                       # nobody found by PointRend: hand the empty mask on (:1278, :1322, :1381)
            return self._result(image, self._empty)
        self.sam_seg_model.set_image(image)
        masks, scores, _ = self.sam_seg_model.predict(box=box, multimask_output=self.is_sam_multitask_output)
        mask = self.merge_mask(masks, scores=scores)
        if not self._exclude_asset:
            return self._result(image, mask)
        a_masks, a_scores, _ = self.sam_seg_model.predict(box=self.presumed_asset_bbox, multimask_output=self.is_sam_multitask_output)
        asset = self.merge_mask(a_masks, scores=a_scores)
        return self._result(image, np.logical_and(mask, np.logical_not(asset)), asset)


class SAMHumanPredictor(PointRendPredictor):
    """:1246-1296  PointRend proposes, SAM refines inside the person's box."""
    _box_policy = "current"

    def __init__(self, sam_key="vit_h", device="cuda", merge_mode="merge", is_sam_multitask_output=False, *pointrend_args,
                 sam_backend=None, **pointrend_kwargs):
        super().__init__(*pointrend_args, **pointrend_kwargs)
        self.sam_seg_model = sam_backend if sam_backend is not None else _segment_anything(sam_key, device)
        self.is_sam_multitask_output = is_sam_multitask_output
        self.initial_human_bbox = None

    def _prompt_box(self, image):
        if self._box_policy == "sticky" and self.initial_human_bbox is not None:
            return self.initial_human_bbox                    # PointRend is not run again (:1368-1371)
        person = self._person_mask(image)
        if person.sum() == 0:
            self._empty = person
            return None
        box = seg2bbox(person)
        if self._box_policy == "sticky":
            self.initial_human_bbox = box
        elif self._box_policy == "accumulate":
            self.initial_human_bbox = box if self.initial_human_bbox is None else merge_bbox([self.initial_human_bbox, box])
            box = self.initial_human_bbox
        return box


class SAMHumanPredictorWithAssetExclusion(SAMHumanPredictor):
    """:1299-1351  ... and the SAM mask of the presumed asset is removed from the person mask."""
    _exclude_asset = True

    def set_presumed_asset_mask(self, presumed_asset_mask: np.ndarray):
        self.presumed_asset_mask = presumed_asset_mask
        self.presumed_asset_bbox = seg2bbox(presumed_asset_mask)


class SAMHumanPredictorWithDefaultBboxAssetExclusion(SAMHumanPredictorWithAssetExclusion):
    """:1356-1408  the person box is fixed by the first detection (or given), later frames go straight to SAM."""
    _box_policy = "sticky"

    def set_initial_human_bbox(self, human_seg_np):
        self.initial_human_bbox = seg2bbox(human_seg_np)

    def reset_initial_human_bbox(self):
        self.initial_human_bbox = None


class SAMHumanPredictorAccumulativeBboxAssetExclusion(SAMHumanPredictorWithDefaultBboxAssetExclusion):
    """:1411-1454  the person box grows to the union of every detection so far."""
    _box_policy = "accumulate"


# adaptive_mask_model_type of src/generation/inpaint.py:73-110 -> (class, takes the SAM multitask flag)
PREDICTOR_TABLE = {
    "p": (PointRendPredictor, False), "baseline": (PointRendPredictor, False), "ps": (SAMHumanPredictor, True),
    "ps_ae": (SAMHumanPredictorWithAssetExclusion, True), "s_pdb_ae": (SAMHumanPredictorWithDefaultBboxAssetExclusion, True),
    "s_db_ae": (SAMHumanPredictorWithDefaultBboxAssetExclusion, True), "s_ab_ae": (SAMHumanPredictorAccumulativeBboxAssetExclusion, True),
}


def build_adaptive_mask_model(adaptive_mask_model_type, pointrend_threshold, use_visualizer=False, enable_sam_multitask_output=False,
                              device="cuda", **backends):
    """The selection of src/generation/inpaint.py:73-110."""
    if adaptive_mask_model_type not in PREDICTOR_TABLE:
        raise ValueError(f"adaptive_mask_model_type '{adaptive_mask_model_type}' not one of {sorted(PREDICTOR_TABLE)}")
    cls, sam = PREDICTOR_TABLE[adaptive_mask_model_type]
    if cls is PointRendPredictor and not use_visualizer and "pointrend_backend" not in backends:
        import os
        if backends.get("pointrend_state") is not None or os.path.exists(COCO_SEG_WEIGHTS_PTH):
            # PointRend only: the whole plug-in runs on the device (x0 image, detector, merged mask never leave HBM; one call per batch)
            from ..seg.predictor import HipPointRendPredictor
            return HipPointRendPredictor(pointrend_thres=pointrend_threshold, device=device, state=backends.get("pointrend_state"),
                                         weights_pth=None if backends.get("pointrend_state") is not None else COCO_SEG_WEIGHTS_PTH)
    backends.pop("pointrend_state", None)
    kw = dict(pointrend_thres=pointrend_threshold, device=device, use_visualizer=use_visualizer)
    if sam:
        kw["is_sam_multitask_output"] = enable_sam_multitask_output
    else:
        backends.pop("sam_backend", None)
    return cls(**kw, **backends)


class PerItemState:
    """One predictor instance serving the B images of a batched pipeline call (an addition: the reference runs one image per call,
    so its predictors keep the state of "the current item" in plain attributes -- `presumed_asset_mask` / `presumed_asset_bbox`
    (:1336-1339), `initial_human_bbox` (:1283-1296, :1380-1386; the `s_ab_ae` type accumulates its boxes there)).  `select(b)`
    parks the attributes of the slot that was current and brings in those of slot b; the networks (the only large members) are
    shared.  The pipeline calls `select(b)` before it hands image b to the plug-in; the harness calls it before priming item b."""
    STATE_ATTRS = ("initial_human_bbox", "presumed_asset_mask", "presumed_asset_bbox", "_empty")
    _MISSING = object()

    def __init__(self, model, batch):
        self.__dict__["model"] = model
        self.__dict__["slot"] = 0
        self.__dict__["states"] = [None] * batch
        first = {k: getattr(model, k, self._MISSING) for k in self.STATE_ATTRS}
        for b in range(batch):
            self.states[b] = dict(first)            # every slot starts from the state the model was constructed with

    def select(self, b, inherit_from=None):
        """inherit_from: slot whose CURRENT state slot b starts from (the harness: the item before it in list order, so that an item
        that is not primed sees what the reference's sequential loop would have left in the plug-in)."""
        m = self.model
        if inherit_from is not None and inherit_from != b:
            src = ({k: getattr(m, k, self._MISSING) for k in self.STATE_ATTRS} if inherit_from == self.slot else self.states[inherit_from])
            if b == self.slot:                      # overwrite the live attributes
                self.__dict__["slot"] = -1
            self.states[b] = dict(src)
        if b == self.slot:
            return
        if self.slot >= 0:
            self.states[self.slot] = {k: getattr(m, k, self._MISSING) for k in self.STATE_ATTRS}
        for k, v in self.states[b].items():
            if v is self._MISSING:
                if hasattr(m, k):
                    delattr(m, k)
            else:
                setattr(m, k, v)
        self.__dict__["slot"] = b

    def __call__(self, image, *a, **kw):
        return self.model(image, *a, **kw)

    def __getattr__(self, name):                    # everything else (use_visualizer, accepts_device_tensor, set_* hooks) is the model's
        return getattr(self.model, name)

    def __setattr__(self, name, value):
        setattr(self.model, name, value)
