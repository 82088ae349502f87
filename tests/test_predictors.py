"""Mask plug-in family: coma_amd.sd.predictors against the golden vectors G19, which
tests/golden/make_golden_inpaint.py produced by running the REAL reference classes
(utils/adaptive_mask_inpainting.py:1182-1454) on a 7-frame sequence with the two third-party networks replaced by the
deterministic stand-ins of tests/fake_seg_backends.py.  Masks and asset masks must be bit-identical, frame by frame."""
import os
import sys
import types

import numpy as np
import pytest

from tests import fake_seg_backends as fb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(ROOT, "tests", "golden", "inpaint_golden.npz"))


def _make(tag):
    from coma_amd.sd import predictors as P
    kw = dict(pointrend_thres=0.2, device="cpu", use_visualizer=False, pointrend_backend=fb.fake_pointrend)
    table = {"p_merge": (P.PointRendPredictor, dict(merge_mode="merge")), "p_maxconf": (P.PointRendPredictor, dict(merge_mode="max-confidence")),
             "ps": (P.SAMHumanPredictor, dict()), "ps_multi": (P.SAMHumanPredictor, dict(is_sam_multitask_output=True)),
             "ps_ae": (P.SAMHumanPredictorWithAssetExclusion, dict()), "s_db_ae": (P.SAMHumanPredictorWithDefaultBboxAssetExclusion, dict()),
             "s_ab_ae": (P.SAMHumanPredictorAccumulativeBboxAssetExclusion, dict(is_sam_multitask_output=True))}
    cls, extra = table[tag]
    if cls is not P.PointRendPredictor:
        kw["sam_backend"] = fb.FakeSam()
    return cls(**kw, **extra)


@pytest.mark.parametrize("tag", ["p_merge", "p_maxconf", "ps", "ps_multi", "ps_ae", "s_db_ae", "s_ab_ae"])
def test_predictor_sequence_matches_reference(g, tag):
    pred = _make(tag)
    if hasattr(pred, "set_presumed_asset_mask"):
        pred.set_presumed_asset_mask(g["g19_asset_mask"])
    frames = g["g19_frames"]
    use = [k for k in range(len(frames)) if not (tag == "p_maxconf" and k in (2, 5))]
    for n, k in enumerate(use):
        out = pred(frames[k])
        assert set(out) == {"asset_mask", "mask", "vis"} and out["vis"] is None
        assert out["mask"].dtype == np.uint8 and np.array_equal(out["mask"], g[f"g19_{tag}_masks"][n]), (tag, k)
        kind = str(g[f"g19_{tag}_kinds"][n])
        # (the reference's SAMHumanPredictor returns a (mask, vis) TUPLE when PointRend finds nobody -- its pipeline cannot
        #  consume that; here every branch returns the dict, with the same mask)
        if kind.endswith(":asset"):
            assert out["asset_mask"].dtype == np.uint8 and np.array_equal(out["asset_mask"], g[f"g19_{tag}_assets"][n])
        else:
            assert out["asset_mask"] is None
    if f"g19_{tag}_final_bbox" in g.files:
        assert np.array_equal(pred.initial_human_bbox, g[f"g19_{tag}_final_bbox"])


def test_default_bbox_can_be_preset_and_reset(g):
    pred = _make("s_db_ae")
    pred.set_presumed_asset_mask(g["g19_asset_mask"])
    seg = np.zeros((40, 48), np.uint8)
    seg[3:30, 5:20] = 1
    pred.set_initial_human_bbox(seg)
    assert pred.initial_human_bbox.tolist() == [5, 3, 20, 30]
    calls = []
    pred.pointrend_seg_model = lambda im: calls.append(1) or fb.fake_pointrend(im)
    pred(g["g19_frames"][0])
    assert not calls                                   # a preset box skips PointRend altogether
    pred.reset_initial_human_bbox()
    pred(g["g19_frames"][0])
    assert calls and pred.initial_human_bbox is not None


def test_selection_table_and_missing_dependency_error():
    from coma_amd.sd import predictors as P
    exp = {"p": "PointRendPredictor", "baseline": "PointRendPredictor", "ps": "SAMHumanPredictor", "ps_ae": "SAMHumanPredictorWithAssetExclusion",
           "s_pdb_ae": "SAMHumanPredictorWithDefaultBboxAssetExclusion", "s_db_ae": "SAMHumanPredictorWithDefaultBboxAssetExclusion",
           "s_ab_ae": "SAMHumanPredictorAccumulativeBboxAssetExclusion"}
    for key, name in exp.items():                      # src/generation/inpaint.py:73-110
        m = P.build_adaptive_mask_model(key, 0.2, enable_sam_multitask_output=True, pointrend_backend=fb.fake_pointrend, sam_backend=fb.FakeSam())
        assert type(m).__name__ == name and m.use_visualizer is False
        if key not in ("p", "baseline"):
            assert m.is_sam_multitask_output is True
    with pytest.raises(ValueError):
        P.build_adaptive_mask_model("nope", 0.2)
    if "detectron2" not in sys.modules:
        with pytest.raises(ImportError, match="detectron2"):
            P.build_adaptive_mask_model("p", 0.2)


def test_real_backends_are_wired_through_a_fake_detectron2(monkeypatch):
    """With a detectron2 / segment_anything that import, the default construction path builds the PointRend config the way the
    reference does (threshold, device, weights) and adapts `instances` to the backend tuple."""
    import torch
    from coma_amd.sd import predictors as P
    seen = {}

    class Cfg(types.SimpleNamespace):
        def merge_from_file(self, pth):
            seen["cfg_file"] = pth
    cfg = Cfg(MODEL=types.SimpleNamespace(ROI_HEADS=types.SimpleNamespace(), WEIGHTS=None, DEVICE=None))

    class DefaultPredictor:
        def __init__(self, c):
            seen["cfg"] = c

        def __call__(self, image):
            m, s, c = fb.fake_pointrend(image)
            return {"instances": types.SimpleNamespace(pred_masks=torch.as_tensor(m), scores=torch.as_tensor(s), pred_classes=torch.as_tensor(c))}
    mods = {"detectron2": types.ModuleType("detectron2"), "detectron2.config": types.ModuleType("c"), "detectron2.engine": types.ModuleType("e"),
            "detectron2.projects": types.ModuleType("p")}
    mods["detectron2.config"].get_cfg = lambda: cfg
    mods["detectron2.engine"].DefaultPredictor = DefaultPredictor
    mods["detectron2.projects"].point_rend = types.SimpleNamespace(add_pointrend_config=lambda c: seen.setdefault("pointrend_cfg", True))
    sam_mod = types.ModuleType("segment_anything")
    sam_mod.sam_model_registry = {"vit_h": lambda checkpoint=None: types.SimpleNamespace(to=lambda d: seen.setdefault("sam_device", d))}
    sam_mod.SamPredictor = lambda sam: fb.FakeSam()
    mods["segment_anything"] = sam_mod
    for k, v in mods.items():
        monkeypatch.setitem(sys.modules, k, v)
    pred = P.build_adaptive_mask_model("ps_ae", 0.35, device="cuda")
    assert seen["pointrend_cfg"] and seen["cfg_file"] == P.COCO_SEG_CONFIG_PTH and cfg.MODEL.WEIGHTS == P.COCO_SEG_WEIGHTS_PTH
    assert cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST == 0.35 and cfg.MODEL.DEVICE == "cuda" and seen["sam_device"] == "cuda"
    img, am = fb.scene(0)
    pred.set_presumed_asset_mask(am)
    out = pred(img)
    assert out["mask"].sum() > 0 and out["asset_mask"].sum() > 0
