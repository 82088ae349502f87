"""Host logic around the inpainting path that needs no GPU: the post-inpaint segmentation stage with a
deterministic stand-in for PointRend, the down-sampling CLIs' surface (8b-4), the inpaint CLI's plug-in selection and the
legacy VAE attention key conversion."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from tests import fake_seg_backends as fb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeDetector:
    """backend contract of coma_amd.sd.predictors.pointrend_backend: __call__ -> (masks, scores, classes), .instances -> dict"""
    def __call__(self, image):
        return fb.fake_pointrend(image[:, :, ::-1])

    def instances(self, image_bgr):
        m, s, c = fb.fake_pointrend(image_bgr[:, :, ::-1])          # the stand-in thinks in RGB
        boxes = np.array([[np.where(x.any(0))[0].min(), np.where(x.any(1))[0].min(), np.where(x.any(0))[0].max() + 1,
                           np.where(x.any(1))[0].max() + 1] for x in m], dtype=np.float32).reshape(-1, 4)
        return dict(pred_boxes=boxes, scores=s, pred_classes=c, pred_masks=m, raw=None)


def _tree(tmp_path):
    from PIL import Image
    base = tmp_path / "inpaintings" / "BEHAVE" / "backpack" / "behave_asset" / "view:00000" / "mask0"
    made = []
    for prompt, person in (("a person carries a backpack", True), ("a person carries a backpack, full body", False),
                           ("a person carries a backpack, upper body", True)):
        d = base / prompt
        d.mkdir(parents=True)
        for k in range(2):
            img, _ = fb.scene(10 + k, person=person)
            Image.fromarray(img).save(d / f"{k:06}.png")
            made.append(str(d / f"{k:06}.png"))
    (tmp_path / "inpaintings" / "BEHAVE" / "backpack" / "unknown_asset" / "view:00000" / "mask0" / "p").mkdir(parents=True)
    Image.fromarray(fb.scene(0)[0]).save(tmp_path / "inpaintings" / "BEHAVE" / "backpack" / "unknown_asset" / "view:00000" / "mask0" / "p" / "000000.png")
    return made


def test_segment_human_work_list_outputs_and_slices(tmp_path):
    from src.generation import segment_human as sh
    _tree(tmp_path)
    common = dict(supercategories=["behave"], categories=None, prompts=None, inpaint_dir=str(tmp_path / "inpaintings"),
                  save_dir=str(tmp_path / "segs"), threshold=0.8, save_full=False, save_vis_in_same_folder=False, save_image=True,
                  verbose=False, detector=FakeDetector())
    # (the reference's processes run concurrently and see the same list; run one after the other, skip_done would shrink it)
    done0 = sh.human_segmentation_coco(parallel_num=2, parallel_idx=0, skip_done=False, **common)
    done1 = sh.human_segmentation_coco(parallel_num=2, parallel_idx=1, skip_done=False, **common)
    # 4 images qualify (no suffix / ", full body"; the ", upper body" prompt and the unregistered asset are skipped): sub = 4//2+1 = 3
    assert len(done0) == 3 and len(done1) == 1 and not set(done0) & set(done1)
    assert all("upper body" not in p and "unknown_asset" not in p for p in done0 + done1)
    with_person = sorted(p for p in done0 + done1 if "full body" not in p)
    with open(with_person[0], "rb") as h:
        rec = pickle.load(h)
    assert set(rec) == {"num_instances", "image_height", "image_width", "pred_boxes", "scores", "pred_classes", "pred_masks"}
    assert rec["image_height"] == 40 and rec["image_width"] == 48 and rec["pred_masks"].shape[0] == rec["num_instances"] == len(rec["scores"])
    assert rec.num_instances == rec["num_instances"]                          # attribute access, as the reference's EasyDict
    person_png = with_person[0].replace(".pickle", ".png")
    assert os.path.exists(person_png)                                         # first person mask, saved as 8-bit image
    nohuman = [p for p in done0 + done1 if "full body" in p]
    assert nohuman and not os.path.exists(nohuman[0].replace(".pickle", ".png"))
    # skip_done: nothing left
    assert sh.human_segmentation_coco(parallel_num=1, parallel_idx=0, skip_done=True, **common) == []


def test_segment_human_cli_flags_match_reference():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "src", "generation", "segment_human.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--supercategories", "--categories", "--prompts", "--inpaint_dir", "--save_dir", "--mode", "--threshold", "--parallel_num",
                 "--parallel_idx", "--disable_save_full", "--save_vis_in_same_folder", "--save_image", "--skip_done", "--verbose", "--seed"):
        assert flag in out.stdout


def test_downsample_cli_flags_and_obj_loader(tmp_path):
    for script, flags in (("downsample_human.py", ("--simplify_method", "--num_human_downsample_points_list", "--use_watertight", "--skip_done", "--debug", "--seed")),
                          ("downsample_objects.py", ("--simplify_method", "--skip_done", "--debug", "--seed", "--obj_pth", "--asset_downsample_dir"))):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "src", "coma", script), "--help"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and all(f in out.stdout for f in flags), script
    from coma_amd.downsample import load_obj, sample_uniform
    (tmp_path / "q.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1 4/1/1\nf -4 -3 -2\n")
    v, f = load_obj(str(tmp_path / "q.obj"))
    assert v.shape == (4, 3) and f.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2]]
    n = np.tile([0.0, 0.0, 1.0], (4, 1))
    p1, n1 = sample_uniform(v, f[:2], n, 50, seed=3)
    p2, _ = sample_uniform(v, f[:2], n, 50, seed=3)
    assert np.array_equal(p1, p2) and (p1[:, :2] >= 0).all() and (p1[:, :2] <= 1).all() and np.allclose(p1[:, 2], 0) and np.allclose(n1, n[:1])


def test_inpaint_cli_selects_plugins_and_requires_text_encoder(monkeypatch):
    from src.generation import inpaint
    p = inpaint.build_parser()
    assert p.get_default("mask_model") == "auto" and p.get_default("adaptive_mask_model_type") == "p"
    calls = []

    class M:
        def set_presumed_asset_mask(self, m): calls.append(("asset", m.sum()))
        def reset_initial_human_bbox(self): calls.append(("reset",))
        def set_initial_human_bbox(self, m): calls.append(("bbox", m.sum()))
    seg, dm = np.ones((4, 4), bool), np.ones((4, 4), bool)
    for kind, exp in (("p", []), ("ps", []), ("ps_ae", ["asset"]), ("s_db_ae", ["asset", "reset", "bbox"]), ("s_pdb_ae", ["asset", "reset"]),
                      ("s_ab_ae", ["asset", "reset"])):          # reference src/generation/inpaint.py:325-337
        calls.clear()
        inpaint.prime_mask_model(M(), kind, seg, dm)
        assert [c[0] for c in calls] == exp, kind


def test_legacy_vae_attention_keys_are_converted():
    import torch
    from coma_amd.sd import weights
    shapes = weights.vae_shapes()
    state = weights.random_state(shapes, seed=1)
    legacy = {}
    ren = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    for k, v in state.items():
        parts = k.split(".")
        if "attentions" in parts and parts[-2] in ("to_q", "to_k", "to_v") or (len(parts) > 2 and ".".join(parts[-3:-1]) == "to_out.0"):
            new = ren["to_out.0"] if ".".join(parts[-3:-1]) == "to_out.0" else ren[parts[-2]]
            base = parts[:-3] if ".".join(parts[-3:-1]) == "to_out.0" else parts[:-2]
            k2 = ".".join(base + [new, parts[-1]])
            legacy[k2] = v[:, :, None, None] if (parts[-1] == "weight" and new != "proj_attn") else v
        else:
            legacy[k] = v
    assert any(".query." in k for k in legacy) and not any(".to_q." in k for k in legacy if "attentions" in k)
    with pytest.raises(ValueError):
        weights.check_state({k: v for k, v in legacy.items() if ".query." not in k}, shapes, "VAE")
    back = weights.check_state(legacy, shapes, "VAE")
    assert set(back) == set(shapes) and all(torch.equal(back[k], state[k]) for k in shapes)
