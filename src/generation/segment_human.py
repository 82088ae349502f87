"""Post-inpaint human segmentation: PointRend on every final HOI image, the stage right after the
inpainting loop -- and the same detector the adaptive-mask plug-ins already hold (coma_amd.sd.predictors.pointrend_backend).

CLI surface, work list, slice rule and outputs of the reference's ``src/generation/segment_human.py``:
  * inputs  {inpaint_dir}/{SC}/{C}/{asset}/{view}/{mask}/{prompt}/{id}.png of registered assets (utils/prepare_renders.py:36-62),
    only prompts without a viewpoint suffix or with ", full body" (:69-76);
  * outputs {save_dir}/.../{id}.pickle -- the full detectron2 Instances (default) or, with --disable_save_full, the
    dependency-free record {num_instances, image_height, image_width, pred_boxes, scores, pred_classes, pred_masks} (:152-166);
    with --save_image the first person mask as {id}.png and a visualisation (HUMAN / NO-HUMAN folders or vis:{id}.png);
  * per-process slice ``sub = len // n + 1`` of the list sorted by pickle path (:116-121).
The detector is third party (detectron2, unpinned); any callable with the backend's `.instances(image_bgr)` method can be
passed to `human_segmentation_coco(detector=...)` -- tests use a deterministic stand-in.
"""
import argparse
import os
import pickle
import sys
from glob import glob

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from coma_amd.misc import EasyDict  # noqa: E402  (easydict when installed, an attribute-dict otherwise)
from constants.generation.inpaint_config import CATEGORY2ASSET  # noqa: E402
from constants.metadata import DEFAULT_SEED  # noqa: E402


def prepare_inpainting_pths(inpaint_dir, supercategories, categories, prompts):
    out = []
    for pth in sorted(glob(f"{inpaint_dir}/*/*/*/*/*/*/*.png")):
        sc_str, c_str, asset_id = pth.split("/")[-7:-4]
        if asset_id not in CATEGORY2ASSET.get(sc_str.replace(":", "/"), {}).get(c_str.replace(":", "/"), []):
            continue
        if supercategories is not None and sc_str.lower() not in supercategories:
            continue
        if categories is not None and c_str.lower() not in categories:
            continue
        if prompts is not None and pth.split("/")[-2].lower() not in prompts:
            continue
        out.append(pth)
    return sorted(out)


def build_work_list(inpaint_pths, save_dir, save_vis_in_same_folder, skip_done, verbose=False):
    items = []
    for pth in sorted(inpaint_pths):
        sc_str, c_str, asset_id, view_id, mask_id, prompt, id_ext = pth.split("/")[-7:]
        inpaint_id, ext = id_ext.split(".")
        parts = prompt.split(",")
        if len(parts) > 1 and parts[-1].strip() != "full body":          # full-body prompts only
            continue
        assert ext == "png", "Inpainting must have '.png' extension"
        rdir = f"{save_dir}/{sc_str}/{c_str}/{asset_id}/{view_id}/{mask_id}/{prompt}"
        seg_pth = f"{rdir}/{inpaint_id}.pickle"
        if save_vis_in_same_folder:
            vis_h = f"{save_dir}_VIZ/{sc_str}/{c_str}/{asset_id}/HUMAN/{view_id}:{mask_id}:{prompt}:{inpaint_id}.png"
            vis_n = f"{save_dir}_VIZ/{sc_str}/{c_str}/{asset_id}/NO-HUMAN/{view_id}:{mask_id}:{prompt}:{inpaint_id}.png"
            os.makedirs(os.path.dirname(vis_h), exist_ok=True)
            os.makedirs(os.path.dirname(vis_n), exist_ok=True)
        else:
            vis_h = vis_n = f"{rdir}/vis:{inpaint_id}.png"
        if os.path.exists(seg_pth) and skip_done:
            if verbose:
                print(f"Continueing '{seg_pth}' Since Already Processed...")
            continue
        os.makedirs(rdir, exist_ok=True)
        items.append(dict(inpaint_pth=pth, vis_img_save_pth_human=vis_h, vis_img_save_pth_nohuman=vis_n,
                          result_img_save_pth=f"{rdir}/{inpaint_id}.png", result_seg_save_pth=seg_pth))
    return sorted(items, key=lambda x: x["result_seg_save_pth"])


def read_bgr(pth):
    """cv2.imread equivalent (8-bit BGR) without OpenCV."""
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(pth).convert("RGB"))[:, :, ::-1])


def detectron2_visualizer():
    """The reference's HUMAN / NO-HUMAN picture (src/generation/segment_human.py:139-140): detectron2's Visualizer over the RGB image
    at half scale, grey-scale background, drawn from the raw Instances the detectron2 backend keeps under "raw"."""
    from detectron2.data import MetadataCatalog
    from detectron2.utils.visualizer import ColorMode, Visualizer
    meta = MetadataCatalog.get("coco_2017_val")

    def draw(image_rgb, inst):
        v = Visualizer(image_rgb, meta, scale=0.5, instance_mode=ColorMode.IMAGE_BW)
        return v.draw_instance_predictions(inst["raw"].to("cpu")).get_image()
    return draw


def human_segmentation_coco(supercategories, categories, prompts, inpaint_dir, save_dir, threshold, parallel_num, parallel_idx, save_full,
                            save_vis_in_same_folder, save_image, skip_done, verbose, detector=None, visualizer=None):
    """detector: backend with `.instances(image_bgr) -> dict(pred_boxes, scores, pred_classes, pred_masks, raw)`; default = PointRend
    through detectron2 (ImportError with instructions when it is not installed).  visualizer(image_rgb, instances_dict) -> RGB image."""
    if detector is None:
        from coma_amd.sd.predictors import pointrend_backend
        detector = pointrend_backend(threshold, "cuda")
        if save_image and visualizer is None:
            visualizer = detectron2_visualizer()
    items = build_work_list(prepare_inpainting_pths(inpaint_dir, supercategories, categories, prompts), save_dir, save_vis_in_same_folder,
                            skip_done, verbose)
    sub = len(items) // parallel_num + 1
    done = []
    for it in items[parallel_idx * sub:(parallel_idx + 1) * sub]:
        im = read_bgr(it["inpaint_pth"])
        H, W, _ = im.shape
        inst = detector.instances(im)
        if save_image:
            from PIL import Image
            if visualizer is not None:
                vis = visualizer(im[:, :, ::-1], inst)
                Image.fromarray(np.asarray(vis)).save(it["vis_img_save_pth_human"] if 0 in inst["pred_classes"] else it["vis_img_save_pth_nohuman"])
            person = inst["pred_masks"][inst["pred_classes"] == 0]
            if len(person) > 0:
                Image.fromarray(person[0]).convert("L").save(it["result_img_save_pth"])
        if save_full and inst.get("raw") is not None:
            payload = inst["raw"].to("cpu") if hasattr(inst["raw"], "to") else inst["raw"]
        else:
            payload = EasyDict(dict(num_instances=len(inst["scores"]), image_height=H, image_width=W, pred_boxes=inst["pred_boxes"],
                                    scores=inst["scores"], pred_classes=inst["pred_classes"], pred_masks=inst["pred_masks"]))
        with open(it["result_seg_save_pth"], "wb") as handle:
            pickle.dump(payload, handle, protocol=pickle.HIGHEST_PROTOCOL)
        done.append(it["result_seg_save_pth"])
    return done


def run_human_segmentation(mode, **kwargs):
    assert mode in ["coco", "lvis", "odise"], f"Segmentation Mode: {mode} --> Not implemented..."
    if mode != "coco":
        raise NotImplementedError
    return human_segmentation_coco(**kwargs)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--supercategories", type=str, nargs="+")
    p.add_argument("--categories", type=str, nargs="+")
    p.add_argument("--prompts", type=str, nargs="+")
    p.add_argument("--inpaint_dir", type=str, default="results/generation/inpaintings")
    p.add_argument("--save_dir", type=str, default="results/generation/human_segs")
    p.add_argument("--mode", type=str, choices=["coco", "lvis", "odise"], default="coco")
    p.add_argument("--threshold", type=float, default=0.8, nargs="?", choices=[0.8, 0.95])
    p.add_argument("--parallel_num", type=int, default=1)
    p.add_argument("--parallel_idx", type=int, default=0)
    p.add_argument("--disable_save_full", action="store_true", help="If False, saves essential information only (without detectron2 dependency)")
    p.add_argument("--save_vis_in_same_folder", action="store_true", default=False)
    p.add_argument("--save_image", action="store_true", default=False)
    p.add_argument("--skip_done", action="store_true")
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--seed", type=int, default=DEFAULT_SEED)
    return p


if __name__ == "__main__":
    args = build_parser().parse_args()
    for name in ("supercategories", "categories", "prompts"):
        if getattr(args, name) is not None:
            setattr(args, name, [x.lower() for x in getattr(args, name)])
    from utils.reproducibility import seed_everything
    seed_everything(args.seed)
    run_human_segmentation(mode=args.mode, supercategories=args.supercategories, categories=args.categories, prompts=args.prompts,
                           inpaint_dir=args.inpaint_dir, save_dir=args.save_dir, threshold=args.threshold, parallel_num=args.parallel_num,
                           parallel_idx=args.parallel_idx, save_full=not args.disable_save_full,
                           save_vis_in_same_folder=args.save_vis_in_same_folder, save_image=args.save_image, skip_done=args.skip_done,
                           verbose=args.verbose)
