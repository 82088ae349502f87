"""Inpaint a human into every rendered view with the adaptive-mask SD loop on MI355X.

CLI surface and host behaviour of the reference's ``src/generation/inpaint.py``:
  * flags and module-level defaults (:356-411) -- the shell wrapper reads the constants by importing this module;
  * work list = render x valid mask x prompt x viewpoint augmentation x inpaint_id (:187-269), per-category /
    per-view override chain for ddim_steps / cfg_scale / strength / enforce_full_mask_ratio / human_detection_thres
    (:253-267), sorted by result path, per-process slice ``sub = len // n + 1`` (:271-278);
  * per item: device generator seeded with ``inpaint_id`` (:308-309), skip-if-exists (:294-297), PNG output (:352);
  * the rank's slice is walked in groups of ``--batch_size`` (default 8) consecutive items with equal per-call settings; one
    pipeline call per group, one generator / prompt / mask / plug-in state per image (the reference's loop runs one item per
    call, :280-352).  Every item carries its own seed, so an item's noise draws do not depend on how the list is cut; its image is
    equal to the one-item-per-call image up to fp16 rounding only (kernel selection -- tile shapes, split-K, Winograd from UNet batch
    8 -- depends on the batch: measured 0.19 grey levels mean, tests/test_inpaint_cli_gpu.py), i.e. statistically equivalent, not
    bit-identical; ``--batch_size 1`` is the reference's own call shape.
The pipeline is :class:`coma_amd.sd.pipeline.AdaptiveMaskInpaintPipeline` (HIP kernels).  Model weights, the CLIP text
encoder and PointRend are third-party assets that cannot be provisioned offline: ``--weights_dir`` points at a
diffusers-format checkpoint directory when one exists (its tokenizer/ + text_encoder/ then embed the prompts); otherwise
seeded random weights are used (pipeline smoke / throughput runs) and prompts are embedded by a deterministic hash
encoder.  ``--mask_model auto`` (default) builds the PointRend / SAM plug-in of ``--adaptive_mask_model_type``
(coma_amd/sd/predictors.py; needs detectron2 / segment-anything), ``--mask_model synthetic`` the dependency-free stand-in.
"""
import argparse
import hashlib
import os
import pickle
import sys
from glob import glob


ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from constants.generation.inpaint_config import (ALLOWED_VIEWPOINT_AUGMENTATIONS, CATEGORY2ASSET, HF_MODEL_KEYS,  # noqa: E402
                                                 SC2DIFFUSERCONFIG, SCV2DIFFUSERCONFIG)
from constants.metadata import DEFAULT_SEED  # noqa: E402

""" HYPERPARAMS (read by scripts/generation/inpaint.sh through `python -c`) """
NUM_IMG_PER_COMBINATION = 10
ASSET_RENDER_DIR = "results/generation/asset_renders"
ASSET_MASK_DIR = "results/generation/asset_masks"
ASSET_SEG_DIR = "results/generation/asset_segs"
PROMPTS_DIR = "results/generation/prompts"
SAVE_DIR = "results/generation/inpaintings"
LDM_MODEL_KEY = "realisticvision"
ADAPTIVE_MASK_MODEL_TYPE = "p"
DEFAULT_CFG_SCALE = 11.0
DEFAULT_STRENGTH = 0.98
DEFAULT_DDIM_STEPS = 50
DEFAULT_POINTREND_THRESHOLD = 0.2
DEFAULT_ENFORCE_FULL_MASK_RATIO = 0.0
DEFAULT_HUMAN_DETECTION_THRES = 0.015
NEGATIVE_PROMPT = "worst quality, normal quality, low quality, bad anatomy, artifacts, blurry, cropped, watermark, greyscale, nsfw"
SKIP_DONE = True
DEFAULT_BATCH_SIZE = 8          # addition: images per pipeline call (the reference runs 1; per-item seeding makes batching output-preserving)


def prepare_asset_render_pths(asset_render_dir, supercategories, categories):
    """Render paths {dir}/{SC}/{C}/{asset}/{view}.png of registered assets, optionally filtered (utils/prepare_renders.py:6-32)."""
    out = []
    for pth in sorted(glob(f"{asset_render_dir}/*/*/*/*.png")):
        sc_str, c_str, asset_id, _ = pth.split("/")[-4:]
        sc, c = sc_str.replace(":", "/"), c_str.replace(":", "/")
        if asset_id not in CATEGORY2ASSET.get(sc, {}).get(c, []):
            continue
        if supercategories is not None and sc_str.lower() not in supercategories:
            continue
        if categories is not None and c_str.lower() not in categories:
            continue
        out.append(pth)
    return out


def resolve_setting(supercategory, category, view_id, key, default):
    """view override -> category override -> CLI default."""
    cat_cfg = SC2DIFFUSERCONFIG[supercategory][category]
    view_cfg = SCV2DIFFUSERCONFIG[supercategory][category].get(view_id, cat_cfg)
    return view_cfg.get(key, cat_cfg.get(key, default))


def build_work_list(asset_render_pths, asset_mask_dir, asset_seg_dir, prompts_dir, save_dir, num_img_per_combination, negative_prompt,
                    defaults, use_visualizer=False, debug=False):
    items = []
    for render in asset_render_pths:
        sc_str, c_str, asset_id, view_ext = render.split("/")[-4:]
        sc, c = sc_str.replace(":", "/"), c_str.replace(":", "/")
        view_id, ext = view_ext.split(".")
        assert ext == "png", "Rendering must have '.png' extension"
        meta = f"{asset_mask_dir}/{sc_str}/{c_str}/{asset_id}/{view_id}.pickle"
        if os.path.exists(meta):
            with open(meta, "rb") as fh:
                mask_ids = pickle.load(fh)["valid_mask_ids"]
        else:
            assert debug, "THIS SHOULD BE ONLY ALLOWED IN DEBUGGING MODE. RUN STEP2 PRIOR"
            mask_ids = [p.split("/")[-1].split(".")[0] for p in sorted(glob(f"{asset_mask_dir}/{sc_str}/{c_str}/{asset_id}/{view_id}/*"))]
        with open(f"{prompts_dir}/{sc}/{c}/{asset_id}/prompts.pickle", "rb") as fh:
            prompts = pickle.load(fh)["prompts"]
        augs = SCV2DIFFUSERCONFIG[sc][c].get(view_id, SC2DIFFUSERCONFIG[sc][c]).get("view_text", ["original"])
        assert type(augs) == list
        for mask_id in mask_ids:
            for prompt in prompts:
                for aug in augs:
                    assert aug in ALLOWED_VIEWPOINT_AUGMENTATIONS, f"viewpoint augmentation: '{aug}' not allowed"
                    if aug == "original":
                        text = prompt
                    elif aug == ", full body":
                        text = prompt + aug
                    else:
                        continue
                    rdir = f"{save_dir}/{sc_str}/{c_str}/{asset_id}/{view_id}/{mask_id}/{text}"
                    for inpaint_id in range(num_img_per_combination):
                        item = dict(asset_render_pth=render, asset_mask_pth=f"{asset_mask_dir}/{sc_str}/{c_str}/{asset_id}/{view_id}/{mask_id}.png",
                                    asset_seg_pth=f"{asset_seg_dir}/{sc_str}/{c_str}/{asset_id}/{view_id}.png", result_save_dir=rdir,
                                    result_save_pth=f"{rdir}/{inpaint_id:06}.png",
                                    visualization_save_dir=f"{rdir}/{inpaint_id:06}" if use_visualizer else None,
                                    inpaint_id=inpaint_id, input_prompt=text, input_negprompt=negative_prompt)
                        for key, default in defaults.items():
                            item[key] = resolve_setting(sc, c, view_id, key, default)
                        items.append(item)
    return sorted(items, key=lambda x: x["result_save_pth"])


def slice_for_process(items, parallel_idx, parallel_num):
    sub = len(items) // parallel_num + 1          # the reference's (unbalanced) rule
    return items[parallel_idx * sub:(parallel_idx + 1) * sub]


class HashTextEncoder:
    """Deterministic stand-in for CLIP's text tower (weights unreachable offline): prompt -> [1,77,768] unit-variance noise."""

    def __call__(self, prompt):
        import torch
        seed = int.from_bytes(hashlib.sha256(prompt.encode()).digest()[:8], "little") % (2**63)
        return torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(seed))


def set_pipeline(ldm_model_key, adaptive_mask_model_type, default_ddim_steps, weights_dir=None, mask_model="auto", device="cuda",
                 default_pointrend_threshold=DEFAULT_POINTREND_THRESHOLD, use_visualizer=False, enable_sam_multitask_output=False,
                 batch_size=1):
    """src/generation/inpaint.py:45-134 of the reference: pipeline + the mask plug-in picked by `adaptive_mask_model_type`
    (PointRend / SAM family, coma_amd.sd.predictors) + the dilate / provoke schedules.  `mask_model="synthetic"` swaps in the
    dependency-free stand-in; "auto" needs detectron2 (and segment-anything for the SAM types) and says so if they are missing."""
    from coma_amd.sd.pipeline import AdaptiveMaskInpaintPipeline, SyntheticHumanMaskPredictor, default_adaptive_mask_settings
    from coma_amd.sd.predictors import build_adaptive_mask_model
    assert ldm_model_key in HF_MODEL_KEYS
    if weights_dir:
        pipeline = AdaptiveMaskInpaintPipeline.from_pretrained(weights_dir, batch_size=batch_size, device=device)
        if pipeline.text_encoder is None:
            raise FileNotFoundError(f"{weights_dir} has no text_encoder/ + tokenizer/: with a real checkpoint the prompts must go "
                                    "through its CLIP text tower (the hash encoder is only for the random-weight path)")
    else:
        pipeline = AdaptiveMaskInpaintPipeline.from_random(batch_size=batch_size, device=device)
    pipeline.scheduler.set_timesteps(default_ddim_steps)
    if mask_model == "synthetic":
        pipeline.register_adaptive_mask_model(SyntheticHumanMaskPredictor())
    elif mask_model == "auto":
        try:
            model = build_adaptive_mask_model(adaptive_mask_model_type, default_pointrend_threshold, use_visualizer,
                                              enable_sam_multitask_output, device=device)
        except ImportError as e:
            if weights_dir:                  # a real checkpoint with a stand-in segmenter would silently produce wrong masks
                raise
            import warnings
            warnings.warn(f"\n{'#' * 100}\n--mask_model auto: {e}\nNo --weights_dir either (seeded random weights): falling back to the "
                          f"dependency-free SyntheticHumanMaskPredictor, i.e. `--mask_model synthetic`.\n{'#' * 100}", RuntimeWarning)
            model = SyntheticHumanMaskPredictor()
        pipeline.register_adaptive_mask_model(model)
    else:
        raise ValueError(f"--mask_model {mask_model!r}: expected 'auto' or 'synthetic'")
    pipeline.register_adaptive_mask_settings(default_adaptive_mask_settings(default_ddim_steps, adaptive_mask_model_type))
    return pipeline


def prime_mask_model(model, adaptive_mask_model_type, asset_seg, default_mask):
    """Per-item state of the SAM plug-ins (reference :325-337): presumed asset mask, reset / preset person box."""
    if adaptive_mask_model_type in ("ps_ae", "s_db_ae", "s_pdb_ae", "s_ab_ae") and hasattr(model, "set_presumed_asset_mask"):
        model.set_presumed_asset_mask(asset_seg)
        if adaptive_mask_model_type != "ps_ae":
            model.reset_initial_human_bbox()
        if adaptive_mask_model_type == "s_db_ae":
            model.set_initial_human_bbox(default_mask)


SETTING_KEYS = ("ddim_steps", "cfg_scale", "strength", "enforce_full_mask_ratio", "human_detection_thres")


def group_for_batches(items, batch_size, primed=None):
    """Walk the (sorted, sliced) work list in order and cut it into groups of at most `batch_size` CONSECUTIVE items that share
    every per-call setting of the override chain (ddim_steps / cfg_scale / strength / enforce_full_mask_ratio /
    human_detection_thres: one pipeline call has one value of each).  The reference runs one item per call
    (src/generation/inpaint.py:280-352); every item keeps its own generator seeded with its `inpaint_id` (:308-309), its own
    prompt, mask and plug-in state, so the images do not depend on how the list is cut.

    `primed(it)` says whether the item brings its own plug-in state (a segmentation file, :325).  An item that does NOT starts, in the
    reference's sequential loop, from what the item before it left in the plug-in AFTER that item's pipeline call (e.g. the accumulated
    `initial_human_bbox`).  Inside a batch the previous slot has not run yet, so such an item always OPENS a group: slot 0 inherits the
    previous group's last slot after its call -- the sequential loop's state exactly, at the price of a shorter group."""
    groups, cur = [], []
    for it in items:
        if cur and (len(cur) == batch_size or any(it[k] != cur[0][k] for k in SETTING_KEYS) or (primed is not None and not primed(it))):
            groups.append(cur)
            cur = []
        cur.append(it)
    if cur:
        groups.append(cur)
    return groups


def inpaint_human(args):
    import numpy as np
    import torch
    from PIL import Image
    from coma_amd.sd.predictors import PerItemState
    renders = prepare_asset_render_pths(args.asset_render_dir, args.supercategories, args.categories)
    defaults = dict(ddim_steps=args.default_ddim_steps, cfg_scale=args.default_cfg_scale, strength=args.default_strength,
                    enforce_full_mask_ratio=args.default_enforce_full_mask_ratio, human_detection_thres=args.default_human_detection_thres)
    items = build_work_list(renders, args.asset_mask_dir, args.asset_seg_dir, args.prompts_dir, args.save_dir, args.num_img_per_combination,
                            args.negative_prompt, defaults, use_visualizer=args.use_visualizer)
    items = slice_for_process(items, args.parallel_idx, args.parallel_num)
    todo = []
    for it in items:                       # skip-if-exists first (:294-297), so that finished items do not occupy batch slots
        os.makedirs(it["result_save_dir"], exist_ok=True)
        if os.path.exists(it["result_save_pth"]) and args.skip_done:
            if args.verbose:
                print(f"Continueing {it['result_save_pth']} Since Already Done!")
            continue
        todo.append(it)
    if not todo:
        return
    B = max(1, min(int(getattr(args, "batch_size", 1)), len(todo)))      # never a batch wider than the work that is left
    pipeline = set_pipeline(args.ldm_model_key, args.adaptive_mask_model_type, args.default_ddim_steps, args.weights_dir, args.mask_model,
                            default_pointrend_threshold=args.default_pointrend_threshold, use_visualizer=args.use_visualizer,
                            enable_sam_multitask_output=args.enable_sam_multitask_output, batch_size=B)
    # a real checkpoint brings its CLIP text tower (prompts are tokenised and encoded as in the reference); only the
    # random-weight path embeds prompts with the deterministic hash encoder
    encode = None if pipeline.text_encoder is not None else HashTextEncoder()
    adaptive = args.adaptive_mask_model_type != "baseline"
    if adaptive and B > 1:
        # one plug-in instance serves B images of a batch: its small per-item state (presumed asset mask, person boxes) is swapped
        # in and out around every call, the networks are shared
        pipeline.register_adaptive_mask_model(PerItemState(pipeline.adaptive_mask_model, B))
    prev_last = None                                   # slot of the last real item of the previous group
    for group in group_for_batches(todo, B, primed=(lambda it: os.path.exists(it["asset_seg_pth"])) if adaptive else None):
        n = len(group)
        slots = group + [group[-1]] * (B - n)          # ragged tail: the last item fills the spare slots, its copies are dropped
        images = [Image.open(it["asset_render_pth"]).convert("RGB") for it in slots]
        masks = [Image.open(it["asset_mask_pth"]).convert("L") for it in slots]
        for b, it in enumerate(slots):
            model = pipeline.adaptive_mask_model
            if isinstance(model, PerItemState):
                # an item without a segmentation file is not primed (reference :325): it starts from what the item BEFORE it in list
                # order left in the plug-in after its call.  group_for_batches() makes such an item slot 0 of its group, so the state it
                # copies is the previous group's LAST REAL slot after that group's pipeline call (not the unrelated item that used this
                # slot one group earlier); slots b > 0 are always primed below and the copy they receive here is overwritten
                model.select(b, inherit_from=(prev_last if b == 0 else b - 1) if B > 1 else None)
            if os.path.exists(it["asset_seg_pth"]):
                prime_mask_model(model, args.adaptive_mask_model_type, np.array(Image.open(it["asset_seg_pth"]).convert("L")) > 0,
                                 np.asarray(masks[b]) > 0)
        generators = []
        for it in slots:                                # one generator per image, seeded with its inpaint_id (:308-309)
            g = torch.Generator(device="cuda")
            g.manual_seed(it["inpaint_id"])
            generators.append(g)
        if encode is None:
            text = dict(prompt=[it["input_prompt"] for it in slots], negative_prompt=[it["input_negprompt"] for it in slots])
        else:
            text = dict(prompt_embeds=torch.cat([encode(it["input_prompt"]) for it in slots]),
                        negative_prompt_embeds=torch.cat([encode(it["input_negprompt"]) for it in slots]))
        it0 = group[0]
        results = pipeline(image=images if B > 1 else images[0], default_mask_image=masks if B > 1 else masks[0],
                           guidance_scale=it0["cfg_scale"], strength=it0["strength"], use_adaptive_mask=adaptive,
                           generator=generators if B > 1 else generators[0], num_inference_steps=it0["ddim_steps"],
                           enforce_full_mask_ratio=it0["enforce_full_mask_ratio"], visualization_save_dir=it0["visualization_save_dir"],
                           human_detection_thres=it0["human_detection_thres"], **text).images
        prev_last = n - 1
        for it, img in zip(group, results[:n]):
            img.save(it["result_save_pth"])


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--num_img_per_combination", type=int, default=NUM_IMG_PER_COMBINATION)
    p.add_argument("--supercategories", type=str, nargs="+")
    p.add_argument("--categories", type=str, nargs="+")
    p.add_argument("--asset_render_dir", type=str, default=ASSET_RENDER_DIR)
    p.add_argument("--asset_mask_dir", type=str, default=ASSET_MASK_DIR)
    p.add_argument("--asset_seg_dir", type=str, default=ASSET_SEG_DIR)
    p.add_argument("--prompts_dir", type=str, default=PROMPTS_DIR)
    p.add_argument("--save_dir", type=str, default=SAVE_DIR)
    p.add_argument("--ldm_model_key", type=str, default=LDM_MODEL_KEY, choices=HF_MODEL_KEYS.keys())
    p.add_argument("--adaptive_mask_model_type", type=str, choices=["baseline", "p", "ps", "ps_ae", "s_pdb_ae", "s_db_ae", "s_ab_ae"],
                   default=ADAPTIVE_MASK_MODEL_TYPE)
    p.add_argument("--default_cfg_scale", type=float, default=DEFAULT_CFG_SCALE)
    p.add_argument("--default_strength", type=float, default=DEFAULT_STRENGTH)
    p.add_argument("--default_ddim_steps", type=int, default=DEFAULT_DDIM_STEPS)
    p.add_argument("--default_pointrend_threshold", type=float, default=DEFAULT_POINTREND_THRESHOLD)
    p.add_argument("--default_enforce_full_mask_ratio", type=float, default=DEFAULT_ENFORCE_FULL_MASK_RATIO)
    p.add_argument("--default_human_detection_thres", type=float, default=DEFAULT_HUMAN_DETECTION_THRES)
    p.add_argument("--enable_sam_multitask_output", action="store_true")
    p.add_argument("--negative_prompt", type=str, default=NEGATIVE_PROMPT)
    p.add_argument("--enable_safety_checker", action="store_true")
    p.add_argument("--use_visualizer", action="store_true")
    p.add_argument("--skip_done", action="store_true", default=SKIP_DONE)
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--seed", type=int, default=DEFAULT_SEED)
    p.add_argument("--parallel_num", type=int, default=1)
    p.add_argument("--parallel_idx", type=int, default=0)
    # additions (not in the reference): offline asset provisioning
    p.add_argument("--weights_dir", type=str, default=None, help="diffusers-format checkpoint dir; default: seeded random weights")
    p.add_argument("--batch_size", type=int, default=DEFAULT_BATCH_SIZE,
                   help="images per pipeline call (the UNet runs 2 x batch_size rows per step); the work list is walked in groups, every "
                        "image keeps its own generator / prompt / mask, so the outputs do not depend on it")
    p.add_argument("--mask_model", type=str, default="auto", choices=["auto", "synthetic"],
                   help="auto: the PointRend / SAM plug-in selected by --adaptive_mask_model_type (needs detectron2 / segment-anything); "
                        "synthetic: dependency-free stand-in")
    return p


if __name__ == "__main__":
    args = build_parser().parse_args()
    if args.supercategories is not None:
        args.supercategories = [s.lower() for s in args.supercategories]
    if args.categories is not None:
        args.categories = [c.lower() for c in args.categories]
    if args.adaptive_mask_model_type == "baseline":
        print("\n\n############################# RUNNING BASELINE MODE!! #############################\n\n")
        args.save_dir = f"{args.save_dir}_noadaptivemask"
    from utils.reproducibility import seed_everything
    seed_everything(args.seed)
    inpaint_human(args)
