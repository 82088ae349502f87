"""CPU: pure-Python glue of the inpainting path against golden vectors captured from the reference (G13-G16),
the DDIM schedule constants, and the per-GPU work-list slicing."""
import os

import numpy as np
import PIL.Image
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gi():
    return np.load(os.path.join(ROOT, "tests", "golden", "inpaint_golden.npz"))


def test_schedules_match_reference(gi):
    from utils.adaptive_mask_inpainting import MaskDilateScheduler, ProvokeScheduler, default_adaptive_mask_settings
    s = default_adaptive_mask_settings(50, "p")
    assert [s.dilate_scheduler(i) for i in range(50)] == list(gi["g13_dilate"])
    assert [i for i in range(50) if s.provoke_scheduler(i)] == list(gi["g14_provoke"])
    assert len(gi["g14_provoke"]) == 21
    d = MaskDilateScheduler(max_dilate_num=15, num_inference_steps=50)
    assert [d(i) for i in range(50)] == list(gi["g13_dilate_default"])
    pz = ProvokeScheduler(num_inference_steps=50, schedule=[0, 3, 49], is_zero_indexing=True)
    assert [i for i in range(50) if pz(i)] == list(gi["g14_provoke_zero"])
    assert s.dilate_kernel.shape == (3, 3)
    base = default_adaptive_mask_settings(50, "baseline")
    assert not any(base.provoke_scheduler(i) for i in range(50))


def test_bbox_helpers(gi):
    from utils.adaptive_mask_inpainting import merge_bbox, seg2bbox
    boxes = [seg2bbox(s) for s in gi["g15_segs"]]
    assert np.array_equal(np.stack(boxes), gi["g15_boxes"])
    assert np.array_equal(merge_bbox(boxes), gi["g15_merged"])


def test_prepare_mask_and_masked_image(gi):
    from utils.adaptive_mask_inpainting import prepare_mask_and_masked_image as prep
    img, m = gi["g16_img_u8"], gi["g16_mask_bool"]
    mk, ms, im = prep(PIL.Image.fromarray(img), PIL.Image.fromarray((m * 255).astype(np.uint8)), 24, 32, return_image=True)
    assert mk.dtype == torch.float32 and np.array_equal(mk.numpy(), gi["g16_pil_mask"])
    assert np.array_equal(ms.numpy(), gi["g16_pil_masked"]) and np.array_equal(im.numpy(), gi["g16_pil_image"])
    mk, ms = prep(img, m.astype(np.float32) * 0.7 + 0.1, 24, 32)
    assert np.array_equal(mk.numpy(), gi["g16_np_mask"]) and np.array_equal(ms.numpy(), gi["g16_np_masked"])
    ti = torch.tensor(img.transpose(2, 0, 1)[None].astype(np.float32) / 127.5 - 1.0)
    tm = torch.tensor(m.astype(np.float32))[None, None]
    mk, ms = prep(ti.clone(), tm.clone(), 24, 32)
    assert np.array_equal(mk.numpy(), gi["g16_pt_mask"]) and np.array_equal(ms.numpy(), gi["g16_pt_masked"])
    errs = []
    for bad in (lambda: prep(ti * 2, tm, 24, 32), lambda: prep(ti, tm * 2, 24, 32), lambda: prep(ti, m, 24, 32),
                lambda: prep(None, tm, 24, 32)):
        try:
            bad()
            errs.append("none")
        except Exception as e:   # noqa: BLE001
            errs.append(type(e).__name__)
    assert errs == list(gi["g16_errors"])


def test_ddim_constants_and_timesteps():
    from oracle import sd_oracle as so
    from utils.adaptive_mask_inpainting import DDIMScheduler
    s = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False,
                      steps_offset=1)
    s.set_timesteps(50)
    ts = s.timesteps.tolist()
    assert ts == so.ddim_timesteps(50) and ts[0] == 981 and ts[-1] == 1 and len(ts) == 50
    a = so.ddim_alphas()
    assert float((s.alphas_cumprod.double() - a).abs().max()) < 1e-6
    a_t, a_p = s.alphas_for(1)
    assert a_p == float(s.alphas_cumprod[0])                     # set_alpha_to_one=False -> final alpha = alpha[0]
    a_t, a_p = s.alphas_for(961)
    assert a_p == float(s.alphas_cumprod[941])


def test_strength_drops_the_first_step():
    """strength 0.98 with 50 steps executes 49 steps starting at t=961 (utils/adaptive_mask_inpainting.py:722-729)."""
    from coma_amd.sd.pipeline import AdaptiveMaskInpaintPipeline
    from coma_amd.sd.scheduler import DDIMScheduler

    class P(AdaptiveMaskInpaintPipeline):
        def __init__(self):
            self.scheduler = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                                           set_alpha_to_one=False, steps_offset=1)
    p = P()
    p.scheduler.set_timesteps(50)
    ts, n = p.get_timesteps(50, 0.98)
    assert n == 49 and int(ts[0]) == 961 and len(ts) == 49
    ts, n = p.get_timesteps(50, 1.0)
    assert n == 50 and int(ts[0]) == 981


def test_work_list_enumeration_sorting_overrides_and_slicing(tmp_path):
    """Hand-computable case of the reference's work-list rules (src/generation/inpaint.py:187-278)."""
    import pickle
    from PIL import Image
    from src.generation import inpaint as gi
    base = tmp_path
    sc, c, asset = "BEHAVE", "backpack", "behave_asset"
    for v in (0, 1):
        d = base / "renders" / sc / c / asset
        d.mkdir(parents=True, exist_ok=True)
        Image.new("RGB", (8, 8)).save(d / f"view:{v:05}.png")
        m = base / "masks" / sc / c / asset
        m.mkdir(parents=True, exist_ok=True)
        pickle.dump({"valid_mask_ids": ["00000", "00003"] if v == 0 else ["00001"]}, open(m / f"view:{v:05}.pickle", "wb"))
    (base / "renders" / sc / c / "unregistered_asset").mkdir()
    Image.new("RGB", (8, 8)).save(base / "renders" / sc / c / "unregistered_asset" / "view:00000.png")
    pd = base / "prompts" / sc / c / asset
    pd.mkdir(parents=True)
    pickle.dump({"prompts": ["1 person wears the backpack", "1 person holds the backpack"], "use_vlm": False}, open(pd / "prompts.pickle", "wb"))
    renders = gi.prepare_asset_render_pths(str(base / "renders"), None, ["backpack"])
    assert len(renders) == 2 and all("unregistered" not in r for r in renders)
    defaults = dict(ddim_steps=50, cfg_scale=11.0, strength=0.5, enforce_full_mask_ratio=0.0, human_detection_thres=0.015)
    items = gi.build_work_list(renders, str(base / "masks"), str(base / "segs"), str(base / "prompts"), str(base / "out"), 10,
                               "neg", defaults)
    # (2 + 1 masks) x 2 prompts x 2 augmentations x 10 seeds
    assert len(items) == 3 * 2 * 2 * 10
    assert [i["result_save_pth"] for i in items] == sorted(i["result_save_pth"] for i in items)
    assert all(i["strength"] == 0.98 for i in items)              # category override beats the CLI default
    assert all(i["cfg_scale"] == 11.0 and i["ddim_steps"] == 50 for i in items)
    assert {i["input_prompt"] for i in items} == {"1 person wears the backpack", "1 person wears the backpack, full body",
                                                  "1 person holds the backpack", "1 person holds the backpack, full body"}
    assert items[0]["result_save_pth"].endswith("/000000.png") and items[0]["inpaint_id"] == 0
    sizes = [len(gi.slice_for_process(items, r, 8)) for r in range(8)]
    assert sizes == [16] * 7 + [8] and sum(sizes) == 120          # sub = 120 // 8 + 1 = 16
    assert gi.resolve_setting("motorcycle,bike", "motorcycle,bike", "view:00099", "strength", 0.5) == 0.9


def test_inpaint_cli_flags_match_reference():
    from src.generation.inpaint import build_parser
    flags = {a.option_strings[0] for a in build_parser()._actions if a.option_strings}
    for f in ("--num_img_per_combination", "--supercategories", "--categories", "--asset_render_dir", "--asset_mask_dir",
              "--asset_seg_dir", "--prompts_dir", "--save_dir", "--ldm_model_key", "--adaptive_mask_model_type", "--default_cfg_scale",
              "--default_strength", "--default_ddim_steps", "--default_pointrend_threshold", "--default_enforce_full_mask_ratio",
              "--default_human_detection_thres", "--enable_sam_multitask_output", "--negative_prompt", "--enable_safety_checker",
              "--use_visualizer", "--skip_done", "--verbose", "--seed", "--parallel_num", "--parallel_idx"):
        assert f in flags, f
    d = build_parser().parse_args([])
    assert (d.default_cfg_scale, d.default_strength, d.default_ddim_steps, d.num_img_per_combination) == (11.0, 0.98, 50, 10)


def test_checkpoint_layout_check():
    """A checkpoint of another family is refused with a readable message before any launch (weights.check_state)."""
    import torch
    from coma_amd.sd import weights
    shapes = weights.unet_shapes()
    meta = {k: torch.empty(v, device="meta") for k, v in shapes.items()}
    assert set(weights.check_state(dict(meta, extra=torch.empty(1, device="meta")), shapes)) == set(shapes)
    t2i = dict(meta)
    t2i["conv_in.weight"] = torch.empty(320, 4, 3, 3, device="meta")          # text-to-image UNet: 4 input channels
    del t2i["mid_block.resnets.0.conv1.bias"]
    with pytest.raises(ValueError, match="1 missing.*conv1.bias.*1 with other shapes.*conv_in.weight"):
        weights.check_state(t2i, shapes, "UNet")


def test_group_for_batches_keeps_order_and_settings():
    """src/generation/inpaint.py: the rank's slice is cut into groups of <= batch_size CONSECUTIVE items with equal per-call settings."""
    from src.generation import inpaint as gi
    base = dict(ddim_steps=50, cfg_scale=11.0, strength=0.98, enforce_full_mask_ratio=0.0, human_detection_thres=0.015)
    items = [dict(base, result_save_pth=f"a/{i:06}.png", inpaint_id=i) for i in range(11)]
    items += [dict(base, strength=0.9, result_save_pth=f"b/{i:06}.png", inpaint_id=i) for i in range(3)]
    items += [dict(base, result_save_pth=f"c/{i:06}.png", inpaint_id=i) for i in range(2)]
    groups = gi.group_for_batches(items, 8)
    assert [len(g) for g in groups] == [8, 3, 3, 2]
    assert [it["result_save_pth"] for g in groups for it in g] == [it["result_save_pth"] for it in items]       # order kept
    for g in groups:
        assert all(all(it[k] == g[0][k] for k in gi.SETTING_KEYS) for it in g)
    assert [len(g) for g in gi.group_for_batches(items, 1)] == [1] * 16
    assert gi.group_for_batches([], 8) == []
    assert gi.build_parser().parse_args([]).batch_size == gi.DEFAULT_BATCH_SIZE == 8
    # an item without its own plug-in state (no segmentation file) always OPENS a group: it inherits the state the item before it left
    # AFTER its pipeline call, as in the reference's sequential loop (src/generation/inpaint.py:325 of the reference)
    unprimed = {3, 4, 10}
    groups = gi.group_for_batches(items[:11], 8, primed=lambda it: it["inpaint_id"] not in unprimed)
    assert [[it["inpaint_id"] for it in g] for g in groups] == [[0, 1, 2], [3], [4, 5, 6, 7, 8, 9], [10]]


def test_per_item_state_swaps_only_the_small_state():
    """coma_amd/sd/predictors.PerItemState: one predictor serves B batch slots, every slot with its own box / asset-mask state."""
    from coma_amd.sd.predictors import PerItemState

    class Net:           # stands for the shared segmentation network
        pass

    class Pred:
        use_visualizer = False

        def __init__(self):
            self.net, self.initial_human_bbox, self.calls = Net(), None, 0

        def set_presumed_asset_mask(self, m):
            self.presumed_asset_mask, self.presumed_asset_bbox = m, ("bbox", int(m.sum()))

        def __call__(self, image):
            self.calls += 1
            box = np.array([image, image, image + 1, image + 1])
            self.initial_human_bbox = box if self.initial_human_bbox is None else np.minimum(self.initial_human_bbox, box)
            return {"mask": self.initial_human_bbox.copy(), "asset": getattr(self, "presumed_asset_bbox", None)}

    p = Pred()
    w = PerItemState(p, 3)
    for b in range(3):
        w.select(b)
        if b != 1:
            w.set_presumed_asset_mask(np.ones((b + 2, b + 2)))
    outs = []
    for step in range(2):
        for b in range(3):
            w.select(b)
            outs.append(w(10 * (b + 1) - step))
    assert p.calls == 6 and w.net is p.net and w.use_visualizer is False
    # slot b saw images 10(b+1) and 10(b+1) - 1: its running box is its own, not the other slots'
    assert [int(o["mask"][0]) for o in outs] == [10, 20, 30, 9, 19, 29]
    assert [o["asset"] for o in outs[:3]] == [("bbox", 4), None, ("bbox", 16)]
    w.select(1)
    assert not hasattr(p, "presumed_asset_mask")
    w.select(2)
    assert p.presumed_asset_mask.shape == (4, 4)


def test_per_item_state_inherits_from_the_previous_item():
    """an item that is not primed (no segmentation file) starts from the state of the item before it in list order --
    what the reference's one-item-per-call loop leaves in the plug-in -- not from the item that used its slot one group earlier."""
    from coma_amd.sd.predictors import PerItemState

    class Pred:
        def __init__(self):
            self.initial_human_bbox = None

    p = Pred()
    w = PerItemState(p, 3)
    for b in range(3):                               # group 1: every slot primed with its own box
        w.select(b, inherit_from=(b - 1) % 3)
        w.initial_human_bbox = ("box", b)
    # group 2: slot 0 inherits from slot 2 (the last item of group 1), slot 1 is primed, slot 2 inherits from slot 1
    w.select(0, inherit_from=2)
    assert p.initial_human_bbox == ("box", 2)
    w.select(1, inherit_from=0)
    w.initial_human_bbox = ("box", "new")
    w.select(2, inherit_from=1)
    assert p.initial_human_bbox == ("box", "new")
    w.select(0)
    assert p.initial_human_bbox == ("box", 2)        # slot 0 kept what it inherited
    w.select(1)
    assert p.initial_human_bbox == ("box", "new")
    w.select(1, inherit_from=0)                      # the live slot can be overwritten too
    assert p.initial_human_bbox == ("box", 2)


def test_synthetic_plug_in_batched_form_equals_the_per_image_form():
    """SyntheticHumanMaskPredictor.predict_batch (one plug-in call per mask re-estimation of a batched pipeline call) returns, image by image,
    the bits of __call__ on a tensor and of __call__ on the reference-contract NumPy array."""
    from coma_amd.sd.pipeline import SyntheticHumanMaskPredictor
    p = SyntheticHumanMaskPredictor()
    g = torch.Generator().manual_seed(3)
    imgs = (torch.rand(5, 64, 48, 3, generator=g) * 255).to(torch.uint8)
    imgs[2] = 255
    imgs[3] = 0
    batch = p.predict_batch(imgs)["mask"]
    assert batch.dtype == torch.uint8 and tuple(batch.shape) == (5, 64, 48)
    for b in range(5):
        one_t = p(imgs[b])["mask"]
        one_n = p(imgs[b].numpy())["mask"]
        assert torch.equal(batch[b], one_t) and np.array_equal(batch[b].numpy(), one_n)
    assert 0 < int(batch[0].sum()) < 64 * 48


def _g20(gi):
    unpack = lambda a: np.unpackbits(a, axis=-1).astype(bool)
    return unpack(gi["g20_segs"]), unpack(gi["g20_default"]), [int(k) for k in gi["g20_ks"]], float(gi["g20_thres"]), unpack(gi["g20_dilated"]), \
        unpack(gi["g20_adapted"])


def test_mask_dilation_restatement_matches_independent_implementation(gi):
    """G20: the oracle's restatement of `cv2.dilate(mask, ones((3,3)), iterations=k)` AND default mask / area fallback
    (utils/adaptive_mask_inpainting.py:1123-1157) against masks produced by scipy.ndimage.binary_dilation (cross-checked with a plain NumPy
    shifted-OR by the generator): blobs on every border and corner, salt noise, the area-below-threshold branch, k in {0, 1, 5, 20} at
    512 x 512.  Checked against scipy, still not against cv2 (absent)."""
    from oracle import sd_oracle as so
    segs, default, ks, thres, dilated, adapted = _g20(gi)
    assert segs.shape == (4, 512, 512) and segs[2].sum() < 512 * 512 * thres <= segs[3].sum()       # both branches of :1132 are present
    for s in range(len(segs)):
        for j, k in enumerate(ks):
            assert np.array_equal(so.dilate_ref(segs[s].astype(np.uint8), k).astype(bool), dilated[s, j]), (s, k)
            got = so.adapt_mask_ref(segs[s].astype(np.uint8), default.astype(np.uint8), k, False, thres)
            assert np.array_equal(got.astype(bool), adapted[s, j]), (s, k)
    assert np.array_equal(adapted[2, 3], default) and not np.array_equal(adapted[0, 3], default)
