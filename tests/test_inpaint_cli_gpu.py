"""GPU: the inpainting CLI (src/generation/inpaint.py) on a synthetic asset tree -- the batched walk of the work list
(--batch_size 4: one full group + a ragged tail) writes the same files as the reference-shaped one-item-per-call run and the same
images up to fp16 summation order (every item keeps its own generator seeded with its inpaint_id, reference :308-309)."""
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tree(base):
    from PIL import Image
    sc, c, asset, view = "BEHAVE", "backpack", "behave_asset", "view:00000"
    rng = np.random.default_rng(0)
    d = base / "renders" / sc / c / asset
    d.mkdir(parents=True)
    img = np.clip(rng.normal(128, 40, size=(64, 64, 3)), 0, 255).astype(np.uint8).repeat(8, 0).repeat(8, 1)
    Image.fromarray(img).save(d / f"{view}.png")
    m = base / "masks" / sc / c / asset
    (m / view).mkdir(parents=True)
    pickle.dump({"valid_mask_ids": ["00000"]}, open(m / f"{view}.pickle", "wb"))
    mask = np.zeros((512, 512), np.uint8)
    mask[96:448, 160:384] = 255
    Image.fromarray(mask).save(m / view / "00000.png")
    p = base / "prompts" / sc / c / asset
    p.mkdir(parents=True)
    pickle.dump({"prompts": ["1 person wears the backpack"], "use_vlm": False}, open(p / "prompts.pickle", "wb"))


def _run(base, out, batch_size):
    from src.generation import inpaint as gi
    args = gi.build_parser().parse_args(
        ["--asset_render_dir", str(base / "renders"), "--asset_mask_dir", str(base / "masks"), "--asset_seg_dir", str(base / "segs"),
         "--prompts_dir", str(base / "prompts"), "--save_dir", str(out), "--num_img_per_combination", "3", "--mask_model", "synthetic",
         "--batch_size", str(batch_size), "--categories", "backpack"])
    args.categories = ["backpack"]
    gi.inpaint_human(args)
    files = sorted(os.path.relpath(os.path.join(r, f), out) for r, _, fs in os.walk(out) for f in fs)
    return files


def test_batched_cli_writes_the_same_files_and_images(tmp_path, hip_lib):
    from PIL import Image
    _tree(tmp_path)
    f1 = _run(tmp_path, tmp_path / "out_b1", 1)
    f4 = _run(tmp_path, tmp_path / "out_b4", 4)
    assert f1 == f4 and len(f1) == 2 * 3                 # "original" + ", full body" (backpack's view_text) x 3 seeds
    assert all(f.endswith((".png",)) and f.split("/")[-1] in ("000000.png", "000001.png", "000002.png") for f in f1)
    worst_mean, worst_frac = 0.0, 0.0
    for f in f1:
        a = np.asarray(Image.open(tmp_path / "out_b1" / f)).astype(np.int32)
        b = np.asarray(Image.open(tmp_path / "out_b4" / f)).astype(np.int32)
        assert a.shape == b.shape == (512, 512, 3)
        d = np.abs(a - b)
        worst_mean, worst_frac = max(worst_mean, d.mean()), max(worst_frac, (d > 8).mean())
    print(f"batched CLI vs one-per-call: worst mean |diff| {worst_mean:.3f} grey levels, worst fraction of pixels off by > 8: {worst_frac:.2e}")
    # different seeds give different images (mean |diff| of two seeds is tens of grey levels): the bar separates the two cleanly
    a = np.asarray(Image.open(tmp_path / "out_b1" / f1[0])).astype(np.int32)
    b = np.asarray(Image.open(tmp_path / "out_b1" / f1[1])).astype(np.int32)
    assert np.abs(a - b).mean() > 10 * max(worst_mean, 0.05)
    assert worst_mean <= 0.4 and worst_frac <= 1e-3           # measured 0.19 grey levels / no pixel off by more than 8
    # skip_done: a second run finds every file and writes nothing
    mt = {f: os.path.getmtime(tmp_path / "out_b4" / f) for f in f4}
    _run(tmp_path, tmp_path / "out_b4", 4)
    assert mt == {f: os.path.getmtime(tmp_path / "out_b4" / f) for f in f4}
