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
