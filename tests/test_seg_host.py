"""CPU: host side of the person-segmentation path -- the oracle's one pinned piece (PIL's bilinear resize), the product's
coefficient tables, the parameter table / checkpoint loader / weight re-layouts of coma_amd/seg/weights.py against plain torch, the oracle's
NMS against a brute-force restatement, and the plug-in factory.  No kernel is launched here."""
import math
import pickle

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from PIL import Image

from coma_amd.seg import model as M
from coma_amd.seg import weights as W
from oracle import seg_oracle as so


@pytest.mark.parametrize("h,w,nh,nw", [(512, 512, 800, 800), (37, 53, 61, 40), (50, 20, 20, 50), (480, 640, 800, 1067)])
def test_resize_restatement_is_pil(h, w, nh, nw):
    """G21: `resize_bilinear_u8_ref` == PIL.Image.resize(BILINEAR) bit for bit (up- and down-scaling, both axes), and the product's
    vectorised tables == the oracle's loops.  This is what detectron2's ResizeTransform.apply_image does to a uint8 image."""
    rng = np.random.default_rng(h * 1000 + w)
    a = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
    assert np.array_equal(so.resize_bilinear_u8_ref(a, nh, nw), np.asarray(Image.fromarray(a).resize((nw, nh), Image.BILINEAR)))
    for i, o in ((w, nw), (h, nh)):
        b0, k0 = so.bilinear_coeffs(i, o)
        b1, k1 = M.bilinear_tables(i, o)
        assert np.array_equal(b0, b1) and np.array_equal(k0, k1)


def test_shortest_edge_rule():
    assert M.shortest_edge_size(512, 512) == so.shortest_edge_size(512, 512) == (800, 800)
    assert M.shortest_edge_size(480, 640) == (800, 1067) and M.shortest_edge_size(640, 480) == (1067, 800)
    assert M.shortest_edge_size(300, 1000) == so.shortest_edge_size(300, 1000) == (400, 1333)          # the 1333 cap
    x, (nh, nw) = so.preprocess(np.zeros((1, 480, 640, 3), np.uint8))
    assert (nh, nw) == (800, 1067) and tuple(x.shape) == (1, 3, 800, 1088)
    assert float(x[0, 0, 0, 0]) == pytest.approx(-103.53) and float(x[0, 2, 0, 0]) == pytest.approx(-123.675) and float(x[0, :, :, 1067:].abs().max()) == 0


def test_architecture_constants():
    assert so.subdivision_schedule() == (28, 3) == (M.INIT_RES, M.SUBDIV_STEPS)           # 7 -> 28, 5 -> 3: maps of 28, 56, 112, 224
    a = so.cell_anchors(32)
    assert torch.allclose(a[0], torch.tensor([-22.6274, -11.3137, 22.6274, 11.3137]), atol=1e-4) and torch.equal(a, M.cell_anchors(32))
    g = so.grid_anchors(2, 3, 4, 32)
    assert tuple(g.shape) == (18, 4) and torch.equal(g[3], a[0] + torch.tensor([4.0, 0.0, 4.0, 0.0]))      # (y, x, anchor) order
    shapes = W.seg_shapes()
    assert shapes["roi_heads.mask_head.point_head.fc1.weight"] == (256, 336, 1) and shapes["roi_heads.mask_head.coarse_head.fc1.weight"] == (1024, 12544)
    assert shapes["roi_heads.box_predictor.bbox_pred.weight"] == (320, 1024) and "backbone.bottom_up.res5.2.conv3.norm.running_var" in shapes
    assert sum(int(np.prod(v)) for k, v in shapes.items() if "running" not in k and "norm" not in k) > 55e6


def test_weight_relayouts_match_torch():
    g = torch.Generator().manual_seed(0)
    s = {"c.weight": torch.randn(8, 5, 3, 3, generator=g), "c.norm.weight": torch.rand(8, generator=g) + 0.5, "c.norm.bias": torch.randn(8, generator=g),
         "c.norm.running_mean": torch.randn(8, generator=g), "c.norm.running_var": torch.rand(8, generator=g) + 0.5}
    x = torch.randn(2, 5, 6, 7, generator=g)
    ref = so._cbn(x, s, "c", padding=1)
    w, b = W.fold_bn(s, "c")
    assert torch.allclose(F.conv2d(x, w, b, padding=1), ref, atol=1e-5)
    # [N][ (ky*kw+kx)*C + c ] with K padded to 32, input channels padded 5 -> 8
    wk = W.conv_weight(w, cpad=8)
    assert tuple(wk.shape) == (8, 96) and float(wk[:, 72:].abs().max()) == 0
    xp = F.pad(x, (1, 1, 1, 1)).permute(0, 2, 3, 1)
    patch = torch.zeros(72)
    for t, (ky, kx) in enumerate((a, c) for a in range(3) for c in range(3)):
        patch[t * 8:t * 8 + 5] = xp[1, 2 + ky, 3 + kx]
    assert torch.allclose(wk[:, :72] @ patch + b, ref[1, :, 2, 3], atol=1e-5)
    # a Linear over NCHW-flattened ROI features == the re-ordered Linear over the NHWC flattening
    fw = torch.randn(6, 4 * 3 * 3, generator=g)
    roi = torch.randn(4, 3, 3, generator=g)
    assert torch.allclose(W.fc_from_chw(fw, 4, 3) @ roi.permute(1, 2, 0).reshape(-1), fw @ roi.reshape(-1), atol=1e-5)


def test_prepare_and_checkpoint_loader(tmp_path):
    state = W.random_state(seed=3)
    P = W.prepare(state)
    assert tuple(P["stem"][0].shape) == (64, 224) and tuple(P["rpn_pred"][0].shape) == (15, 256) and tuple(P["box_pred"][0].shape) == (401, 1024)
    assert tuple(P["point_fc1"][0].shape) == (256, 352) and tuple(P["coarse_pred"][0].shape) == (3920, 1024) and all(w.shape[1] % 32 == 0 for k, (w, _) in P.items() if k != "point_pred")
    # the coarse prediction leaves the GEMM as [7][7][80]: row (y, x, c) of the re-ordered weight = row (c, y, x) of detectron2's
    wp = state["roi_heads.mask_head.coarse_head.prediction.weight"]
    assert torch.equal(P["coarse_pred"][0][(2 * 7 + 3) * 80 + 5], wp[5 * 49 + 2 * 7 + 3])
    pth = tmp_path / "model_final.pkl"
    with open(pth, "wb") as f:
        pickle.dump({"model": {k: v.numpy() for k, v in state.items()}, "__author__": "test"}, f)
    back = W.load_detectron2_pkl(pth)
    assert all(torch.equal(back[k], state[k]) for k in state)
    bad = {k: v.numpy() for k, v in state.items()}
    del bad["roi_heads.mask_head.point_head.fc2.bias"]
    with open(pth, "wb") as f:
        pickle.dump({"model": bad}, f)
    with pytest.raises(KeyError, match="point_head.fc2.bias"):
        W.load_detectron2_pkl(pth)


def test_oracle_nms_against_brute_force():
    rng = np.random.default_rng(5)
    n = 300
    c = rng.uniform(0, 100, (n, 2))
    wh = rng.uniform(5, 40, (n, 2))
    boxes = torch.tensor(np.concatenate([c - wh / 2, c + wh / 2], 1), dtype=torch.float32)
    scores = torch.tensor(rng.normal(size=n), dtype=torch.float32)
    scores[10:20] = scores[10]
    groups = rng.integers(0, 3, n)
    keep = so.nms_ref(boxes, scores, groups, 0.5).tolist()
    order = sorted(range(n), key=lambda i: (-float(scores[i]), i))
    area = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])).numpy()
    kept = []
    for i in order:
        ok = True
        for j in kept:
            if groups[i] != groups[j]:
                continue
            iw = max(0.0, min(float(boxes[i, 2]), float(boxes[j, 2])) - max(float(boxes[i, 0]), float(boxes[j, 0])))
            ih = max(0.0, min(float(boxes[i, 3]), float(boxes[j, 3])) - max(float(boxes[i, 1]), float(boxes[j, 1])))
            if np.float32(iw * ih) / np.float32(area[i] + area[j] - np.float32(iw * ih)) > 0.5:
                ok = False
                break
        if ok:
            kept.append(i)
    assert keep == kept


def test_oracle_point_sample_and_paste_conventions():
    """point_sample at the centres of a regular grid reads the pixels themselves; a mask pasted into its own box at the mask's resolution is itself."""
    x = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).view(2, 3, 4, 5)
    cc = torch.stack(torch.meshgrid((torch.arange(4) + 0.5) / 4, (torch.arange(5) + 0.5) / 5, indexing="ij"), -1).flip(-1).reshape(1, 20, 2).expand(2, -1, -1)
    assert torch.allclose(so.point_sample(x, cc), x.flatten(2), atol=1e-4)
    assert torch.allclose(so.regular_grid(1, 5)[0, 7], torch.tensor([0.5, 0.3]))                         # x fastest: point 7 = (x 2, y 1)
    p = torch.rand(1, 6, 6, generator=torch.Generator().manual_seed(1))
    out = so.paste_masks(p, torch.tensor([[2.0, 3.0, 8.0, 9.0]]), 12, 12, threshold=0.5)
    assert torch.equal(out[0, 3:9, 2:8], p[0] >= 0.5) and not bool(out[0, :3].any()) and not bool(out[0, :, 8:].any())


def test_plugin_factory_prefers_the_device_detector():
    from coma_amd.sd import predictors as pr
    from coma_amd.seg.predictor import HipPointRendBackend, HipPointRendPredictor
    state = {"x": torch.zeros(1)}                       # never prepared: plans are built on first use
    m = pr.build_adaptive_mask_model("p", 0.2, device="cuda", pointrend_state=state)
    assert isinstance(m, HipPointRendPredictor) and m.accepts_device_tensor and hasattr(m, "predict_batch") and m.merge_mode == "merge"
    assert isinstance(pr.pointrend_backend(0.8, "cuda", state=state), HipPointRendBackend)
    assert pr.COCO_SEG_WEIGHTS_PTH.endswith("imports/pointrend/weights/model_final_edd263.pkl")          # constants/segmentation.py:5
    with pytest.raises(ImportError, match="detectron2"):                                                  # neither checkpoint nor detectron2
        pr.build_adaptive_mask_model("p", 0.2)
    with pytest.raises(ValueError, match="0.125"):
        from coma_amd.seg.model import HipPointRend
        HipPointRend(state, 1, 64, 64, "cpu", score_thresh=0.05)
