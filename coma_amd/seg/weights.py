"""Parameter table of the person-segmentation network (detectron2 GeneralizedRCNN: ResNet-50 + FPN, RPN, StandardROIHeads with the
PointRend mask head) in detectron2's checkpoint key naming, seeded random initialisation (the checkpoint `model_final_edd263.pkl`,
constants/segmentation.py:5 of the reference, cannot be downloaded here), the loader of a real detectron2 .pkl, and the re-layouts the
HIP kernels want (FrozenBatchNorm folded into the convolution, [N][kh*kw*C] K-contiguous fp32 weights, FC rows permuted to NHWC order).

Architecture constants: imports/pointrend/config/{Base-RCNN-FPN,Base-PointRend-RCNN-FPN,pointrend_rcnn_R_50_FPN_3x_coco}.yaml of the
reference; what those files leave to detectron2's defaults is [3rd-party default, unpinned] (listed in oracle/seg_oracle.py).
"""
from __future__ import annotations

import math
import pickle

import numpy as np
import torch

RES_BLOCKS = (3, 4, 6, 3)                 # RESNETS.DEPTH 50
RES_WIDTH = (64, 128, 256, 512)           # bottleneck channels of res2..res5; outputs are 4 x
FPN_DIM = 256
NUM_CLASSES = 80
NUM_ANCHORS = 3
BOX_FC, MASK_FC, POINT_FC = 1024, 1024, 256
BOX_RES, MASK_POOL, MASK_SIDE = 7, 14, 7
BN_EPS = 1e-5


def seg_shapes():
    s = {}

    def conv_bn(p, cout, cin, k):
        s[p + ".weight"] = (cout, cin, k, k)
        for n in ("weight", "bias", "running_mean", "running_var"):
            s[f"{p}.norm.{n}"] = (cout,)

    bu = "backbone.bottom_up"
    conv_bn(f"{bu}.stem.conv1", 64, 3, 7)
    cin = 64
    for i, (n, w) in enumerate(zip(RES_BLOCKS, RES_WIDTH)):
        for j in range(n):
            p = f"{bu}.res{i + 2}.{j}"
            if j == 0:
                conv_bn(p + ".shortcut", 4 * w, cin, 1)
            conv_bn(p + ".conv1", w, cin, 1)
            conv_bn(p + ".conv2", w, w, 3)
            conv_bn(p + ".conv3", 4 * w, w, 1)
            cin = 4 * w
    for lvl, w in zip((2, 3, 4, 5), RES_WIDTH):
        s[f"backbone.fpn_lateral{lvl}.weight"], s[f"backbone.fpn_lateral{lvl}.bias"] = (FPN_DIM, 4 * w, 1, 1), (FPN_DIM,)
        s[f"backbone.fpn_output{lvl}.weight"], s[f"backbone.fpn_output{lvl}.bias"] = (FPN_DIM, FPN_DIM, 3, 3), (FPN_DIM,)
    r = "proposal_generator.rpn_head"
    s[r + ".conv.weight"], s[r + ".conv.bias"] = (FPN_DIM, FPN_DIM, 3, 3), (FPN_DIM,)
    s[r + ".objectness_logits.weight"], s[r + ".objectness_logits.bias"] = (NUM_ANCHORS, FPN_DIM, 1, 1), (NUM_ANCHORS,)
    s[r + ".anchor_deltas.weight"], s[r + ".anchor_deltas.bias"] = (4 * NUM_ANCHORS, FPN_DIM, 1, 1), (4 * NUM_ANCHORS,)
    h = "roi_heads"
    s[h + ".box_head.fc1.weight"], s[h + ".box_head.fc1.bias"] = (BOX_FC, FPN_DIM * BOX_RES * BOX_RES), (BOX_FC,)
    s[h + ".box_head.fc2.weight"], s[h + ".box_head.fc2.bias"] = (BOX_FC, BOX_FC), (BOX_FC,)
    s[h + ".box_predictor.cls_score.weight"], s[h + ".box_predictor.cls_score.bias"] = (NUM_CLASSES + 1, BOX_FC), (NUM_CLASSES + 1,)
    s[h + ".box_predictor.bbox_pred.weight"], s[h + ".box_predictor.bbox_pred.bias"] = (4 * NUM_CLASSES, BOX_FC), (4 * NUM_CLASSES,)
    c = h + ".mask_head.coarse_head"
    s[c + ".reduce_spatial_dim_conv.weight"], s[c + ".reduce_spatial_dim_conv.bias"] = (FPN_DIM, FPN_DIM, 2, 2), (FPN_DIM,)
    s[c + ".fc1.weight"], s[c + ".fc1.bias"] = (MASK_FC, FPN_DIM * (MASK_POOL // 2) ** 2), (MASK_FC,)
    s[c + ".fc2.weight"], s[c + ".fc2.bias"] = (MASK_FC, MASK_FC), (MASK_FC,)
    s[c + ".prediction.weight"], s[c + ".prediction.bias"] = (NUM_CLASSES * MASK_SIDE * MASK_SIDE, MASK_FC), (NUM_CLASSES * MASK_SIDE * MASK_SIDE,)
    q = h + ".mask_head.point_head"
    cin = FPN_DIM + NUM_CLASSES
    for k in (1, 2, 3):
        s[f"{q}.fc{k}.weight"], s[f"{q}.fc{k}.bias"] = (POINT_FC, cin, 1), (POINT_FC,)
        cin = POINT_FC + NUM_CLASSES
    s[q + ".predictor.weight"], s[q + ".predictor.bias"] = (NUM_CLASSES, cin, 1), (NUM_CLASSES,)
    return s


def random_state(seed=0, cls_gain=6.0, person_bias=1.5, delta_gain=0.3, device="cpu"):
    """Seeded random fp32 parameters that keep every stage in a usable range (activations O(1), proposals that overlap, class scores
    peaked enough to pass the reference's 0.2 threshold -- `cls_gain` -- with the person class favoured by `person_bias`)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in seg_shapes().items():
        if k.endswith("norm.weight"):
            v = 0.75 + 0.5 * torch.rand(shp, generator=g)
            if ".conv3." in k:
                v = v * 0.35                               # residual branches add little: the trunk does not blow up over 16 blocks
        elif k.endswith("running_var"):
            v = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith("running_mean") or k.endswith("norm.bias"):
            v = 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            v = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = int(np.prod(shp[1:]))
            v = torch.randn(shp, generator=g) * math.sqrt(2.0 / fan_in)
            if "stem" in k:
                v = v / 60.0                               # the input is (pixel - mean): +- 128
            if "fpn_" in k or "prediction" in k or "predictor" in k or "objectness" in k:
                v = v * math.sqrt(0.5)                     # no ReLU behind these
            if "anchor_deltas" in k or "bbox_pred" in k:
                v = v * delta_gain
            if "cls_score" in k:
                v = v * cls_gain
        out[k] = v.to(device)
    out["roi_heads.box_predictor.cls_score.bias"][0] += person_bias
    out["roi_heads.box_predictor.cls_score.bias"][NUM_CLASSES] -= 1.0
    return out


def load_detectron2_pkl(path):
    """A detectron2 model-zoo checkpoint ({"model": {name: ndarray}, ...}; names already in the model's own naming for checkpoints
    written by detectron2's trainer, which `model_final_edd263.pkl` is) -> state dict of fp32 tensors, checked against seg_shapes()."""
    with open(path, "rb") as f:
        blob = pickle.load(f, encoding="latin1")
    model = blob.get("model", blob)
    shapes, out = seg_shapes(), {}
    for k, shp in shapes.items():
        if k not in model:
            raise KeyError(f"{path}: parameter '{k}' is missing (is this the PointRend R50-FPN checkpoint?)")
        v = torch.as_tensor(np.asarray(model[k]), dtype=torch.float32)
        if tuple(v.shape) != tuple(shp):
            raise ValueError(f"{path}: '{k}' has shape {tuple(v.shape)}, expected {shp}")
        out[k] = v
    return out


# ------------------------------------------------------------------ layouts for the kernels
def _pad_k(w2d, mult=32):
    n, k = w2d.shape
    kp = -(-k // mult) * mult
    if kp == k:
        return w2d.contiguous()
    out = torch.zeros(n, kp, dtype=w2d.dtype)
    out[:, :k] = w2d
    return out


def conv_weight(w, cpad=None):
    """[N, C, kh, kw] -> fp32 [N][(ky * kw + kx) * Cp + c], K zero-padded to a multiple of 32; cpad: input channels padded (3 -> 4)."""
    n, c, kh, kw = w.shape
    w = w.permute(0, 2, 3, 1)
    if cpad is not None and cpad != c:
        w = torch.cat([w, torch.zeros(n, kh, kw, cpad - c, dtype=w.dtype)], dim=3)
    return _pad_k(w.reshape(n, -1))


def fold_bn(state, p):
    """conv `p` followed by FrozenBatchNorm2d `p.norm` -> (weight scaled per output channel, bias)."""
    scale = state[p + ".norm.weight"] * (state[p + ".norm.running_var"] + BN_EPS).rsqrt()
    return state[p + ".weight"] * scale.view(-1, 1, 1, 1), state[p + ".norm.bias"] - state[p + ".norm.running_mean"] * scale


def fc_from_chw(w, c, side):
    """Linear over a flattened NCHW [c, side, side] input -> the same Linear over the NHWC flattening ((y * side + x) * c + ch)."""
    n = w.shape[0]
    return w.view(n, c, side, side).permute(0, 2, 3, 1).reshape(n, -1)


def prepare(state):
    """Everything the device plan reads, as fp32 CPU tensors keyed by a short name: (w [N][Kpad], bias [N]) pairs."""
    s = {k: v.detach().float().cpu() for k, v in state.items()}
    P = {}
    bu = "backbone.bottom_up"
    w, b = fold_bn(s, f"{bu}.stem.conv1")
    P["stem"] = (conv_weight(w, cpad=4), b)
    for i, n in enumerate(RES_BLOCKS):
        for j in range(n):
            p = f"{bu}.res{i + 2}.{j}"
            for c in ("shortcut", "conv1", "conv2", "conv3"):
                if f"{p}.{c}.weight" in s:
                    w, b = fold_bn(s, f"{p}.{c}")
                    P[f"res{i + 2}.{j}.{c}"] = (conv_weight(w), b)
    for lvl in (2, 3, 4, 5):
        for n in ("lateral", "output"):
            P[f"fpn_{n}{lvl}"] = (conv_weight(s[f"backbone.fpn_{n}{lvl}.weight"]), s[f"backbone.fpn_{n}{lvl}.bias"])
    r = "proposal_generator.rpn_head"
    P["rpn_conv"] = (conv_weight(s[r + ".conv.weight"]), s[r + ".conv.bias"])
    # objectness (3) and anchor deltas (12) as ONE 1x1 convolution with 15 output channels: [logit a0..a2 | a0: dx dy dw dh | a1 ... | a2 ...]
    P["rpn_pred"] = (conv_weight(torch.cat([s[r + ".objectness_logits.weight"], s[r + ".anchor_deltas.weight"]])),
                     torch.cat([s[r + ".objectness_logits.bias"], s[r + ".anchor_deltas.bias"]]))
    h = "roi_heads"
    P["box_fc1"] = (_pad_k(fc_from_chw(s[h + ".box_head.fc1.weight"], FPN_DIM, BOX_RES)), s[h + ".box_head.fc1.bias"])
    P["box_fc2"] = (_pad_k(s[h + ".box_head.fc2.weight"]), s[h + ".box_head.fc2.bias"])
    # class scores (81) and box deltas (320) as ONE linear with 401 outputs
    P["box_pred"] = (_pad_k(torch.cat([s[h + ".box_predictor.cls_score.weight"], s[h + ".box_predictor.bbox_pred.weight"]])),
                     torch.cat([s[h + ".box_predictor.cls_score.bias"], s[h + ".box_predictor.bbox_pred.bias"]]))
    c = h + ".mask_head.coarse_head"
    P["coarse_conv"] = (conv_weight(s[c + ".reduce_spatial_dim_conv.weight"]), s[c + ".reduce_spatial_dim_conv.bias"])
    P["coarse_fc1"] = (_pad_k(fc_from_chw(s[c + ".fc1.weight"], FPN_DIM, MASK_POOL // 2)), s[c + ".fc1.bias"])
    P["coarse_fc2"] = (_pad_k(s[c + ".fc2.weight"]), s[c + ".fc2.bias"])
    # prediction rows (class, y, x) -> (y, x, class): the coarse map leaves the GEMM as NHWC [R, 7, 7, 80]
    wp = s[c + ".prediction.weight"].view(NUM_CLASSES, MASK_SIDE, MASK_SIDE, -1).permute(1, 2, 0, 3).reshape(NUM_CLASSES * MASK_SIDE * MASK_SIDE, -1)
    bp = s[c + ".prediction.bias"].view(NUM_CLASSES, MASK_SIDE, MASK_SIDE).permute(1, 2, 0).reshape(-1)
    P["coarse_pred"] = (_pad_k(wp), bp.contiguous())
    q = h + ".mask_head.point_head"
    for k in (1, 2, 3):
        P[f"point_fc{k}"] = (_pad_k(s[f"{q}.fc{k}.weight"].squeeze(-1)), s[f"{q}.fc{k}.bias"])
    P["point_pred"] = (s[q + ".predictor.weight"].squeeze(-1).contiguous(), s[q + ".predictor.bias"])      # [80][336], read row-wise by class
    return P
