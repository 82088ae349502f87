"""CPU oracle (TEST INFRASTRUCTURE -- never imported by the product path) for K13 / SURVEY.md 8f-2: the person-segmentation
network behind `PointRendPredictor.__call__` (utils/adaptive_mask_inpainting.py:1225-1236: `DefaultPredictor(cfg)(image)` -> instances ->
person masks merged) and the post-inpaint stage (src/generation/segment_human.py:24-169), restated in torch fp32 (the precision the
reference runs it in: cfg.MODEL.DEVICE = "cuda", no autocast).

**PARITY UNPINNED.**  detectron2 (GeneralizedRCNN, ResNet, FPN, RPN, StandardROIHeads, the PointRend project) and torchvision
(roi_align, nms) are third-party and absent from this image, and the checkpoint (`model_final_edd263.pkl`) cannot be fetched: nothing
reference-held can pin this file.  What IS under /root/reference pins the architecture constants -- every one is cited below to
imports/pointrend/config/{Base-RCNN-FPN,Base-PointRend-RCNN-FPN,pointrend_rcnn_R_50_FPN_3x_coco}.yaml ("cfg:" tags) or to the call
sites; everything else is detectron2's / torchvision's published default, marked [3rd-party default, unpinned].  The one piece that IS
pinned is the image resize: PIL is in the image, and `resize_bilinear_u8_ref` is checked against PIL.Image.resize bit for bit.

Tie rules (torch.topk / sort tie order is unspecified upstream): every "top-k" here takes ties in ASCENDING INDEX order, every NMS
visits boxes by (score descending, list position ascending) -- written out so that the device path can be compared index for index.

Layout: NCHW fp32 tensors, detectron2 checkpoint key names (coma_amd/seg/weights.py builds the same table).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

# ------------------------------------------------------------------ constants
# cfg: Base-RCNN-FPN.yaml
ANCHOR_SIZES = (32, 64, 128, 256, 512)            # ANCHOR_GENERATOR.SIZES, one per level p2..p6
ANCHOR_RATIOS = (0.5, 1.0, 2.0)                   # ANCHOR_GENERATOR.ASPECT_RATIOS
RPN_PRE_NMS_TOPK = 1000                           # RPN.PRE_NMS_TOPK_TEST (per level)
RPN_POST_NMS_TOPK = 1000                          # RPN.POST_NMS_TOPK_TEST
BOX_POOLER_RES = 7                                # ROI_BOX_HEAD.POOLER_RESOLUTION; NUM_FC 2
MASK_POOLER_RES = 14                              # ROI_MASK_HEAD.POOLER_RESOLUTION (Base-RCNN-FPN.yaml; PointRend's coarse head pools 14 x 14)
# cfg: Base-PointRend-RCNN-FPN.yaml
MASK_FC_DIM, MASK_NUM_FC, MASK_SIDE = 1024, 2, 7  # ROI_MASK_HEAD.{FC_DIM, NUM_FC, OUTPUT_SIDE_RESOLUTION}; IN_FEATURES ["p2"]
POINT_FC_DIM, POINT_NUM_FC = 256, 3               # POINT_HEAD.{FC_DIM, NUM_FC}; IN_FEATURES ["p2"]
# cfg: pointrend_rcnn_R_50_FPN_3x_coco.yaml: RESNETS.DEPTH 50
RES_BLOCKS = (3, 4, 6, 3)
# [3rd-party default, unpinned] detectron2/config/defaults.py and projects/PointRend/point_rend/config.py
PIXEL_MEAN = (103.530, 116.280, 123.675)          # MODEL.PIXEL_MEAN (BGR order); PIXEL_STD = 1
MIN_SIZE_TEST, MAX_SIZE_TEST = 800, 1333          # INPUT.{MIN,MAX}_SIZE_TEST
SIZE_DIVISIBILITY = 32                            # FPN backbone
FPN_DIM = 256                                     # FPN.OUT_CHANNELS
STRIDE_IN_1X1 = True                              # RESNETS.STRIDE_IN_1X1 (MSRA R-50)
BN_EPS = 1e-5                                     # FrozenBatchNorm2d
RPN_NMS_THRESH = 0.7                              # RPN.NMS_THRESH
RPN_BBOX_WEIGHTS = (1.0, 1.0, 1.0, 1.0)           # RPN.BBOX_REG_WEIGHTS
ROI_BBOX_WEIGHTS = (10.0, 10.0, 5.0, 5.0)         # ROI_BOX_HEAD.BBOX_REG_WEIGHTS
SCALE_CLAMP = math.log(1000.0 / 16)               # Box2BoxTransform
NUM_CLASSES = 80                                  # ROI_HEADS.NUM_CLASSES = POINT_HEAD.NUM_CLASSES
ROI_NMS_THRESH = 0.5                              # ROI_HEADS.NMS_THRESH_TEST
DETECTIONS_PER_IMAGE = 100                        # TEST.DETECTIONS_PER_IMAGE
BOX_FC_DIM = 1024                                 # ROI_BOX_HEAD.FC_DIM
CANONICAL_BOX_SIZE, CANONICAL_LEVEL = 224, 4      # ROIPooler
SUBDIVISION_STEPS, SUBDIVISION_NUM_POINTS = 5, 28 * 28      # POINT_HEAD.{SUBDIVISION_STEPS, SUBDIVISION_NUM_POINTS}
MASK_THRESHOLD = 0.5                              # detector_postprocess


def subdivision_schedule():
    """PointRendMaskHead._init_point_head: while 4 * res^2 <= num_points the first subdivision would recompute every pixel anyway, so the
    initial grid doubles and a step is dropped: 7 -> 28, 5 -> 3 steps (28, 56, 112, 224)."""
    res, steps = MASK_SIDE, SUBDIVISION_STEPS
    while 4 * res * res <= SUBDIVISION_NUM_POINTS:
        res, steps = res * 2, steps - 1
    return res, steps


# ------------------------------------------------------------------ image resize (DefaultPredictor: ResizeShortestEdge -> PIL bilinear on uint8)
def shortest_edge_size(h, w, short=MIN_SIZE_TEST, max_size=MAX_SIZE_TEST):
    """detectron2 ResizeShortestEdge.get_output_shape [3rd-party, unpinned]."""
    scale = short * 1.0 / min(h, w)
    newh, neww = (short, scale * w) if h < w else (scale * h, short)
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def bilinear_coeffs(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR (triangle, support 1) filter: per output position the first
    source index, the number of taps, and the fixed-point weights (22 fractional bits).  -> bounds int32 [out,2], kk int32 [out,ksize]."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = np.zeros(ksize, np.float64)
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            v = -v if v < 0 else v
            w[x] = 1.0 - v if v < 1.0 else 0.0
        ww = w[:xmax].sum()
        if ww != 0.0:
            w[:xmax] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << 22)) if w[x] < 0 else int(0.5 + w[x] * (1 << 22))
    return bounds, kk


def _resample_axis_u8(img, bounds, kk, axis):
    """One pass of ImagingResample{Horizontal,Vertical}_8bpc: ss = sum(pixel * k) + 2^21, >> 22, clipped to [0, 255]."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + img.shape[1:], np.uint8)
    for o in range(bounds.shape[0]):
        x0, n = int(bounds[o, 0]), int(bounds[o, 1])
        ss = np.tensordot(kk[o, :n].astype(np.int64), img[x0:x0 + n], axes=(0, 0)) + (1 << 21)
        out[o] = np.clip(ss >> 22, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8_ref(img_u8, new_h, new_w):
    """PIL.Image.fromarray(img).resize((new_w, new_h), BILINEAR) restated: horizontal pass first (uint8 in between), then vertical."""
    h, w = img_u8.shape[:2]
    out = img_u8
    if new_w != w:
        out = _resample_axis_u8(out, *bilinear_coeffs(w, new_w), axis=1)
    if new_h != h:
        out = _resample_axis_u8(out, *bilinear_coeffs(h, new_h), axis=0)
    return out


def preprocess(images_u8):
    """DefaultPredictor.__call__ + GeneralizedRCNN.preprocess_image for images of one size: uint8 [B,H,W,3] IN THE CHANNEL ORDER GIVEN
    (INPUT.FORMAT = "BGR" means no flip: the adaptive loop hands RGB, utils/adaptive_mask_inpainting.py:1227 -- so R meets the B mean;
    src/generation/segment_human.py:128 hands cv2's BGR) -> (x fp32 [B,3,Hp,Wp] = resized - mean, zero-padded to a multiple of 32,
    (new_h, new_w))."""
    images_u8 = np.asarray(images_u8)
    B, H, W, _ = images_u8.shape
    nh, nw = shortest_edge_size(H, W)
    x = np.stack([resize_bilinear_u8_ref(im, nh, nw) for im in images_u8]).astype(np.float32)
    x = torch.from_numpy(x).permute(0, 3, 1, 2) - torch.tensor(PIXEL_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    hp, wp = -(-nh // SIZE_DIVISIBILITY) * SIZE_DIVISIBILITY, -(-nw // SIZE_DIVISIBILITY) * SIZE_DIVISIBILITY
    return F.pad(x, (0, wp - nw, 0, hp - nh)), (nh, nw)


# ------------------------------------------------------------------ backbone
def _bn(x, s, p):
    """FrozenBatchNorm2d: x * scale + bias with scale = weight * rsqrt(running_var + eps), bias = bias - running_mean * scale."""
    scale = s[p + ".weight"] * (s[p + ".running_var"] + BN_EPS).rsqrt()
    shift = s[p + ".bias"] - s[p + ".running_mean"] * scale
    return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def _cbn(x, s, p, stride=1, padding=0):
    return _bn(F.conv2d(x, s[p + ".weight"], None, stride=stride, padding=padding), s, p + ".norm")


def resnet50(x, s, pre="backbone.bottom_up"):
    x = F.relu(_cbn(x, s, f"{pre}.stem.conv1", stride=2, padding=3))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = {}
    for i, n in enumerate(RES_BLOCKS):
        for j in range(n):
            p = f"{pre}.res{i + 2}.{j}"
            stride = 2 if (j == 0 and i > 0) else 1
            sc = _cbn(x, s, p + ".shortcut", stride=stride) if (p + ".shortcut.weight") in s else x
            s1, s3 = (stride, 1) if STRIDE_IN_1X1 else (1, stride)
            h = F.relu(_cbn(x, s, p + ".conv1", stride=s1))
            h = F.relu(_cbn(h, s, p + ".conv2", stride=s3, padding=1))
            h = _cbn(h, s, p + ".conv3")
            x = F.relu(h + sc)
        outs[f"res{i + 2}"] = x
    return outs


def fpn(res, s, pre="backbone"):
    """FPN (fuse_type "sum", no norm) + LastLevelMaxPool -> p2..p6."""
    out, prev = {}, None
    for lvl in (5, 4, 3, 2):
        lat = F.conv2d(res[f"res{lvl}"], s[f"{pre}.fpn_lateral{lvl}.weight"], s[f"{pre}.fpn_lateral{lvl}.bias"])
        prev = lat if prev is None else lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
        out[f"p{lvl}"] = F.conv2d(prev, s[f"{pre}.fpn_output{lvl}.weight"], s[f"{pre}.fpn_output{lvl}.bias"], padding=1)
    out["p6"] = F.max_pool2d(out["p5"], kernel_size=1, stride=2, padding=0)
    return out


# ------------------------------------------------------------------ boxes
def cell_anchors(size):
    a = []
    for r in ANCHOR_RATIOS:
        w = math.sqrt(size * size / r)
        h = r * w
        a.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(a, dtype=torch.float32)


def grid_anchors(h, w, stride, size):
    """DefaultAnchorGenerator (offset 0): [(y * w + x) * 3 + a, 4]."""
    sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
    return (shifts.view(-1, 1, 4) + cell_anchors(size).view(1, -1, 4)).reshape(-1, 4)


def apply_deltas(deltas, boxes, weights):
    """Box2BoxTransform.apply_deltas; deltas [N, k*4], boxes [N, 4] -> [N, k*4]."""
    boxes = boxes.to(deltas.dtype)
    widths, heights = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    ctr_x, ctr_y = boxes[:, 0] + 0.5 * widths, boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx, dy = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy
    dw, dh = torch.clamp(deltas[:, 2::4] / ww, max=SCALE_CLAMP), torch.clamp(deltas[:, 3::4] / wh, max=SCALE_CLAMP)
    pcx, pcy = dx * widths[:, None] + ctr_x[:, None], dy * heights[:, None] + ctr_y[:, None]
    pw, ph = torch.exp(dw) * widths[:, None], torch.exp(dh) * heights[:, None]
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=-1).reshape(deltas.shape)


def clip_boxes(b, h, w):
    b = b.clone()
    b[..., 0::2] = b[..., 0::2].clamp(min=0, max=w)
    b[..., 1::2] = b[..., 1::2].clamp(min=0, max=h)
    return b


def topk_stable(x, k):
    """Indices of the k largest entries of a 1-D tensor, descending, ties in ascending index order (the rule of this build)."""
    return torch.sort(x, descending=True, stable=True)[1][:k]


def nms_ref(boxes, scores, groups, thresh):
    """torchvision.ops.nms semantics per group ("vanilla" batched_nms: IoU on the raw coordinates, only boxes of the same group suppress
    each other; torchvision's coordinate-trick variant adds group offsets to the coordinates first, which can move an IoU by an ulp --
    [3rd-party, unpinned]): visit by (score descending, position ascending); a box is dropped when an earlier KEPT box of its group has
    IoU > thresh, IoU = inter / (area_i + area_j - inter) in fp32.  -> kept positions in visiting order (int64)."""
    b = boxes.detach().numpy().astype(np.float32)
    g = np.asarray(groups).astype(np.int64)
    order = torch.sort(scores, descending=True, stable=True)[1].numpy()
    b, g = b[order], g[order]
    n = len(order)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(order[i])
        if i + 1 == n:
            break
        xx1, yy1 = np.maximum(b[i, 0], b[i + 1:, 0]), np.maximum(b[i, 1], b[i + 1:, 1])
        xx2, yy2 = np.minimum(b[i, 2], b[i + 1:, 2]), np.minimum(b[i, 3], b[i + 1:, 3])
        inter = np.maximum(np.float32(0), xx2 - xx1) * np.maximum(np.float32(0), yy2 - yy1)
        with np.errstate(invalid="ignore", divide="ignore"):
            iou = inter / (area[i] + area[i + 1:] - inter)
        dead[i + 1:] |= (iou > np.float32(thresh)) & (g[i + 1:] == g[i])
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


# ------------------------------------------------------------------ RPN
def rpn_head(feats, s, pre="proposal_generator.rpn_head"):
    logits, deltas = [], []
    for f in feats:
        t = F.relu(F.conv2d(f, s[pre + ".conv.weight"], s[pre + ".conv.bias"], padding=1))
        logits.append(F.conv2d(t, s[pre + ".objectness_logits.weight"], s[pre + ".objectness_logits.bias"]))
        deltas.append(F.conv2d(t, s[pre + ".anchor_deltas.weight"], s[pre + ".anchor_deltas.bias"]))
    return logits, deltas


def rpn_proposals(logits, deltas, image_size, strides=(4, 8, 16, 32, 64)):
    """RPN.predict_proposals + find_top_rpn_proposals for ONE image (tensors [1, ...]): -> dict(boxes [n,4], logits [n], level [n],
    cand_* = the pre-NMS candidate list in concatenation order)."""
    cb, cs, cl, ci = [], [], [], []
    for lvl, (lg, dl) in enumerate(zip(logits, deltas)):
        _, A, H, W = lg.shape
        lg = lg[0].permute(1, 2, 0).reshape(-1)                                   # (H, W, A)
        dl = dl[0].view(A, 4, H, W).permute(2, 3, 0, 1).reshape(-1, 4)
        anchors = grid_anchors(H, W, strides[lvl], ANCHOR_SIZES[lvl])
        k = min(lg.numel(), RPN_PRE_NMS_TOPK)
        idx = topk_stable(lg, k)
        cb.append(apply_deltas(dl[idx], anchors[idx], RPN_BBOX_WEIGHTS))
        cs.append(lg[idx])
        cl.append(torch.full((k,), lvl, dtype=torch.int64))
        ci.append(idx)
    boxes, scores, lvls = torch.cat(cb), torch.cat(cs), torch.cat(cl)
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores)
    boxes = clip_boxes(boxes, *image_size)
    ok = valid & ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)      # nonempty(threshold = min_box_size = 0)
    pos = torch.nonzero(ok).squeeze(1)
    keep = pos[nms_ref(boxes[ok], scores[ok], lvls[ok], RPN_NMS_THRESH)][:RPN_POST_NMS_TOPK]
    return dict(boxes=boxes[keep], logits=scores[keep], level=lvls[keep], keep=keep, cand_boxes=boxes, cand_scores=scores, cand_level=lvls,
                cand_anchor=torch.cat(ci), cand_ok=ok)


# ------------------------------------------------------------------ ROIAlign (torchvision.ops.roi_align, aligned=True, sampling_ratio=0)
def _bilinear_roialign(feat, y, x):
    """torchvision's bilinear_interpolate: feat [C,H,W]; y, x 1-D sample coordinates (outer product grid) -> [C, len(y), len(x)]."""
    C, H, W = feat.shape
    vy, vx = (y >= -1.0) & (y <= H), (x >= -1.0) & (x <= W)
    y, x = y.clamp(min=0), x.clamp(min=0)
    y0, x0 = y.floor().long(), x.floor().long()
    y_hi, x_hi = y0 >= H - 1, x0 >= W - 1
    y0, x0 = torch.where(y_hi, torch.full_like(y0, H - 1), y0), torch.where(x_hi, torch.full_like(x0, W - 1), x0)
    y1, x1 = torch.where(y_hi, y0, y0 + 1), torch.where(x_hi, x0, x0 + 1)
    y, x = torch.where(y_hi, y0.to(y.dtype), y), torch.where(x_hi, x0.to(x.dtype), x)
    ly, lx = y - y0, x - x0
    hy, hx = 1.0 - ly, 1.0 - lx
    f = lambda yi, xi: feat[:, yi][:, :, xi]
    w1, w2, w3, w4 = hy[:, None] * hx[None], hy[:, None] * lx[None], ly[:, None] * hx[None], ly[:, None] * lx[None]
    val = w1 * f(y0, x0) + w2 * f(y0, x1) + w3 * f(y1, x0) + w4 * f(y1, x1)
    return val * (vy[:, None] & vx[None]).to(val.dtype)


def roi_align_ref(feat, boxes, out_size, scale):
    """feat [C,H,W], boxes [R,4] -> [R,C,out,out]; aligned (half-pixel shift), grid = ceil(roi / out) samples per bin and axis, mean."""
    out = []
    for b in boxes:
        x1, y1, x2, y2 = (float(v) for v in (b * scale - 0.5).to(torch.float32))
        x1, y1, x2, y2 = np.float32(x1), np.float32(y1), np.float32(x2), np.float32(y2)
        rw, rh = np.float32(x2 - x1), np.float32(y2 - y1)
        bw, bh = np.float32(rw / np.float32(out_size)), np.float32(rh / np.float32(out_size))
        gh, gw = max(int(math.ceil(float(rh) / out_size)), 0), max(int(math.ceil(float(rw) / out_size)), 0)
        if gh == 0 or gw == 0:
            out.append(torch.zeros(feat.shape[0], out_size, out_size))
            continue
        iy = torch.arange(out_size * gh, dtype=torch.float32)
        ix = torch.arange(out_size * gw, dtype=torch.float32)
        ys = float(y1) + (iy // gh) * float(bh) + ((iy % gh) + 0.5) * float(bh) / gh
        xs = float(x1) + (ix // gw) * float(bw) + ((ix % gw) + 0.5) * float(bw) / gw
        v = _bilinear_roialign(feat, ys, xs)
        out.append(v.view(-1, out_size, gh, out_size, gw).sum(dim=(2, 4)) / float(max(gh * gw, 1)))
    return torch.stack(out) if out else torch.zeros(0, feat.shape[0], out_size, out_size)


def assign_levels(boxes, min_level=2, max_level=5):
    """ROIPooler assign_boxes_to_levels -> level index 0..3."""
    sizes = torch.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    lv = torch.floor(CANONICAL_LEVEL + torch.log2(sizes / CANONICAL_BOX_SIZE + 1e-8))
    return (torch.clamp(lv, min=min_level, max=max_level).to(torch.int64) - min_level)


def box_pooler(feats, boxes):
    """feats: [p2..p5] each [C,H,W] of one image -> [R, C, 7, 7]."""
    lv = assign_levels(boxes)
    out = torch.zeros(len(boxes), feats[0].shape[0], BOX_POOLER_RES, BOX_POOLER_RES)
    for l, f in enumerate(feats):
        idx = torch.nonzero(lv == l).squeeze(1)
        if len(idx):
            out[idx] = roi_align_ref(f, boxes[idx], BOX_POOLER_RES, 1.0 / (4 << l))
    return out, lv


# ------------------------------------------------------------------ box head + detections
def box_head(x, s, pre="roi_heads"):
    x = x.flatten(1)
    x = F.relu(F.linear(x, s[f"{pre}.box_head.fc1.weight"], s[f"{pre}.box_head.fc1.bias"]))
    x = F.relu(F.linear(x, s[f"{pre}.box_head.fc2.weight"], s[f"{pre}.box_head.fc2.bias"]))
    return (F.linear(x, s[f"{pre}.box_predictor.cls_score.weight"], s[f"{pre}.box_predictor.cls_score.bias"]),
            F.linear(x, s[f"{pre}.box_predictor.bbox_pred.weight"], s[f"{pre}.box_predictor.bbox_pred.bias"]))


def fast_rcnn_inference(cls_logits, box_deltas, proposals, image_size, score_thresh):
    """FastRCNNOutputLayers.inference for one image -> dict(boxes [n,4], scores [n], classes [n], roi [n], cand_* pre-NMS lists)."""
    probs = F.softmax(cls_logits, dim=-1)
    boxes = apply_deltas(box_deltas, proposals, ROI_BBOX_WEIGHTS)
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(probs).all(dim=1)
    scores = probs[:, :-1]
    boxes = clip_boxes(boxes.view(-1, NUM_CLASSES, 4), *image_size)
    mask = (scores > score_thresh) & valid[:, None]
    inds = torch.nonzero(mask)                                    # (roi, class), row-major
    cb, cs = boxes[mask], scores[mask]
    keep = nms_ref(cb, cs, inds[:, 1], ROI_NMS_THRESH)[:DETECTIONS_PER_IMAGE]
    return dict(boxes=cb[keep], scores=cs[keep], classes=inds[keep, 1], roi=inds[keep, 0], cand_boxes=cb, cand_scores=cs, cand_inds=inds,
                keep=keep, probs=probs)


# ------------------------------------------------------------------ PointRend mask head
def point_sample(inp, coords):
    """point_rend.point_features.point_sample: inp [N,C,H,W], coords [N,P,2] in [0,1] (x, y) -> [N,C,P]."""
    return F.grid_sample(inp, 2.0 * coords.unsqueeze(2) - 1.0, mode="bilinear", padding_mode="zeros", align_corners=False).squeeze(3)


def regular_grid(R, side):
    """generate_regular_grid_point_coords: ((i + 0.5) / side) in x-fastest order."""
    v = (torch.arange(side, dtype=torch.float32) + 0.5) / side
    yy, xx = torch.meshgrid(v, v, indexing="ij")
    return torch.stack((xx.reshape(-1), yy.reshape(-1)), dim=1).unsqueeze(0).expand(R, -1, -1)


def fine_features(p2, boxes, coords, scale=0.25):
    """point_sample_fine_grained_features on one level of one image: p2 [1,C,H,W]; box-relative coords -> image -> feature-map
    normalised coordinates; -> [R, C, P]."""
    _, _, H, W = p2.shape
    pts = coords.clone()
    pts[:, :, 0] = pts[:, :, 0] * (boxes[:, None, 2] - boxes[:, None, 0]) + boxes[:, None, 0]
    pts[:, :, 1] = pts[:, :, 1] * (boxes[:, None, 3] - boxes[:, None, 1]) + boxes[:, None, 1]
    pts = pts / (torch.tensor([W, H], dtype=torch.float32) / scale)
    R, P, _ = pts.shape
    if R == 0:
        return torch.zeros(0, p2.shape[1], P)
    return point_sample(p2, pts.reshape(1, R * P, 2)).squeeze(0).transpose(0, 1).reshape(R, P, -1).transpose(1, 2)


def coarse_head(x, s, pre="roi_heads.mask_head.coarse_head"):
    """ConvFCHead: (no channel-reduction conv: 256 <= CONV_DIM 256 [3rd-party default]) 2x2 / stride 2 conv + ReLU, 2 FCs, prediction."""
    x = F.relu(F.conv2d(x, s[pre + ".reduce_spatial_dim_conv.weight"], s[pre + ".reduce_spatial_dim_conv.bias"], stride=2))
    x = x.flatten(1)
    for k in range(1, MASK_NUM_FC + 1):
        x = F.relu(F.linear(x, s[f"{pre}.fc{k}.weight"], s[f"{pre}.fc{k}.bias"]))
    return F.linear(x, s[pre + ".prediction.weight"], s[pre + ".prediction.bias"]).view(-1, NUM_CLASSES, MASK_SIDE, MASK_SIDE)


def point_head(fine, coarse, s, pre="roi_heads.mask_head.point_head"):
    """StandardPointHead (COARSE_PRED_EACH_LAYER): fine [R,256,P], coarse [R,80,P] -> [R,80,P]."""
    x = torch.cat((fine, coarse), dim=1)
    for k in range(1, POINT_NUM_FC + 1):
        x = F.relu(F.conv1d(x, s[f"{pre}.fc{k}.weight"], s[f"{pre}.fc{k}.bias"]))
        x = torch.cat((x, coarse), dim=1)
    return F.conv1d(x, s[pre + ".predictor.weight"], s[pre + ".predictor.bias"])


def mask_head(p2, boxes, classes, s, trace=None):
    """PointRendMaskHead.forward (inference) for one image: p2 [1,256,H,W], boxes [R,4] (network-input pixels), classes [R]
    -> logits of each instance's OWN class [R, 224, 224] (mask_rcnn_inference reads only that channel; the other 79 never interact
    with it: interpolation, uncertainty and scatter are per channel / per predicted class)."""
    R = len(boxes)
    res, steps = subdivision_schedule()
    if R == 0:
        return torch.zeros(0, res << steps, res << steps)
    pooled = fine_features(p2, boxes, regular_grid(R, MASK_POOLER_RES)).reshape(R, -1, MASK_POOLER_RES, MASK_POOLER_RES)
    coarse = coarse_head(pooled, s)
    ar = torch.arange(R)
    logits = None
    for step in range(steps + 1):
        if logits is None:
            coords, idx = regular_grid(R, res), None
        else:
            logits = F.interpolate(logits, scale_factor=2, mode="bilinear", align_corners=False)
            Hm, Wm = logits.shape[-2:]
            unc = -logits[ar, classes].abs().reshape(R, Hm * Wm)                  # calculate_uncertainty on the predicted class
            k = min(Hm * Wm, SUBDIVISION_NUM_POINTS)
            idx = torch.stack([topk_stable(u, k) for u in unc])
            coords = torch.stack((1.0 / (2 * Wm) + (idx % Wm).float() / Wm, 1.0 / (2 * Hm) + (idx // Wm).float() / Hm), dim=2)
        pl = point_head(fine_features(p2, boxes, coords), point_sample(coarse, coords), s)
        if logits is None:
            logits = pl.reshape(R, NUM_CLASSES, res, res)
        else:
            logits = logits.reshape(R, NUM_CLASSES, Hm * Wm).scatter_(2, idx.unsqueeze(1).expand(-1, NUM_CLASSES, -1), pl).view(R, NUM_CLASSES, Hm, Wm)
        if trace is not None:
            trace.append(dict(step=step, idx=idx, coords=coords.clone(), own=logits[ar, classes].clone(), point_logits=pl[ar, classes].clone()))
    if trace is not None:
        trace.append(dict(coarse=coarse, pooled=pooled))
    return logits[ar, classes]


def paste_masks(probs, boxes, out_h, out_w, threshold=MASK_THRESHOLD):
    """detectron2 paste_masks_in_image (GPU form: every mask sampled over the whole image): probs [R,S,S], boxes [R,4] -> bool [R,H,W]."""
    R = len(boxes)
    if R == 0:
        return torch.zeros(0, out_h, out_w, dtype=torch.bool)
    x0, y0, x1, y1 = torch.split(boxes, 1, dim=1)
    img_y = (torch.arange(0, out_h, dtype=torch.float32) + 0.5 - y0) / (y1 - y0) * 2 - 1
    img_x = (torch.arange(0, out_w, dtype=torch.float32) + 0.5 - x0) / (x1 - x0) * 2 - 1
    grid = torch.stack((img_x[:, None, :].expand(R, out_h, out_w), img_y[:, :, None].expand(R, out_h, out_w)), dim=3)
    return F.grid_sample(probs[:, None], grid, align_corners=False)[:, 0] >= threshold


# ------------------------------------------------------------------ the whole predictor
def segment(state, images_u8, score_thresh, with_masks=True, trace=None):
    """DefaultPredictor(cfg)(image)["instances"] for each image of uint8 [B,H,W,3] (one size): list of dict(pred_boxes fp32 [n,4] in
    input-image pixels, scores [n], pred_classes int64 [n], pred_masks bool [n,H,W]).  trace: dict filled with intermediates."""
    s = state
    images_u8 = np.asarray(images_u8)
    B, H, W, _ = images_u8.shape
    x, (nh, nw) = preprocess(images_u8)
    feats = fpn(resnet50(x, s), s)
    levels = [feats[f"p{l}"] for l in (2, 3, 4, 5, 6)]
    logits, deltas = rpn_head(levels, s)
    out = []
    if trace is not None:
        trace.update(x=x, feats=feats, rpn_logits=logits, rpn_deltas=deltas, per_image=[])
    for b in range(B):
        t = {}
        prop = rpn_proposals([l[b:b + 1] for l in logits], [d[b:b + 1] for d in deltas], (nh, nw))
        pooled, lv = box_pooler([feats[f"p{l}"][b] for l in (2, 3, 4, 5)], prop["boxes"])
        cls_logits, box_deltas = box_head(pooled, s)
        det = fast_rcnn_inference(cls_logits, box_deltas, prop["boxes"], (nh, nw), score_thresh)
        res = dict(scores=det["scores"], pred_classes=det["classes"])
        sx, sy = W / nw, H / nh                                          # detector_postprocess
        ob = det["boxes"].clone()
        ob[:, 0::2] *= sx
        ob[:, 1::2] *= sy
        ob = clip_boxes(ob, H, W)
        nonempty = ((ob[:, 2] - ob[:, 0]) > 0) & ((ob[:, 3] - ob[:, 1]) > 0)
        mtrace = [] if trace is not None else None
        if with_masks:
            own = mask_head(feats["p2"][b:b + 1], det["boxes"], det["classes"], s, trace=mtrace)
            res["pred_masks"] = paste_masks(own.sigmoid(), ob, H, W)[nonempty]
            t.update(mask_logits=own, mask_trace=mtrace)
        res["pred_boxes"] = ob[nonempty]
        res["scores"], res["pred_classes"] = res["scores"][nonempty], res["pred_classes"][nonempty]
        out.append(res)
        if trace is not None:
            t.update(prop=prop, pooled=pooled, roi_level=lv, cls_logits=cls_logits, box_deltas=box_deltas, det=det, out_boxes=ob, nonempty=nonempty)
            trace["per_image"].append(t)
    return out


def person_mask(instances, H, W, cat_id=0):
    """PointRendPredictor.__call__ (utils/adaptive_mask_inpainting.py:1230-1236), merge_mode "merge": np.any over the person masks."""
    m = instances["pred_masks"][instances["pred_classes"] == cat_id].numpy()
    return np.any(m, axis=0).astype(np.uint8) if len(m) else np.zeros((H, W), np.uint8)
