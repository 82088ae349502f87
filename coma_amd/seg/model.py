"""The person-segmentation network as ONE static launch plan on the device: what `DefaultPredictor(cfg)(image)` does
for the reference's PointRend config (utils/adaptive_mask_inpainting.py:1190-1236, src/generation/segment_human.py:43-55) -- resize, ResNet-50
+ FPN, RPN, box head, per-class NMS, PointRend's coarse head + point head with subdivision inference, mask paste -- for a batch of B
same-sized uint8 images, in fp32 (the reference's precision), with no host synchronisation inside: every data-dependent size (proposals
kept, detections) stays on the device as a count the next launch reads, so the whole forward is one hipGraph replay.

detectron2 is absent from this image, so the architecture follows the reference's yaml files plus detectron2's published defaults (listed
in oracle/seg_oracle.py, which is what the tests compare this plan with: parity unpinned).  Static shapes: 1000 proposals (RPN
POST_NMS_TOPK_TEST) and 100 detections (TEST.DETECTIONS_PER_IMAGE) per image are the capacities; the candidate lists hold 8192 entries
(5 x 1000 RPN candidates; detections need score_thresh >= 1 / 8: at most floor(1 / t) classes of one ROI can pass a softmax threshold t).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from ..sd.graph import LaunchGraph
from . import ops
from . import weights as W

F32, I32, I64, U8 = torch.float32, torch.int32, torch.int64, torch.uint8
_DEBUG_SYNC = __import__("os").environ.get("SEG_DEBUG_SYNC") == "1"
PIXEL_MEAN = (103.530, 116.280, 123.675)            # MODEL.PIXEL_MEAN [3rd-party default]; applied to the channels IN THE ORDER GIVEN
MIN_SIZE, MAX_SIZE, DIVIS = 800, 1333, 32
ANCHOR_SIZES, ANCHOR_RATIOS = (32, 64, 128, 256, 512), (0.5, 1.0, 2.0)
PRE_TOPK, POST_TOPK, RPN_NMS = 1000, 1000, 0.7
DET_MAX, DET_NMS = 100, 0.5
CAP = 8192
SPLITK_WS_FLOATS = 16 << 20                          # 64 MiB: 256 partial tiles of 128 x 128 x up to 4 slices ... 512 slots' worth
POINTS = 28 * 28                                     # POINT_HEAD.SUBDIVISION_NUM_POINTS
INIT_RES, SUBDIV_STEPS = 28, 3                       # 7 -> 28 and 5 -> 3 by PointRendMaskHead._init_point_head's doubling rule


def shortest_edge_size(h, w, short=MIN_SIZE, max_size=MAX_SIZE):
    scale = short * 1.0 / min(h, w)
    newh, neww = (short, scale * w) if h < w else (scale * h, short)
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def bilinear_tables(in_size, out_size):
    """Pillow's coefficient tables for a BILINEAR resize of one axis (Resample.c precompute_coeffs + normalize_coeffs_8bpc; triangle filter,
    support max(scale, 1)): bounds int32 [out, 2] = (first source index, taps), kk int32 [out, ksize] = weights in 22-bit fixed point."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = fscale
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)           # int() truncation of a value >= -0.5 ...
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)                      # ... and the clamp at 0
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    j = np.arange(ksize)[None, :]
    arg = np.abs((j + xmin[:, None] - center[:, None] + 0.5) / fscale)
    w = np.where((arg < 1.0) & (j < xmax[:, None]), 1.0 - arg, 0.0)
    tot = w.sum(axis=1, keepdims=True)
    w = np.where(tot != 0.0, w / np.where(tot == 0.0, 1.0, tot), w)
    kk = np.where(w < 0, (-0.5 + w * (1 << 22)).astype(np.int64), (0.5 + w * (1 << 22)).astype(np.int64)).astype(np.int32)
    return np.stack([xmin, xmax], axis=1).astype(np.int32), kk


def cell_anchors(size):
    a = []
    for r in ANCHOR_RATIOS:
        w = math.sqrt(size * size / r)
        h = r * w
        a.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(a, dtype=torch.float32)


class HipPointRend:
    """plan = HipPointRend(state, batch, height, width, device, score_thresh); plan(images_u8 [B,H,W,3] device tensor) -> dict of DEVICE tensors:
    count i32 [B], boxes f32 [B,100,4] (input-image pixels), scores f32 [B,100], classes i32 [B,100], valid i32 [B,100] (detector_postprocess's
    non-empty filter), masks u8 [B,100,H,W] (keep_masks), person u8 [B,H,W] (np.any over the masks of cat_id: the plug-in's output)."""

    def __init__(self, state, batch, height, width, device="cuda", score_thresh=0.2, keep_masks=True, use_graph=True, cat_id=0, stage="masks",
                 debug=False, detections_per_image=DET_MAX):
        """detections_per_image: TEST.DETECTIONS_PER_IMAGE (100 [3rd-party default]) -- the capacity of the mask head's buffers as well."""
        if score_thresh < 1.0 / 8:
            raise ValueError(f"score threshold {score_thresh} < 0.125: a ROI could put more than 8 classes on the candidate list ({CAP} slots for 1000 ROIs)")
        assert stage in ("boxes", "masks")
        self.device = torch.device(device)
        self.B, self.H, self.W = batch, height, width
        self.score_thresh, self.use_graph, self.keep_masks, self.cat_id, self.stage, self.debug = score_thresh, use_graph, keep_masks, cat_id, stage, debug
        self.D = int(detections_per_image)
        assert 1 <= self.D <= DET_MAX
        self.nh, self.nw = shortest_edge_size(height, width)
        self.hp, self.wp = -(-self.nh // DIVIS) * DIVIS, -(-self.nw // DIVIS) * DIVIS
        self.P = {k: (w.to(self.device).contiguous(), b.to(self.device).contiguous()) for k, (w, b) in W.prepare(state).items()}
        self.g = LaunchGraph(self.device, plan="segment")
        self.t = {}                      # named intermediates (tests, debugging)
        self._build()

    # ---- recording helpers
    def _const(self, arr, dtype):
        t = torch.as_tensor(np.ascontiguousarray(arr)).to(self.device, dtype).contiguous()
        self.g.model.register(t)         # PERSISTENT: a constant table
        return t

    def _conv(self, x, name, *, batch, h, w, c, kh=1, stride=1, pad=0, relu=False, res=None, res_mode=0, ldo=0, ldx=0, out=None, m_dev=None,
              rows_per_item=1, unit_rows=0, tag=None, k_alg=None):
        """k_alg: the layer's own K where `c` includes zero padding (flop count and tag follow the layer, not the padded buffer)."""
        wt, bias = self.P[name]
        n = wt.shape[0]
        oh, ow = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kh) // stride + 1
        if out is None:
            out = self.g.buf(batch * oh * ow, ldo or n, dtype=F32)
        M = batch * oh * ow
        self.g.add(lambda: ops.conv_gemm(x, wt, out, batch=batch, in_h=h, in_w=w, c=c, n=n, kh=kh, kw=kh, stride=stride, pad=pad, out_h=oh, out_w=ow,
                                         bias=bias, res=res, res_mode=res_mode, ldo=ldo, ldx=ldx, relu=relu, m_dev=m_dev,
                                         rows_per_item=rows_per_item, unit_rows=unit_rows, workspace=self.splitk_ws),
                   flops=2 * M * n * (k_alg or kh * kh * c), nbytes=4 * (batch * h * w * c + n * kh * kh * c + M * n * (2 if res is not None else 1)),
                   tag=tag or f"seg gemm {name} M={M} N={n} K={k_alg or kh * kh * c}")
        return out, oh, ow

    def _build(self):
        g, B, dev = self.g, self.B, self.device
        H, Wd, nh, nw, hp, wp = self.H, self.W, self.nh, self.nw, self.hp, self.wp
        t = self.t
        # split-K scratch of the GEMMs whose tiles leave most of the chip idle (deep levels, the mask head's fully connected layers): one
        # buffer, the launches of a plan run in order on one stream
        self.splitk_ws = g.buf(SPLITK_WS_FLOATS, dtype=F32)
        # ---- DefaultPredictor: ResizeShortestEdge (PIL bilinear) + preprocess_image
        self.images = g.buf(B, H, Wd, 3, dtype=U8)
        bx, kx = bilinear_tables(Wd, nw)
        by, ky = bilinear_tables(H, nh)
        bx, kx, by, ky = (self._const(a, I32) for a in (bx, kx, by, ky))
        tmp = g.buf(B, H, nw, 3, dtype=U8)
        x0 = g.buf(B, hp, wp, 4, dtype=F32)
        t["resized"] = g.buf(B, nh, nw, 3, dtype=U8) if self.debug else None
        g.add(lambda: ops.resize_normalize(self.images, tmp, x0, batch=B, h=H, w=Wd, new_h=nh, new_w=nw, pad_h=hp, pad_w=wp, bounds_x=bx, kk_x=kx,
                                           bounds_y=by, kk_y=ky, mean=PIXEL_MEAN, resized=t["resized"]),
              tag="seg resize + normalise", nbytes=B * (3 * H * Wd + 6 * H * nw + 16 * hp * wp))
        t["x0"] = x0
        # ---- ResNet-50
        x, h, w = self._conv(x0, "stem", batch=B, h=hp, w=wp, c=4, kh=7, stride=2, pad=3, relu=True)
        ph, pw = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        pooled = g.buf(B * ph * pw, 64, dtype=F32)
        g.add(lambda x=x, h=h, w=w: ops.maxpool3x3s2(x, pooled, batch=B, h=h, w=w, c=64), tag="seg maxpool", nbytes=4 * B * (h * w + ph * pw) * 64)
        x, h, w, cin = pooled, ph, pw, 64
        res = {}
        for i, (nblk, width) in enumerate(zip(W.RES_BLOCKS, W.RES_WIDTH)):
            for j in range(nblk):
                p = f"res{i + 2}.{j}"
                stride = 2 if (j == 0 and i > 0) else 1
                if f"{p}.shortcut" in self.P:
                    sc, _, _ = self._conv(x, f"{p}.shortcut", batch=B, h=h, w=w, c=cin, stride=stride)
                else:
                    sc = x
                y, h2, w2 = self._conv(x, f"{p}.conv1", batch=B, h=h, w=w, c=cin, stride=stride, relu=True)        # STRIDE_IN_1X1
                y, _, _ = self._conv(y, f"{p}.conv2", batch=B, h=h2, w=w2, c=width, kh=3, pad=1, relu=True)
                x, _, _ = self._conv(y, f"{p}.conv3", batch=B, h=h2, w=w2, c=width, relu=True, res=sc, res_mode=1)
                h, w, cin = h2, w2, 4 * width
            res[i + 2] = (x, h, w, cin)
        # ---- FPN (top-down: the coarser level is the residual read at (oy >> 1, ox >> 1))
        feats, prev = {}, None
        for lvl in (5, 4, 3, 2):
            x, h, w, cin = res[lvl]
            lat, _, _ = self._conv(x, f"fpn_lateral{lvl}", batch=B, h=h, w=w, c=cin, res=prev, res_mode=2 if prev is not None else 0)
            out, _, _ = self._conv(lat, f"fpn_output{lvl}", batch=B, h=h, w=w, c=256, kh=3, pad=1)
            feats[lvl] = (out, h, w)
            prev = lat
        p5, h5, w5 = feats[5]
        h6, w6 = (h5 - 1) // 2 + 1, (w5 - 1) // 2 + 1
        p6 = g.buf(B * h6 * w6, 256, dtype=F32)
        g.add(lambda: ops.subsample2(p5, p6, batch=B, h=h5, w=w5, c=256), tag="seg p6")
        feats[6] = (p6, h6, w6)
        for lvl in feats:
            t[f"p{lvl}"] = feats[lvl][0]
        self.feat_dims = {lvl: feats[lvl][1:] for lvl in feats}
        # ---- RPN: head per level, top-k + decode, one sort, one NMS
        ck, cb, cg = g.buf(B, CAP, dtype=I64), g.buf(B, CAP, 4, dtype=F32), g.buf(B, CAP, dtype=I32)
        g.add(lambda: ops.memset(ck, 0xFF), tag="seg memset keys")
        off = 0
        t["rpn_pred"] = {}
        preds, cells, dims = [], [], []
        for li, lvl in enumerate((2, 3, 4, 5, 6)):
            f, fh, fw = feats[lvl]
            hid, _, _ = self._conv(f, "rpn_conv", batch=B, h=fh, w=fw, c=256, kh=3, pad=1, relu=True)
            pred, _, _ = self._conv(hid, "rpn_pred", batch=B, h=fh, w=fw, c=256, ldo=16)
            t["rpn_pred"][lvl] = pred
            preds.append(pred)
            cells.append(self._const(cell_anchors(ANCHOR_SIZES[li]).numpy(), F32))
            dims.append((fh, fw))
            off += min(fh * fw * 3, PRE_TOPK)
        # one launch for the five levels (a level is one workgroup per image: side by side they take as long as p2 alone)
        kbuf = g.buf(B, sum(fh * fw * 3 for fh, fw in dims), dtype=I32)          # the objectness keys, dense (the selection's five passes read them coalesced)
        g.add(lambda: ops.rpn_select_levels(preds, cells, ck, cb, cg, ld=16, batch=B, dims=dims, first_stride=4, pre_topk=PRE_TOPK, img_h=nh, img_w=nw,
                                            cap=CAP, key_scratch=kbuf), tag="seg rpn select p2..p6")
        self.n_rpn_cand = off
        sb, ss, sg, ssrc, nv = g.buf(B, CAP, 4, dtype=F32), g.buf(B, CAP, dtype=F32), g.buf(B, CAP, dtype=I32), g.buf(B, CAP, dtype=I32), g.buf(B, dtype=I32)
        mask_ws = g.buf(B, CAP, CAP // 64, dtype=I64)
        g.add(lambda: ops.sort_candidates(ck, cb, cg, sb, ss, sg, ssrc, nv, batch=B, cap=CAP), tag="seg sort (rpn)")
        R = POST_TOPK
        prop, prop_sc, prop_lv, prop_src = g.buf(B, R, 4, dtype=F32), g.buf(B, R, dtype=F32), g.buf(B, R, dtype=I32), g.buf(B, R, dtype=I32)
        prop_pos, prop_n = g.buf(B, R, dtype=I32), g.buf(B, dtype=I32)
        g.add(lambda: ops.nms(sb, ss, sg, ssrc, nv, mask_ws, prop_pos, prop, prop_sc, prop_lv, prop_src, prop_n, batch=B, cap=CAP, thresh=RPN_NMS,
                              max_keep=R), tag="seg nms (rpn)")
        t.update(cand_keys=ck, cand_boxes=cb, cand_group=cg, sorted_boxes=sb, sorted_scores=ss, sorted_group=sg, sorted_src=ssrc, n_valid=nv,
                 proposals=prop, prop_scores=prop_sc, prop_level=prop_lv, prop_src=prop_src, prop_count=prop_n, prop_pos=prop_pos)
        # ---- box head
        p2, h2, w2 = feats[2]
        roi = g.buf(B * R, 49 * 256, dtype=F32)
        lv = g.buf(B * R, dtype=I32)
        g.add(lambda: ops.roi_align([feats[l][0] for l in (2, 3, 4, 5)], prop, prop_n, roi, lv, h2=h2, w2=w2, c=256, batch=B, R=R, out_size=7),
              tag="seg roi_align", nbytes=4 * B * R * 49 * 256 * 5)
        f1, _, _ = self._conv(roi, "box_fc1", batch=B * R, h=1, w=1, c=49 * 256, relu=True)
        f2, _, _ = self._conv(f1, "box_fc2", batch=B * R, h=1, w=1, c=1024, relu=True)
        bp, _, _ = self._conv(f2, "box_pred", batch=B * R, h=1, w=1, c=1024, ldo=404)
        t.update(roi_feat=roi, roi_level=lv, box_pred=bp)
        dk, db, dg, dn = g.buf(B, CAP, dtype=I64), g.buf(B, CAP, 4, dtype=F32), g.buf(B, CAP, dtype=I32), g.buf(B, dtype=I32)
        probs = g.buf(B * R, 81, dtype=F32) if self.debug else None
        g.add(lambda: (ops.memset(dk, 0xFF), ops.memset(dn, 0)), tag="seg memset det")
        g.add(lambda: ops.box_predict(bp, prop, prop_n, dk, db, dg, dn, probs, ld=404, batch=B, R=R, img_h=nh, img_w=nw, score_thresh=self.score_thresh,
                                      cap=CAP), tag="seg box predict")
        dsb, dss, dsg, dssrc, dnv = g.buf(B, CAP, 4, dtype=F32), g.buf(B, CAP, dtype=F32), g.buf(B, CAP, dtype=I32), g.buf(B, CAP, dtype=I32), g.buf(B, dtype=I32)
        g.add(lambda: ops.sort_candidates(dk, db, dg, dsb, dss, dsg, dssrc, dnv, batch=B, cap=CAP), tag="seg sort (det)")
        D = self.D
        det, det_sc, det_cls, det_src = g.buf(B, D, 4, dtype=F32), g.buf(B, D, dtype=F32), g.buf(B, D, dtype=I32), g.buf(B, D, dtype=I32)
        det_pos, det_n = g.buf(B, D, dtype=I32), g.buf(B, dtype=I32)
        g.add(lambda: ops.nms(dsb, dss, dsg, dssrc, dnv, mask_ws, det_pos, det, det_sc, det_cls, det_src, det_n, batch=B, cap=CAP, thresh=DET_NMS,
                              max_keep=D), tag="seg nms (det)")
        ob, valid = g.buf(B, D, 4, dtype=F32), g.buf(B, D, dtype=I32)
        g.add(lambda: ops.finalize_detections(det, det_n, ob, valid, batch=B, R=D, img_h=nh, img_w=nw, out_h=H, out_w=Wd), tag="seg finalize")
        t.update(det_cand_keys=dk, det_cand_boxes=db, det_cand_group=dg, det_cand_count=dn, det_sorted_boxes=dsb, det_sorted_scores=dss,
                 det_sorted_group=dsg, det_sorted_src=dssrc, det_n_valid=dnv, det_boxes=det, det_scores=det_sc, det_classes=det_cls, det_src=det_src,
                 det_count=det_n, out_boxes=ob, valid=valid, probs=probs, det_pos=det_pos)
        self.out = dict(count=det_n, boxes=ob, scores=det_sc, classes=det_cls, valid=valid, net_boxes=det)
        if self.stage == "boxes":
            return
        # ---- PointRend mask head: coarse head on a 14 x 14 point-sampled grid of p2 ...
        NR = B * D
        grid14 = g.buf(NR * 196, 256, dtype=F32)
        g.add(lambda: ops.point_sample(p2, grid14, fh=h2, fw=w2, c=256, per_roi=False, feat_scale=0.25, boxes=det, count=det_n, batch=B, R=D, P=196,
                                       grid_side=14, ldo=256), tag="seg mask pooler (14 x 14 point grid)")
        gate = dict(m_dev=det_n)
        c1, _, _ = self._conv(grid14, "coarse_conv", batch=NR, h=14, w=14, c=256, kh=2, stride=2, relu=True, rows_per_item=49, unit_rows=D * 49, **gate)
        c2, _, _ = self._conv(c1, "coarse_fc1", batch=NR, h=1, w=1, c=49 * 256, relu=True, rows_per_item=1, unit_rows=D, **gate)
        c3, _, _ = self._conv(c2, "coarse_fc2", batch=NR, h=1, w=1, c=1024, relu=True, rows_per_item=1, unit_rows=D, **gate)
        coarse, _, _ = self._conv(c3, "coarse_pred", batch=NR, h=1, w=1, c=1024, rows_per_item=1, unit_rows=D, **gate)      # [NR][7][7][80]
        t.update(grid14=grid14, coarse=coarse)
        # ... then the point head on 784 points per instance and step: a regular 28 x 28 grid, then the most uncertain points of the x2 map
        NP = NR * POINTS
        # point-head inputs [fine 256 | coarse 80 | 16 zero columns]: rows of 352 floats, so that a 32-wide K chunk never straddles the end of
        # a row and the GEMM takes its wave-uniform addressing path (weights are zero-padded to K = 352 by weights._pad_k; the pad columns
        # are zeroed once, nothing writes them)
        XW = 352
        X = g.buf(4, NP, XW, dtype=F32, zero=True)
        wp_, bp_ = self.P["point_pred"]
        idx, coords = g.buf(NR, POINTS, dtype=I32), g.buf(NR, POINTS, 2, dtype=F32)
        s, cur = INIT_RES, None
        t["maps"], t["idx"] = [], []
        for step in range(SUBDIV_STEPS + 1):
            if step > 0:
                s *= 2
                nxt = g.buf(NR, s, s, dtype=F32)
                g.add(lambda cur=cur, nxt=nxt, s=s: ops.upsample2x(cur, det_n, nxt, batch=B, R=D, s=s // 2), tag=f"seg upsample -> {s}")
                g.add(lambda nxt=nxt, s=s: ops.topk_points(nxt, det_n, idx, coords, batch=B, R=D, s=s, k=POINTS), tag=f"seg uncertain points {s}")
                cur = nxt
            else:
                cur = g.buf(NR, s, s, dtype=F32)
            cd = coords if step > 0 else None
            g.add(lambda cd=cd: ops.point_sample(p2, X[0], fh=h2, fw=w2, c=256, per_roi=False, feat_scale=0.25, boxes=det, count=det_n, batch=B, R=D,
                                                 coords=cd, P=POINTS, grid_side=INIT_RES, ldo=XW), tag="seg point features (p2)")
            g.add(lambda cd=cd: ops.point_sample(coarse, X[0], fh=7, fw=7, c=80, per_roi=True, count=det_n, batch=B, R=D, coords=cd, P=POINTS,
                                                 grid_side=INIT_RES, ldo=XW, col0=256, n_copies=4, copy_stride=NP * XW), tag="seg point features (coarse)")
            for k in (1, 2, 3):
                self._conv(X[k - 1], f"point_fc{k}", batch=NP, h=1, w=1, c=XW, k_alg=336, relu=True, out=X[k], ldo=XW, rows_per_item=POINTS,
                           unit_rows=D * POINTS, **gate)
            ix = idx if step > 0 else None
            g.add(lambda cur=cur, ix=ix, s=s: ops.point_logit_scatter(X[3], wp_, bp_, det_cls, det_n, cur, ix, ldx=XW, kdim=336, batch=B, R=D, P=POINTS,
                                                                       s=s), tag=f"seg point logits -> {s}")
            if self.debug:
                keep, kidx = g.buf(NR, s, s, dtype=F32), g.buf(NR, POINTS, dtype=I32)
                g.add(lambda cur=cur, keep=keep, kidx=kidx: (ops_copy(keep, cur), ops_copy(kidx, idx)), tag="seg debug copy")
                t["maps"].append(keep)
                t["idx"].append(kidx)
        self.mask_side = s
        merged = g.buf(B, H, Wd, dtype=U8)
        masks = g.buf(B, D, H, Wd, dtype=U8) if self.keep_masks else None
        g.add(lambda: ops.paste_masks(cur, ob, valid, det_cls, det_n, merged, masks, s=s, batch=B, R=D, out_h=H, out_w=Wd, cat_id=self.cat_id),
              tag="seg paste + merge")
        t.update(mask_logits=cur, X=X)
        self.out.update(masks=masks, person=merged)

    # ---- execution
    def __call__(self, images_u8):
        assert tuple(images_u8.shape) == (self.B, self.H, self.W, 3) and images_u8.dtype == U8, (images_u8.shape, images_u8.dtype)
        self.images.copy_(images_u8.to(self.device), non_blocking=True)
        if _DEBUG_SYNC:                  # SEG_DEBUG_SYNC=1: eager, one synchronisation and one stderr line per launch (locating a faulting launch)
            import sys
            for fn, (tag, _) in zip(self.g.launches, self.g.tags):
                print("seg ->", tag, file=sys.stderr, flush=True)
                fn()
                torch.cuda.synchronize(self.device)
            return self.out
        if self.use_graph:
            self.g.replay()
        else:
            self.g.run()
        return self.out

    def instances(self, b=0):
        """Host copy of image b's detections in the reference's record layout (src/generation/segment_human.py:152-166)."""
        o = self.out
        n = int(o["count"][b])
        ok = o["valid"][b, :n].bool().cpu().numpy()
        rec = dict(pred_boxes=o["boxes"][b, :n].cpu().numpy()[ok], scores=o["scores"][b, :n].cpu().numpy()[ok],
                   pred_classes=o["classes"][b, :n].cpu().numpy().astype(np.int64)[ok])
        if o.get("masks") is not None:
            rec["pred_masks"] = o["masks"][b, :n].cpu().numpy().astype(bool)[ok]
        return rec


def ops_copy(dst, src):
    from ..sd import ops as sd_ops
    sd_ops.copy_d2d(dst.view(-1), src.view(-1))
