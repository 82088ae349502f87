"""Thin Python bindings of the seg_* C ABI (include/seg_hip.h).  Activations are NHWC fp32 tensors on a HIP device; every function launches on
torch's current stream (or is recorded into the plan this thread is recording) and raises ComaHipError on failure.  No CPU fallback."""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from ..sd.ops import _p, _stream

F32, I32, U8, I64 = torch.float32, torch.int32, torch.uint8, torch.int64


class SegConvDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("batch", C.c_int), ("in_h", C.c_int), ("in_w", C.c_int), ("c", C.c_int), ("ldx", C.c_int),
                ("w", C.c_void_p), ("n", C.c_int), ("kpad", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
                ("out_h", C.c_int), ("out_w", C.c_int), ("bias", C.c_void_p), ("res", C.c_void_p), ("ldr", C.c_int), ("res_mode", C.c_int),
                ("out", C.c_void_p), ("ldo", C.c_int), ("relu", C.c_int), ("m_dev", C.c_void_p), ("rows_per_item", C.c_int), ("unit_rows", C.c_int), ("tile", C.c_int),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("split_k", C.c_int)]


def conv_gemm(x, w, out, *, batch, in_h, in_w, c, n, kh=1, kw=1, stride=1, pad=0, out_h=None, out_w=None, bias=None, res=None, res_mode=0,
              ldr=0, ldo=0, ldx=0, relu=False, m_dev=None, rows_per_item=1, unit_rows=0, tile=0, workspace=None, split_k=0):
    d = SegConvDesc()
    d.x, d.batch, d.in_h, d.in_w, d.c, d.ldx = _p(x, "x", F32), batch, in_h, in_w, c, ldx
    d.w, d.n, d.kpad = _p(w, "w", F32), n, w.shape[-1]
    d.kh, d.kw, d.stride, d.pad = kh, kw, stride, pad
    d.out_h = out_h if out_h is not None else (in_h + 2 * pad - kh) // stride + 1
    d.out_w = out_w if out_w is not None else (in_w + 2 * pad - kw) // stride + 1
    d.bias, d.res, d.ldr, d.res_mode = _p(bias, "bias", F32), _p(res, "res", F32), ldr, res_mode
    d.out, d.ldo, d.relu = _p(out, "out", F32), ldo, 1 if relu else 0
    d.m_dev, d.rows_per_item, d.unit_rows, d.tile = _p(m_dev, "m_dev", I32), rows_per_item, unit_rows, tile
    d.workspace, d.workspace_bytes, d.split_k = _p(workspace, "workspace", F32), (workspace.numel() * 4 if workspace is not None else 0), split_k
    _lib.check(_lib.lib().seg_conv_gemm_f32(C.byref(d), _stream(out)), "seg_conv_gemm_f32")
    return out


def resize_normalize(src, tmp, out, *, batch, h, w, new_h, new_w, pad_h, pad_w, bounds_x, kk_x, bounds_y, kk_y, mean, resized=None):
    rc = _lib.lib().seg_resize_normalize_u8(_p(src, "src", U8), batch, h, w, new_h, new_w, pad_h, pad_w, _p(bounds_x, "bounds_x", I32),
                                            _p(kk_x, "kk_x", I32), kk_x.shape[1], _p(bounds_y, "bounds_y", I32), _p(kk_y, "kk_y", I32), kk_y.shape[1],
                                            float(mean[0]), float(mean[1]), float(mean[2]), _p(tmp, "tmp", U8), _p(resized, "resized", U8),
                                            _p(out, "out", F32), _stream(out))
    _lib.check(rc, "seg_resize_normalize_u8")
    return out


def maxpool3x3s2(x, out, *, batch, h, w, c):
    _lib.check(_lib.lib().seg_maxpool3x3s2_f32(_p(x, "x", F32), batch, h, w, c, _p(out, "out", F32), _stream(out)), "seg_maxpool3x3s2_f32")
    return out


def subsample2(x, out, *, batch, h, w, c):
    _lib.check(_lib.lib().seg_subsample2_f32(_p(x, "x", F32), batch, h, w, c, _p(out, "out", F32), _stream(out)), "seg_subsample2_f32")
    return out


def memset(t, byte):
    _lib.check(_lib.lib().seg_memset(_p(t, "dst", t.dtype), byte, t.numel() * t.element_size(), _stream(t)), "seg_memset")
    return t


def rpn_select(pred, cell, keys, boxes, group, *, ld, batch, fh, fw, stride, level, anchor_base, pre_topk, img_h, img_w, cand_offset, cap):
    rc = _lib.lib().seg_rpn_select(_p(pred, "pred", F32), ld, batch, fh, fw, stride, _p(cell, "cell_anchors", F32), level, anchor_base, pre_topk,
                                   float(img_h), float(img_w), cand_offset, cap, _p(keys, "keys", I64), _p(boxes, "boxes", F32), _p(group, "group", I32),
                                   _stream(pred))
    _lib.check(rc, "seg_rpn_select")


def rpn_select_levels(preds, cells, keys, boxes, group, *, ld, batch, dims, first_stride, pre_topk, img_h, img_w, cap, key_scratch=None):
    """seg_rpn_select for consecutive FPN levels in one launch; dims: [(fh, fw), ...] per level."""
    n = len(preds)
    pa = (C.c_void_p * n)(*[_p(t, "pred", F32) for t in preds])
    ca = (C.c_void_p * n)(*[_p(t, "cell_anchors", F32) for t in cells])
    fh = (C.c_int * n)(*[d[0] for d in dims])
    fw = (C.c_int * n)(*[d[1] for d in dims])
    rc = _lib.lib().seg_rpn_select_levels(pa, ca, fh, fw, n, first_stride, ld, batch, pre_topk, float(img_h), float(img_w), cap, _p(keys, "keys", I64),
                                          _p(boxes, "boxes", F32), _p(group, "group", I32), _p(key_scratch, "key_scratch", I32), _stream(preds[0]))
    _lib.check(rc, "seg_rpn_select_levels")


def sort_candidates(keys, boxes, group, s_boxes, s_scores, s_group, s_src, n_valid, *, batch, cap):
    rc = _lib.lib().seg_sort_candidates(_p(keys, "keys", I64), _p(boxes, "boxes", F32), _p(group, "group", I32), batch, cap, _p(s_boxes, "s_boxes", F32),
                                        _p(s_scores, "s_scores", F32), _p(s_group, "s_group", I32), _p(s_src, "s_src", I32), _p(n_valid, "n_valid", I32),
                                        _stream(keys))
    _lib.check(rc, "seg_sort_candidates")


def nms(s_boxes, s_scores, s_group, s_src, n_valid, mask_ws, keep_pos, out_boxes, out_scores, out_group, out_src, out_count, *, batch, cap, thresh,
        max_keep):
    rc = _lib.lib().seg_nms(_p(s_boxes, "s_boxes", F32), _p(s_scores, "s_scores", F32), _p(s_group, "s_group", I32), _p(s_src, "s_src", I32),
                            _p(n_valid, "n_valid", I32), batch, cap, float(thresh), max_keep, _p(mask_ws, "mask_ws", I64), _p(keep_pos, "keep_pos", I32),
                            _p(out_boxes, "out_boxes", F32), _p(out_scores, "out_scores", F32), _p(out_group, "out_group", I32),
                            _p(out_src, "out_src", I32), _p(out_count, "out_count", I32), _stream(s_boxes))
    _lib.check(rc, "seg_nms")


def roi_align(feats, boxes, count, out, level=None, *, h2, w2, c, batch, R, out_size):
    rc = _lib.lib().seg_roi_align_f32(*[_p(f, f"p{l + 2}", F32) for l, f in enumerate(feats)], h2, w2, c, _p(boxes, "boxes", F32), _p(count, "count", I32),
                                      batch, R, out_size, _p(out, "out", F32), _p(level, "level", I32), _stream(out))
    _lib.check(rc, "seg_roi_align_f32")
    return out


def box_predict(pred, proposals, count, keys, boxes, group, cand_count, probs=None, *, ld, batch, R, img_h, img_w, score_thresh, cap):
    rc = _lib.lib().seg_box_predict(_p(pred, "pred", F32), ld, _p(proposals, "proposals", F32), _p(count, "count", I32), batch, R, float(img_h),
                                    float(img_w), float(score_thresh), cap, _p(keys, "keys", I64), _p(boxes, "boxes", F32), _p(group, "group", I32),
                                    _p(cand_count, "cand_count", I32), _p(probs, "probs", F32), _stream(pred))
    _lib.check(rc, "seg_box_predict")


def finalize_detections(det_boxes, count, out_boxes, valid, *, batch, R, img_h, img_w, out_h, out_w):
    rc = _lib.lib().seg_finalize_detections(_p(det_boxes, "det_boxes", F32), _p(count, "count", I32), batch, R, float(img_h), float(img_w), out_h, out_w,
                                            _p(out_boxes, "out_boxes", F32), _p(valid, "valid", I32), _stream(det_boxes))
    _lib.check(rc, "seg_finalize_detections")


def point_sample(feat, out, *, fh, fw, c, per_roi, feat_scale=1.0, boxes=None, count, batch, R, coords=None, P, grid_side=0, ldo, col0=0, n_copies=1,
                 copy_stride=0):
    rc = _lib.lib().seg_point_sample_f32(_p(feat, "feat", F32), fh, fw, c, 1 if per_roi else 0, float(feat_scale), _p(boxes, "boxes", F32),
                                         _p(count, "count", I32), batch, R, _p(coords, "coords", F32), P, grid_side, _p(out, "out", F32), ldo, col0,
                                         n_copies, copy_stride, _stream(out))
    _lib.check(rc, "seg_point_sample_f32")
    return out


def upsample2x(x, count, out, *, batch, R, s):
    _lib.check(_lib.lib().seg_upsample2x_f32(_p(x, "x", F32), _p(count, "count", I32), batch, R, s, _p(out, "out", F32), _stream(out)), "seg_upsample2x_f32")
    return out


def topk_points(logits, count, idx, coords, *, batch, R, s, k):
    rc = _lib.lib().seg_topk_points(_p(logits, "logits", F32), _p(count, "count", I32), batch, R, s, k, _p(idx, "idx", I32), _p(coords, "coords", F32),
                                    _stream(logits))
    _lib.check(rc, "seg_topk_points")


def point_logit_scatter(x, w, bias, classes, count, logit_map, idx=None, *, ldx, kdim, batch, R, P, s):
    rc = _lib.lib().seg_point_logit_scatter(_p(x, "x", F32), ldx, kdim, _p(w, "w", F32), _p(bias, "bias", F32), _p(classes, "classes", I32),
                                            _p(count, "count", I32), batch, R, P, _p(idx, "idx", I32), _p(logit_map, "map", F32), s, _stream(x))
    _lib.check(rc, "seg_point_logit_scatter")


def paste_masks(logits, out_boxes, valid, classes, count, merged, masks=None, *, s, batch, R, out_h, out_w, cat_id=0):
    rc = _lib.lib().seg_paste_masks(_p(logits, "logits", F32), s, _p(out_boxes, "out_boxes", F32), _p(valid, "valid", I32), _p(classes, "classes", I32),
                                    _p(count, "count", I32), batch, R, out_h, out_w, cat_id, _p(masks, "masks", U8), _p(merged, "merged", U8),
                                    _stream(logits))
    _lib.check(rc, "seg_paste_masks")
