"""Targets the optimisation app reads off a learned ComA state, computed on MI355X.

Mirrors `src/application/optimize.py:186-196` of the reference (the only consumer of the accumulator pickle besides the
inference CLI): for every human vertex the most likely relative-orientation bin of a reference object point and its
direction, and the human vertices whose contact expectation exceeds a threshold together with their most-contacted
object point.  Index vectors follow NumPy exactly (first maximum; NaN is a maximum for argmax and poisons max).
"""
from __future__ import annotations

import pickle

import numpy as np
import torch

from . import _lib


def _dev(x, device):
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    return t.to(device=device, dtype=torch.float32).contiguous()


def row_argmax(x, n, row_stride=None, col_offset=0, rows=None, want_max=False):
    """idx[m] = np.argmax(x.reshape(rows, row_stride)[m, col_offset:col_offset+n]) on the device (int64)."""
    row_stride = n if row_stride is None else row_stride
    rows = x.numel() // row_stride if rows is None else rows
    idx = torch.empty(rows, dtype=torch.int64, device=x.device)
    val = torch.empty(rows, dtype=torch.float32, device=x.device) if want_max else None
    rc = _lib.lib().coma_row_argmax_i64(_lib.ptr(x, torch.float32, "x"), rows, n, row_stride, col_offset, _lib.ptr(idx),
                                        _lib.ptr(val), _lib.stream_ptr(x.device))
    _lib.check(rc, "coma_row_argmax_i64")
    return (idx, val) if want_max else idx


def orientation_and_contact_targets(affordance_info, reference_object_vertex_index, contact_threshold, device="cuda"):
    """affordance_info: the dict `ComA.export` pickles (or its path).  Returns the four arrays of optimize.py:190-196:
    max_prob_indices i64 [H], relative_orientation_GT [H,3] (dtype of canon_normal_grid), selected_human_indices
    (tuple of one i64 array, as np.nonzero returns), corresponding_object_indices i64 [k]."""
    if isinstance(affordance_info, (str, bytes)):
        with open(affordance_info, "rb") as handle:
            affordance_info = pickle.load(handle)
    prob = _dev(affordance_info["prob_grid_canon_human_wrt_obj"], device)              # [H,O,N]
    H, O, N = prob.shape
    o = int(reference_object_vertex_index)
    if not -O <= o < O:
        raise IndexError(f"index {o} is out of bounds for axis 1 with size {O}")
    o %= O
    max_prob_indices = row_argmax(prob, N, row_stride=O * N, col_offset=o * N, rows=H).cpu().numpy()
    relative_orientation_GT = np.asarray(affordance_info["canon_normal_grid"])[max_prob_indices].reshape(H, 3)
    nom = _dev(affordance_info["contact_dist_expectation_grid_nom"], device)
    den = _dev(affordance_info["contact_dist_expectation_grid_denom"], device)
    Hc, Oc = nom.shape
    sel = torch.empty(Hc, dtype=torch.uint8, device=nom.device)
    rc = _lib.lib().coma_contact_select_u8(_lib.ptr(nom), _lib.ptr(den), Hc, Oc, float(contact_threshold), _lib.ptr(sel),
                                           _lib.stream_ptr(nom.device))
    _lib.check(rc, "coma_contact_select_u8")
    obj = row_argmax(nom, Oc)
    selected = torch.nonzero(sel, as_tuple=True)[0]
    return (max_prob_indices, relative_orientation_GT, (selected.cpu().numpy(),), obj[selected].cpu().numpy())
