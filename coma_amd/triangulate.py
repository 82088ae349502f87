"""Multi-view lifting helpers on MI355X: batched two-view DLT triangulation and the RANSAC inlier search.

Host mirror of `compute_ransac_inclusives_with_triangulation` (/root/reference/src/generation/optimize_depth.py:143-368),
the only other data-parallel numeric kernel of the 2D -> 3D lifting stage: same arguments, same file
layout (camera pickles, human-prediction pickles, sentinel strings), same return value (list of inlier dicts sorted by
reprojection error).  The per-pair pseudo-inverses, the candidate scores and the candidates^2 reprojection matrix run in
`coma_dlt_score_f64` / `coma_ransac_mse_f64` (coma_amd/csrc/triangulate.hip) in f64; selection and ordering follow the
reference's Python (`sorted` is stable, the FIRST candidate with the strictly largest inlier count wins).
No CPU fallback: tensors must live on a HIP device.
"""
from __future__ import annotations

import pickle
from glob import glob

import numpy as np
import torch

from . import _lib

# constants/generation/visualizers.py:4 of the reference
COMPATIBILITY_MATRIX_OPENGL_TO_BLENDER = np.array([[1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, -1.0]])

# Rows of the 137-joint SMPL-X skeleton that take part (body + both hands, no face): the value of
# utils.smpl.smpl_to_openpose("smplx", use_hands=True, use_face=False, use_face_contour=False) (a fixed index table;
# tests/golden/triangulation_golden.npz holds the reference's own copy and tests/test_triangulation.py compares).
BODY_HAND_INDICES = np.array([55, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65,
                              20, 37, 38, 39, 66, 25, 26, 27, 67, 28, 29, 30, 68, 34, 35, 36, 69, 31, 32, 33, 70,
                              21, 52, 53, 54, 71, 40, 41, 42, 72, 43, 44, 45, 73, 49, 50, 51, 74, 46, 47, 48, 75], dtype=np.int64)

VIEW_DOUBLES = 28


def view_record(camera_config) -> np.ndarray:
    """28 doubles per camera, computed with the reference's own expressions (optimize_depth.py:164-200)."""
    res, scale = camera_config["resolution"], camera_config["scale"]
    R, t = np.asarray(camera_config["R"], dtype=np.float64), np.asarray(camera_config["t"], dtype=np.float64).reshape((1, 3))
    C = COMPATIBILITY_MATRIX_OPENGL_TO_BLENDER
    rotation = (C @ R.T) / scale * max(res)
    translation = (-C @ R.T @ t.T) / scale * max(res)
    mr = R @ C
    tmr = t @ mr
    return np.concatenate([rotation.ravel(), translation.ravel(), mr.ravel(), tmr.ravel(),
                           [float(scale), float(max(res)), res[0] / 2, res[1] / 2]]).astype(np.float64)


def _f64(x, device):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float64).to(device).contiguous()


def dlt_score(views, ref_view, ref_xy, cand_view, cand_xy, device="cuda"):
    """views [V,28]; ref_xy [J,2]; cand_view [P] view index per candidate; cand_xy [P,J,2] ->
    (tri [P,J,3], ref_mse [P], other_mse [P]) f64 device tensors."""
    v, rxy, cxy = _f64(views, device), _f64(ref_xy, device), _f64(cand_xy, device)
    cv = torch.as_tensor(np.ascontiguousarray(cand_view), dtype=torch.int32).to(device)
    P, J = int(cxy.shape[0]), int(rxy.shape[0])
    tri = torch.empty(P, J, 3, dtype=torch.float64, device=device)
    rm, om = torch.empty(P, dtype=torch.float64, device=device), torch.empty(P, dtype=torch.float64, device=device)
    f64 = torch.float64
    rc = _lib.lib().coma_dlt_score_f64(_lib.ptr(v, f64, "views"), int(v.shape[0]), int(ref_view), _lib.ptr(rxy, f64), _lib.ptr(cv, torch.int32),
                                       _lib.ptr(cxy, f64), P, J, _lib.ptr(tri, f64), _lib.ptr(rm, f64), _lib.ptr(om, f64),
                                       _lib.stream_ptr(tri.device))
    _lib.check(rc, "coma_dlt_score_f64")
    return tri, rm, om, (v, cv, cxy)


def ransac_matrix(dev_state, tri, sel, threshold):
    """mse [C,C] and inlier counts [C] over the selected candidates `sel` (indices into the scored candidates)."""
    v, cv, cxy = dev_state
    s = torch.as_tensor(np.ascontiguousarray(sel), dtype=torch.int32).to(tri.device)
    C, J = int(s.numel()), int(tri.shape[1])
    mse = torch.empty(C, C, dtype=torch.float64, device=tri.device)
    counts = torch.empty(C, dtype=torch.int32, device=tri.device)
    f64 = torch.float64
    rc = _lib.lib().coma_ransac_mse_f64(_lib.ptr(v, f64), _lib.ptr(tri, f64), _lib.ptr(cv, torch.int32), _lib.ptr(cxy, f64),
                                        _lib.ptr(s, torch.int32), C, J, float(threshold), _lib.ptr(mse, f64), _lib.ptr(counts, torch.int32),
                                        _lib.stream_ptr(tri.device))
    _lib.check(rc, "coma_ransac_mse_f64")
    return mse, counts


def select_inliers(ref_xy, ref_cam, preds, maximum_candidates, ransac_threshold=200, triangulation_threshold=10,
                   body_hand_indices=BODY_HAND_INDICES, device="cuda"):
    """Array-level core.  preds: list of (joints_proj [137,2], camera_config).  Returns (ordered inlier positions into
    `preds`, their reprojection MSEs, dict of intermediates)."""
    idx = np.asarray(body_hand_indices)
    cams, cam_ids = [ref_cam], {id(ref_cam): 0}
    cand_view = []
    for _, cam in preds:
        if id(cam) not in cam_ids:
            cam_ids[id(cam)] = len(cams)
            cams.append(cam)
        cand_view.append(cam_ids[id(cam)])
    if not preds:
        return [], [], dict(best=[], tri=None)
    views = np.stack([view_record(c) for c in cams])
    ref_j = np.asarray(ref_xy, dtype=np.float64)[idx]
    cand_xy = np.stack([np.asarray(xy, dtype=np.float64)[idx] for xy, _ in preds])
    tri, rm, om, st = dlt_score(views, 0, ref_j, cand_view, cand_xy, device)
    rm_h, om_h = rm.cpu().numpy(), om.cpu().numpy()
    total = rm_h + om_h                                              # optimize_depth.py:293
    keep = [i for i in range(len(preds)) if rm_h[i] < triangulation_threshold]
    best = sorted(keep, key=lambda i: total[i])[:maximum_candidates]   # :327 (stable)
    info = dict(best=best, tri=tri, ref_mse=rm_h, other_mse=om_h)
    if not best:
        return [], [], info
    mse, counts = ransac_matrix(st, tri, best, ransac_threshold)
    counts_h = counts.cpu().numpy()
    info["mse"], info["counts"] = mse, counts_h
    winner, max_incl = -1, 0
    for a in range(len(best)):                                       # :360-363: strict '>' keeps the first maximum
        if counts_h[a] > max_incl:
            winner, max_incl = a, int(counts_h[a])
    if winner < 0:
        return [], [], info
    row = mse[winner].cpu().numpy()
    incl = sorted([b for b in range(len(best)) if row[b] < ransac_threshold], key=lambda b: row[b])   # :366
    info["winner"] = winner
    return [best[b] for b in incl], [float(row[b]) for b in incl], info


def compute_ransac_inclusives_with_triangulation(joints_proj, inpaint_pth, human_preds_dir, camera_dir, maximum_candidates,
                                                 ransac_threshold=200, triangulation_threshold=10,
                                                 enable_aggregate_total_prompts=False, allowed_viewpoint_prompts=None, *,
                                                 perturb_view_num=None, body_hand_indices=BODY_HAND_INDICES, device="cuda"):
    """Drop-in for optimize_depth.py:143-368.  `perturb_view_num`: the reference restricts the search to the reference view's
    group of `view_num` cameras when CATEGORY2PERTURB_CONFIG[..]["need_perturb"] (:271-275); pass that view_num (None = all)."""
    supercategory, category, asset_id, view_id, _, prompt, __ = inpaint_pth.split("/")[-7:]
    cam_cache = {}

    def camera(vid):
        if vid not in cam_cache:
            with open(f"{camera_dir}/{supercategory}/{category}/{asset_id}/{vid}.pickle", "rb") as handle:
                d = pickle.load(handle)
            cam_cache[vid] = dict(R=d["R"], t=d["t"], resolution=d["resolution"], scale=d["scale"])
        return cam_cache[vid]

    parts = prompt.split(",")
    mainprompt = parts[0]
    this_view_prompt = "original" if len(parts) == 1 else parts[-1].strip().lower()
    assert this_view_prompt in allowed_viewpoint_prompts
    base = f"{human_preds_dir}/{supercategory}/{category}/{asset_id}/*[!{view_id}]*"
    pths = []
    for vp in allowed_viewpoint_prompts:                              # :248-268
        if enable_aggregate_total_prompts:
            if vp == "original":
                pths += [p for p in glob(f"{base}/*/*/*.pickle") if "," not in p.split("/")[-2]]
            else:
                pths += list(glob(f"{base}/*/*{vp}*/*.pickle"))
        elif vp == "original":
            pths += list(glob(f"{base}/*/{mainprompt}/*.pickle"))
        else:
            pths += list(glob(f"{base}/*/*{mainprompt}*{vp}*/*.pickle"))
    if not enable_aggregate_total_prompts:
        pths = list(set(pths))
    if perturb_view_num:
        group = int(view_id.split(":")[-1]) // perturb_view_num
        pths = [p for p in pths if int(p.split("/")[-4].split(":")[-1]) // perturb_view_num == group]
    pths = sorted(pths)          # the reference iterates glob / set order; only ties of its stable sorts depend on it
    preds, meta = [], []
    for pth in pths:
        with open(pth, "rb") as handle:
            hp = pickle.load(handle)
        if type(hp) == str:      # sentinel written by the upstream stage for "no human"
            continue
        other_view = pth.split("/")[-4]
        preds.append((hp["joints_proj"], camera(other_view)))
        meta.append((pth, other_view, hp["joints_proj"]))
    order, mses, _ = select_inliers(joints_proj, camera(view_id), preds, maximum_candidates, ransac_threshold, triangulation_threshold,
                                    body_hand_indices, device)

    def to_tensor(x):
        if torch.is_tensor(x):
            return x.to(device=device).float()
        if isinstance(x, np.ndarray):
            return torch.from_numpy(x).to(device=device).float()
        return x

    out = []
    for n, e in zip(order, mses):
        pth, other_view, xy = meta[n]
        out.append(dict(human_pred_pth=pth, view_id=other_view, camera_config={k: to_tensor(v) for k, v in camera(other_view).items()},
                        joints_proj=to_tensor(xy).unsqueeze(0), joints_MSE=e))
    return out
