#!/usr/bin/env python3
"""Golden vectors G18 for the two-view DLT triangulation + RANSAC inlier search, produced by the REAL
reference function `compute_ransac_inclusives_with_triangulation` (/root/reference/src/generation/optimize_depth.py:143-368).

The function is imported from /root/reference with its third-party imports stubbed (pytorch3d, trimesh, smplx, COAP: none
is touched by this code path), `to_tensor(.., "cuda")` redirected to the CPU, and run on synthetic camera / human-prediction
pickles laid out the way the reference's globs expect.  numpy's `array` and `mean` are instrumented inside that module so
that the triangulated joints and every MSE the function computes on the way are captured too -- the function itself only
returns the final inlier list.  The oracle (oracle/triangulation_oracle.py) is asserted against all of it, then inputs and
outputs are stored in tests/golden/triangulation_golden.npz.   Run: python tests/golden/make_golden_triangulation.py
"""
import importlib
import os
import pickle
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def import_reference():
    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return lambda *a, **kw: None
    for name in ["pytorch3d", "pytorch3d.io", "trimesh", "trimesh.boolean", "smplx", "smplx.utils", "imports", "imports.coap",
                 "open3d", "cv2", "easydict"]:
        sys.modules.setdefault(name, _Any(name))
    sys.modules["smplx.utils"].SMPLXOutput = object
    # the repo has same-named drop-in packages (src, utils, constants): the reference's must be the ones imported, so the
    # repo root only joins sys.path afterwards (for oracle/)
    assert ROOT not in sys.path
    sys.path.insert(0, REF)
    m = importlib.import_module("src.generation.optimize_depth")
    assert m.__file__.startswith(REF), m.__file__
    from utils.smpl import smpl_to_openpose
    idx = smpl_to_openpose(model_type="smplx", use_hands=True, use_face=False, use_face_contour=False)
    sys.path.remove(REF)
    sys.path.insert(0, ROOT)
    return m, np.asarray(idx)


class NPProxy:
    """numpy with `array` and `mean` logged (used to see the intermediates of the reference function)."""
    def __init__(self):
        self.arrays, self.means = [], []

    def __getattr__(self, k):
        return getattr(np, k)

    def array(self, *a, **kw):
        r = np.array(*a, **kw)
        if a and isinstance(a[0], list) and len(a[0]) and getattr(a[0][0], "shape", None) == (3, 1):
            self.arrays.append(r.reshape(-1, 3).copy())
        return r

    def mean(self, *a, **kw):
        r = np.mean(*a, **kw)
        self.means.append(float(r))
        return r


def look_at(eye, target=np.zeros(3)):
    f = target - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, [0.0, 0.0, 1.0])
    r /= np.linalg.norm(r)
    u = np.cross(r, f)
    return np.stack([r, u, -f], axis=1)        # camera-to-world rotation, columns = camera axes (OpenGL: -z forward)


def make_scene(seed, n_views, preds_per_view, n_out, noise):
    """Cameras on a ring around a skeleton; per other view a few 2D predictions: inliers = projections + noise, outliers =
    projections of a displaced / scaled skeleton.  Returns everything the reference reads from disk."""
    from oracle import triangulation_oracle as T
    rng = np.random.default_rng(seed)
    J = 137
    skel = rng.normal(scale=[0.25, 0.15, 0.45], size=(J, 3)) + np.array([0.0, 0.0, 0.9])
    cams = []
    for v in range(n_views):
        ang = 2 * np.pi * v / n_views + rng.normal(scale=0.05)
        eye = np.array([2.6 * np.cos(ang), 2.6 * np.sin(ang), 1.2 + rng.normal(scale=0.2)])
        cams.append(dict(R=look_at(eye, np.array([0, 0, 0.9])), t=eye.copy(), resolution=(512, 512), scale=float(2.4 + 0.1 * v)))
    ref_xy = T.render(skel.copy(), cams[0]) + rng.normal(scale=noise, size=(J, 2))
    preds = []                                   # (view index, joints_proj [J,2]) or a sentinel string
    for v in range(1, n_views):
        for k in range(preds_per_view):
            if k < preds_per_view - n_out:
                xy = T.render(skel.copy(), cams[v]) + rng.normal(scale=noise * (1 + k), size=(J, 2))
            else:
                bad = skel * rng.uniform(0.6, 1.5) + rng.normal(scale=0.35, size=3)
                xy = T.render(bad.copy(), cams[v]) + rng.normal(scale=noise, size=(J, 2))
            preds.append((v, xy))
    return cams, ref_xy, preds


def write_tree(root, cams, preds, sentinel_at=()):
    sup, cat, asset, prompt = "BEHAVE", "backpack", "000", "a person carrying a backpack"
    cam_dir, pred_dir = os.path.join(root, "cameras"), os.path.join(root, "human_preds")
    for v, c in enumerate(cams):
        d = os.path.join(cam_dir, sup, cat, asset)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"view:{v:03d}.pickle"), "wb") as h:
            pickle.dump(dict(R=c["R"], t=c["t"], resolution=c["resolution"], scale=c["scale"]), h)
    paths = []
    for n, (v, xy) in enumerate(preds):
        d = os.path.join(pred_dir, sup, cat, asset, f"view:{v:03d}", "mask0", prompt)
        os.makedirs(d, exist_ok=True)
        pth = os.path.join(d, f"{n:04d}.pickle")
        with open(pth, "wb") as h:
            pickle.dump("NO HUMAN" if n in sentinel_at else dict(joints_proj=xy), h)
        paths.append(pth)
    inpaint = os.path.join(root, "inpaint", sup, cat, asset, "view:000", "mask0", prompt, "0000.png")
    return cam_dir, pred_dir, inpaint, paths


def main():
    m, idx = import_reference()
    from oracle import triangulation_oracle as T
    m.to_tensor = (lambda f: (lambda x, device: f(x, "cpu")))(m.to_tensor)
    out = {"body_hand_indices": idx.astype(np.int64)}
    checks = []
    cases = [("a", dict(seed=1, n_views=6, preds_per_view=4, n_out=1, noise=1.5), dict(maximum_candidates=400, ransac_threshold=200, triangulation_threshold=100), ()),
             ("b", dict(seed=2, n_views=9, preds_per_view=6, n_out=3, noise=3.0), dict(maximum_candidates=7, ransac_threshold=120, triangulation_threshold=60), (3, 11)),
             ("c", dict(seed=3, n_views=4, preds_per_view=3, n_out=3, noise=1.0), dict(maximum_candidates=400, ransac_threshold=1e-6, triangulation_threshold=100), ())]
    for tag, scene, par, sentinels in cases:
        cams, ref_xy, preds = make_scene(**scene)
        root = tempfile.mkdtemp(prefix="g18_")
        try:
            cam_dir, pred_dir, inpaint, paths = write_tree(root, cams, preds, sentinels)
            proxy = NPProxy()
            m.np = proxy
            # the reference iterates glob order (after list(set(..))): capture the order it actually used via the logged arrays
            res = m.compute_ransac_inclusives_with_triangulation(ref_xy, inpaint, pred_dir, cam_dir, par["maximum_candidates"],
                                                                par["ransac_threshold"], par["triangulation_threshold"], False,
                                                                ["original"])
            m.np = np
        finally:
            shutil.rmtree(root)
        sel_paths = [r["human_pred_pth"] for r in res]
        sel_n = [paths.index(p) for p in sel_paths]
        sel_mse = [float(r["joints_MSE"]) for r in res]
        # ---- oracle on the same inputs, candidates in prediction order (the final result does not depend on the glob
        # order except through ties of the stable sorts, which the scenes avoid)
        # BEHAVE/backpack has need_perturb=True, view_num=4 (constants/generation/assets.py): only predictions from the
        # reference view's group of 4 views take part (optimize_depth.py:271-275)
        valid = [n for n in range(len(preds)) if n not in sentinels and preds[n][0] // 4 == 0]
        pl = [(preds[n][1], cams[preds[n][0]]) for n in valid]
        tri, rm, om = T.score_candidates(ref_xy, cams[0], pl, idx)
        best = T.select_candidates(rm, om, par["triangulation_threshold"], par["maximum_candidates"])
        mse, winner, incl, incl_mse = T.ransac(tri, pl, best, idx, par["ransac_threshold"])
        o_sel = [valid[best[b]] for b in incl]
        # intermediates logged from the reference: one tri array per valid prediction (in ITS iteration order)
        assert len(proxy.arrays) == len(valid), (len(proxy.arrays), len(valid))
        n_match = 0
        for arr in proxy.arrays:
            d = [float(np.abs(arr - tri[i]).max()) for i in range(len(valid))]
            i = int(np.argmin(d))
            ok = d[i] <= 1e-9 * max(1.0, float(np.abs(tri[i]).max()))
            n_match += ok
        checks.append((f"G18{tag} triangulated joints ({len(valid)} candidates)", n_match == len(valid)))
        ref_means = np.array(proxy.means[:2 * len(valid)]).reshape(-1, 2)
        checks.append((f"G18{tag} candidate MSEs", np.allclose(np.sort(ref_means[:, 0]), np.sort(rm), rtol=1e-9) and
                       np.allclose(np.sort(ref_means[:, 1]), np.sort(om), rtol=1e-9)))
        ref_mat = np.array(proxy.means[2 * len(valid):])
        checks.append((f"G18{tag} RANSAC matrix ({len(best)}^2 entries)", ref_mat.size == mse.size and
                       np.allclose(np.sort(ref_mat), np.sort(mse.ravel()), rtol=1e-9)))
        checks.append((f"G18{tag} inlier set + order ({len(sel_n)} inliers)", sel_n == o_sel))
        checks.append((f"G18{tag} inlier MSEs", np.allclose(sel_mse, incl_mse, rtol=1e-9) if sel_n else incl_mse == []))
        out[f"g18{tag}_cam_R"] = np.stack([c["R"] for c in cams])
        out[f"g18{tag}_cam_t"] = np.stack([c["t"] for c in cams])
        out[f"g18{tag}_cam_res"] = np.array([c["resolution"] for c in cams], dtype=np.int64)
        out[f"g18{tag}_cam_scale"] = np.array([c["scale"] for c in cams])
        out[f"g18{tag}_ref_xy"] = ref_xy
        out[f"g18{tag}_pred_view"] = np.array([v for v, _ in preds], dtype=np.int64)
        out[f"g18{tag}_pred_xy"] = np.stack([xy for _, xy in preds])
        out[f"g18{tag}_sentinels"] = np.array(sentinels, dtype=np.int64)
        out[f"g18{tag}_valid"] = np.array(valid, dtype=np.int64)
        out[f"g18{tag}_params"] = np.array([par["maximum_candidates"], par["ransac_threshold"], par["triangulation_threshold"]], dtype=np.float64)
        out[f"g18{tag}_tri"] = tri                         # oracle values, asserted equal to the reference's above
        out[f"g18{tag}_ref_mse"], out[f"g18{tag}_other_mse"] = rm, om
        out[f"g18{tag}_best"] = np.array([valid[b] for b in best], dtype=np.int64)
        out[f"g18{tag}_ransac_mse"] = mse
        out[f"g18{tag}_selected"] = np.array(sel_n, dtype=np.int64)      # from the REFERENCE's return value
        out[f"g18{tag}_selected_mse"] = np.array(sel_mse)
    for name, ok in checks:
        print(f"  oracle vs reference  {name:56s} {'OK' if ok else 'MISMATCH'}")
    assert all(ok for _, ok in checks)
    np.savez_compressed(os.path.join(HERE, "triangulation_golden.npz"), **out)
    print(f"{len(checks)} checks OK; wrote triangulation_golden.npz ({os.path.getsize(os.path.join(HERE, 'triangulation_golden.npz')) / 1e3:.0f} kB)")


if __name__ == "__main__":
    main()
