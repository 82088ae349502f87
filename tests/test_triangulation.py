"""Two-view DLT triangulation + RANSAC inlier search.
CPU part: the oracle (oracle/triangulation_oracle.py) against the golden vectors captured from the REAL reference function
(tests/golden/make_golden_triangulation.py).  GPU part: the HIP path through the C ABI against those vectors and against
the oracle on larger seeded scenes.  Bars: inlier index sets and their order bit-exact; floats <= 1e-6 relative (the brief
allows 1e-3; both sides are f64, they differ only by pinv-via-SVD vs normal equations and by summation order)."""
import os
import pickle

import numpy as np
import pytest

from oracle import triangulation_oracle as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["a", "b", "c"]


@pytest.fixture(scope="module")
def g18():
    return np.load(os.path.join(ROOT, "tests", "golden", "triangulation_golden.npz"))


def _scene(g, tag):
    cams = [dict(R=g[f"g18{tag}_cam_R"][v], t=g[f"g18{tag}_cam_t"][v], resolution=tuple(int(x) for x in g[f"g18{tag}_cam_res"][v]),
                 scale=float(g[f"g18{tag}_cam_scale"][v])) for v in range(len(g[f"g18{tag}_cam_scale"]))]
    valid = [int(n) for n in g[f"g18{tag}_valid"]]
    preds = [(g[f"g18{tag}_pred_xy"][n], cams[int(g[f"g18{tag}_pred_view"][n])]) for n in valid]
    mc, rt, tt = g[f"g18{tag}_params"]
    return cams, g[f"g18{tag}_ref_xy"], preds, valid, int(mc), float(rt), float(tt)


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)) if a.size else 0.0


# ------------------------------------------------------------------------------------------------ CPU: oracle vs reference vectors
@pytest.mark.parametrize("tag", CASES)
def test_oracle_reproduces_the_reference(g18, tag):
    cams, ref_xy, preds, valid, mc, rt, tt = _scene(g18, tag)
    idx = g18["body_hand_indices"]
    tri, rm, om = T.score_candidates(ref_xy, cams[0], preds, idx)
    assert _rel(tri, g18[f"g18{tag}_tri"]) <= 1e-12 and _rel(rm, g18[f"g18{tag}_ref_mse"]) <= 1e-12
    best = T.select_candidates(rm, om, tt, mc)
    assert [valid[b] for b in best] == g18[f"g18{tag}_best"].tolist()
    mse, winner, incl, incl_mse = T.ransac(tri, preds, best, idx, rt)
    assert [valid[best[b]] for b in incl] == g18[f"g18{tag}_selected"].tolist()        # the reference's own return value
    assert _rel(incl_mse, g18[f"g18{tag}_selected_mse"]) <= 1e-12


def test_index_table_and_view_record_match_the_reference(g18):
    from coma_amd import triangulate as tr
    assert np.array_equal(tr.BODY_HAND_INDICES, g18["body_hand_indices"])
    cams, *_ = _scene(g18, "a")
    for cam in cams:
        rec = tr.view_record(cam)
        rot, trans = T.projection(cam)
        assert np.array_equal(rec[:9].reshape(3, 3), rot) and np.array_equal(rec[9:12], trans.ravel())
        assert np.array_equal(rec[12:21].reshape(3, 3), cam["R"] @ T.COMPAT)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_hip_path_matches_the_reference_vectors(g18, tag, hip_lib):
    from coma_amd import triangulate as tr
    cams, ref_xy, preds, valid, mc, rt, tt = _scene(g18, tag)
    order, mses, info = tr.select_inliers(ref_xy, cams[0], preds, mc, rt, tt, device="cuda:0")
    assert _rel(info["tri"].cpu().numpy(), g18[f"g18{tag}_tri"]) <= 1e-6
    assert _rel(info["ref_mse"], g18[f"g18{tag}_ref_mse"]) <= 1e-6 and _rel(info["other_mse"], g18[f"g18{tag}_other_mse"]) <= 1e-6
    assert [valid[b] for b in info["best"]] == g18[f"g18{tag}_best"].tolist()
    if info["best"]:
        assert _rel(info["mse"].cpu().numpy(), g18[f"g18{tag}_ransac_mse"]) <= 1e-6
    assert [valid[n] for n in order] == g18[f"g18{tag}_selected"].tolist()             # bit-exact index set AND order
    assert _rel(mses, g18[f"g18{tag}_selected_mse"]) <= 1e-6


@pytest.mark.gpu
def test_file_level_drop_in(g18, tmp_path, hip_lib):
    """The reference's directory layout (camera pickles, prediction pickles, sentinel strings, view-group filter)."""
    import torch
    from coma_amd import triangulate as tr
    tag = "b"
    g = g18
    sup, cat, asset, prompt = "BEHAVE", "backpack", "000", "a person carrying a backpack"
    sentinels = set(g[f"g18{tag}_sentinels"].tolist())
    for v in range(len(g[f"g18{tag}_cam_scale"])):
        d = tmp_path / "cameras" / sup / cat / asset
        d.mkdir(parents=True, exist_ok=True)
        with open(d / f"view:{v:03d}.pickle", "wb") as h:
            pickle.dump(dict(R=g[f"g18{tag}_cam_R"][v], t=g[f"g18{tag}_cam_t"][v], resolution=tuple(int(x) for x in g[f"g18{tag}_cam_res"][v]),
                             scale=float(g[f"g18{tag}_cam_scale"][v])), h)
    paths = []
    for n, v in enumerate(g[f"g18{tag}_pred_view"]):
        d = tmp_path / "human_preds" / sup / cat / asset / f"view:{int(v):03d}" / "mask0" / prompt
        d.mkdir(parents=True, exist_ok=True)
        with open(d / f"{n:04d}.pickle", "wb") as h:
            pickle.dump("NO HUMAN" if n in sentinels else dict(joints_proj=g[f"g18{tag}_pred_xy"][n]), h)
        paths.append(str(d / f"{n:04d}.pickle"))
    inpaint = str(tmp_path / "inpaint" / sup / cat / asset / "view:000" / "mask0" / prompt / "0000.png")
    mc, rt, tt = g[f"g18{tag}_params"]
    res = tr.compute_ransac_inclusives_with_triangulation(g[f"g18{tag}_ref_xy"], inpaint, str(tmp_path / "human_preds"), str(tmp_path / "cameras"),
                                                          int(mc), float(rt), float(tt), False, ["original"], perturb_view_num=4, device="cuda:0")
    assert [paths.index(r["human_pred_pth"]) for r in res] == g[f"g18{tag}_selected"].tolist()
    assert _rel([r["joints_MSE"] for r in res], g[f"g18{tag}_selected_mse"]) <= 1e-6
    r0 = res[0]
    assert set(r0) == {"human_pred_pth", "view_id", "camera_config", "joints_proj", "joints_MSE"}
    assert r0["joints_proj"].shape == (1, 137, 2) and r0["joints_proj"].dtype == torch.float32 and r0["joints_proj"].is_cuda
    assert r0["camera_config"]["R"].is_cuda and r0["view_id"].startswith("view:")


@pytest.mark.gpu
def test_full_size_candidate_set_against_the_oracle(hip_lib):
    """maximum_candidates = 400 (the reference's default): 600 predictions over 12 views, half of them outliers."""
    from coma_amd import triangulate as tr
    rng = np.random.default_rng(7)
    J = 137
    skel = rng.normal(scale=[0.25, 0.15, 0.45], size=(J, 3)) + np.array([0.0, 0.0, 0.9])
    cams = []
    for v in range(12):
        ang = 2 * np.pi * v / 12
        eye = np.array([2.6 * np.cos(ang), 2.6 * np.sin(ang), 1.2])
        f = (np.array([0, 0, 0.9]) - eye) / np.linalg.norm(np.array([0, 0, 0.9]) - eye)
        r = np.cross(f, [0.0, 0.0, 1.0]); r /= np.linalg.norm(r)
        cams.append(dict(R=np.stack([r, np.cross(r, f), -f], axis=1), t=eye, resolution=(512, 384), scale=2.5))
    ref_xy = T.render(skel.copy(), cams[0]) + rng.normal(scale=1.0, size=(J, 2))
    preds = []
    for n in range(600):
        v = 1 + n % 11
        s = skel if n % 2 == 0 else skel * rng.uniform(0.7, 1.4) + rng.normal(scale=0.3, size=3)
        preds.append((T.render(s.copy(), cams[v]) + rng.normal(scale=1.0 + 0.01 * n, size=(J, 2)), cams[v]))
    order, mses, info = tr.select_inliers(ref_xy, cams[0], preds, 400, 200, 100, device="cuda:0")
    tri, rm, om = T.score_candidates(ref_xy, cams[0], preds, tr.BODY_HAND_INDICES)
    best = T.select_candidates(rm, om, 100, 400)
    mse, winner, incl, incl_mse = T.ransac(tri, preds, best, tr.BODY_HAND_INDICES, 200)
    assert info["best"] == best and len(best) == 400 or len(best) == len(info["best"])
    assert info["best"] == best
    assert order == [best[b] for b in incl] and len(order) > 50
    assert _rel(mses, incl_mse) <= 1e-6 and _rel(info["mse"].cpu().numpy(), mse) <= 1e-6
