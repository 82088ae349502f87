"""CPU restatement (NumPy, float64) of the reference's two-view DLT triangulation and RANSAC inlier search -- TEST
INFRASTRUCTURE ONLY.  Nothing under coma_amd/, utils/, src/ or the timed region of bench.py may import this module.

Follows /root/reference/src/generation/optimize_depth.py:
    get_projection_matrix        :164-183   rotation = (C @ R^T) / scale * max(res), translation = (-C @ R^T @ t^T) / scale * max(res)
    get_view2joints_render       :185-200   X @ (R C) - t (R C), pixel scale, image-centre offset
    solve_DLT                    :202-237   per joint: A = [ref_rot[0:2]; other_rot[0:2]] (4x3), b (4x1), x = pinv(A) b
    candidate scoring            :291-324   ref / other reprojection MSE of the triangulated joints
    candidate selection          :326-327   ref MSE < triangulation_threshold, stable sort by total MSE, first maximum_candidates
    RANSAC                       :329-366   best candidate's joints reprojected into every other candidate's view, MSE < ransac_threshold,
                                            FIRST candidate with the strictly largest inlier count wins, inliers sorted by MSE (stable)
C = COMPATIBILITY_MATRIX_OPENGL_TO_BLENDER = diag(1, -1, -1) (constants/generation/visualizers.py:4).

Pinned: tests/golden/make_golden_triangulation.py runs the REAL reference function on synthetic camera / prediction
pickles (third-party imports stubbed, `to_tensor(.., "cuda")` redirected to the CPU) with numpy's `array` / `mean`
instrumented, and asserts this restatement against every intermediate (triangulated joints, MSEs, RANSAC matrix entries)
and the returned inlier list; the captured vectors are tests/golden/triangulation_golden.npz.
"""
import numpy as np

COMPAT = np.array([[1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, -1.0]])


def projection(cam):
    """optimize_depth.py:164-183 -> (rotation [3,3], translation [3,1])."""
    res, scale, R, t = cam["resolution"], cam["scale"], cam["R"], cam["t"].reshape((1, 3))
    rotation = (COMPAT @ R.T) / scale * max(res)
    translation = (-COMPAT @ R.T @ t.T) / scale * max(res)
    return rotation, translation


def render(joints, cam):
    """optimize_depth.py:185-200: [J,3] world joints -> [J,2] pixels."""
    res, scale, R, t = cam["resolution"], cam["scale"], cam["R"], cam["t"]
    jc = joints @ (R @ COMPAT) - t.reshape((1, 3)) @ (R @ COMPAT)
    jc[:, 0] = jc[:, 0] / scale * max(res) + res[0] / 2
    jc[:, 1] = jc[:, 1] / scale * max(res) + res[1] / 2
    jc[:, 2] = jc[:, 2] / scale * max(res)
    return jc[:, :2]


def solve_dlt(ref_xy, ref_cam, other_xy, other_cam):
    """optimize_depth.py:202-237: [J,2] pixel joints in two views -> [J,3]."""
    r0 = ref_xy - np.array(ref_cam["resolution"]).reshape((1, 2)) / 2
    o0 = other_xy - np.array(other_cam["resolution"]).reshape((1, 2)) / 2
    rr, rt = projection(ref_cam)
    orot, ot = projection(other_cam)
    out = []
    for (rx, ry), (ox, oy) in zip(r0, o0):
        A = np.vstack([rr[0, :], rr[1, :], orot[0, :], orot[1, :]])
        b = np.array([rx - rt[0, 0], ry - rt[1, 0], ox - ot[0, 0], oy - ot[1, 0]]).reshape(4, 1)
        out.append((np.linalg.pinv(A) @ b).reshape((3, 1)))
    return np.array(out).reshape((-1, 3))


def score_candidates(ref_xy_all, ref_cam, preds, idx):
    """optimize_depth.py:291-324.  preds: list of (joints_proj [137,2], cam).  -> tri [P,J,3], ref_mse [P], other_mse [P]."""
    tri, rm, om = [], [], []
    for xy, cam in preds:
        t = solve_dlt(ref_xy_all[idx], ref_cam, xy[idx], cam)
        tri.append(t)
        rm.append(np.mean(np.sum((render(t, ref_cam) - ref_xy_all[idx]) ** 2, axis=1)))
        om.append(np.mean(np.sum((render(t, cam) - xy[idx]) ** 2, axis=1)))
    return np.array(tri), np.array(rm), np.array(om)


def select_candidates(ref_mse, other_mse, triangulation_threshold, maximum_candidates):
    """optimize_depth.py:326-327 -> indices into the candidate list, in the reference's order (stable sort by total MSE)."""
    total = ref_mse + other_mse
    keep = [i for i in range(len(total)) if ref_mse[i] < triangulation_threshold]
    return sorted(keep, key=lambda i: total[i])[:maximum_candidates]


def ransac(tri, preds, best, idx, ransac_threshold):
    """optimize_depth.py:329-366 -> (mse matrix [C,C] over the best candidates, winner position or -1, ordered inlier
    positions (into `best`), their MSEs)."""
    C = len(best)
    mse = np.zeros((C, C))
    for a, ia in enumerate(best):
        for b_, ib in enumerate(best):
            xy, cam = preds[ib]
            mse[a, b_] = np.mean(np.sum((xy[idx] - render(tri[ia], cam)) ** 2, axis=1))
    winner, max_incl = -1, 0
    for a in range(C):
        n = int(np.sum(mse[a] < ransac_threshold))
        if n > max_incl:
            winner, max_incl = a, n
    if winner < 0:
        return mse, -1, [], []
    incl = [b_ for b_ in range(C) if mse[winner, b_] < ransac_threshold]
    incl = sorted(incl, key=lambda b_: mse[winner, b_])
    return mse, winner, incl, [mse[winner, b_] for b_ in incl]
