"""CPU oracle for the ComA aggregation path  --  TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the arithmetic of the reference's ComA accumulators and
reducers.  It exists so that the HIP kernels in ``coma_amd/csrc`` can be checked on a box where
``/root/reference`` does not exist.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product path (``coma_amd``) never does and
fails loudly when the HIP library is missing.

Pinning: every function below is asserted against the *real* reference, imported on CPU in the build
container, by ``tests/golden/make_golden.py``; the resulting vectors are committed under
``tests/golden/*.npz`` and re-checked by ``tests/test_oracle_golden.py``.

dtype flow follows the reference exactly:
  * sphere bins are f64 while learning (utils/coma.py:204-205),
  * samples are cast to f32 (utils/misc.py:48-50 via utils/coma.py:274),
  * K1/K2 run in f32, K3 runs in f64 and is added in place into f32 grids (utils/coma.py:312-323),
  * occupancy distance test runs in f64 against an f64 voxel grid (utils/coma_occupancy.py:171,292).
Reduction orders that decide bit-exact outputs (3-term sums) are written out explicitly as
``(x0 + x1) + x2`` which is what both torch and numpy do for a length-3 axis (probed).
"""
from __future__ import annotations

import math
import numpy as np

F32 = np.float32
F64 = np.float64


# --------------------------------------------------------------------------------------------
# A.1  sphere bins                                   reference: utils/coma.py:18-26
# --------------------------------------------------------------------------------------------
def fibonacci_sphere(num_points: int) -> np.ndarray:
    """[N,3] f64 unit vectors; bin k of the orientation histogram."""
    u = np.arange(0, num_points, dtype=float) + 0.5
    phi = np.arccos(1 - 2 * u / num_points)
    theta = np.pi * (1 + 5**0.5) * u
    return np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=-1)


# --------------------------------------------------------------------------------------------
# helpers                                            reference: utils/transformations.py:8-17
# --------------------------------------------------------------------------------------------
def _sum3(x: np.ndarray) -> np.ndarray:
    """sum over a trailing axis of length 3 in the order torch/numpy use: (x0+x1)+x2."""
    return (x[..., 0] + x[..., 1]) + x[..., 2]


def normalize_rows(v: np.ndarray, eps: float) -> np.ndarray:
    assert v.ndim == 2 and v.shape[-1] == 3
    n = np.sqrt(_sum3(np.square(v)))[:, None]
    return v / (n + v.dtype.type(eps))


# --------------------------------------------------------------------------------------------
# A.4  K2: canonicalise a w.r.t. the rotation b -> p      reference: utils/coma.py:123-172
# --------------------------------------------------------------------------------------------
def canonicalize(a: np.ndarray, b: np.ndarray, p: np.ndarray, sub_p: np.ndarray, eps: float) -> np.ndarray:
    """a [A,3], b [B,3], p [3], sub_p [3] (all f32) -> [A,B,3] f32.

    Literal restatement, including the reference's incomplete skew matrix (row 2 lacks its
    [2,1] entry and [0,0] is set to b0), so that results agree for *any* p, not only p = z.
    """
    a = normalize_rows(a.astype(F32), eps)
    b = normalize_rows(b.astype(F32), eps)
    p = normalize_rows(p.astype(F32)[None], eps)[0]
    sp = normalize_rows(sub_p.astype(F32)[None], eps)[0]

    c = _sum3(b * p[None])[None, :]                      # [1,B]   b.p
    ab = _sum3(a[:, None, :] * b[None, :, :])            # [A,B]
    ap = _sum3(a * p[None])[:, None]                     # [A,1]
    asp = _sum3(a * sp[None])[:, None]                   # [A,1]

    opposite = ((F32(1) + c) < F32(eps))[:, :, None]     # [1,B,1]
    mirrored = F32(2) * asp[:, :, None] * sp[None, None, :] - a[:, None, :]   # [A,1,3]

    # v_j = M_j p with M_j = [[b0,-b2,b1],[b2,0,-b0],[-b1,0,0]]  (utils/coma.py:149-155)
    M = np.zeros([b.shape[0], 3, 3], dtype=F32)
    M[:, 0, 0] = b[:, 0]
    M[:, 0, 1] = -b[:, 2]
    M[:, 0, 2] = b[:, 1]
    M[:, 1, 0] = b[:, 2]
    M[:, 1, 2] = -b[:, 0]
    M[:, 2, 0] = -b[:, 1]
    v = np.einsum("bij,j->bi", M, p).astype(F32)         # [B,3]
    av = _sum3(a[:, None, :] * v[None, :, :])            # [A,B]

    out = v[None, :, :] * av[:, :, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where(opposite, F32(0), out / (F32(1) + c[:, :, None]))
    out = out + c[:, :, None] * a[:, None, :]
    out = out + ab[:, :, None] * p[None, None, :]
    out = out - ap[:, :, None] * b[None, :, :]
    out = np.where(opposite, mirrored, out).astype(F32)
    nrm = np.sqrt(_sum3(np.square(out)))[..., None]
    return (out / nrm).astype(F32)


# --------------------------------------------------------------------------------------------
# A.5  K3: geodesic-Gaussian soft histogram           reference: utils/coma.py:102-112
# --------------------------------------------------------------------------------------------
def geodesic_gaussian(grid: np.ndarray, canon: np.ndarray, sigma: float, eps: float) -> np.ndarray:
    """grid [N,3] (f64 while learning), canon [H,O,3] f32 -> [H,O,N] in the promoted dtype."""
    g = grid[None, None, :, :]
    c = canon[:, :, None, :]
    prod = g * c
    cos = (prod[..., 0] + prod[..., 1]) + prod[..., 2]
    geo = np.arccos(np.clip(cos, -1.0 + eps, 1.0 - eps))
    return 1.0 / np.exp(geo**2 / sigma**2)


def negative_exp(x, spatial_grid_size, spatial_grid_thres=None):
    """proximity score, reference: utils/coma.py:116-119."""
    return np.exp(-x / F32(spatial_grid_size)).astype(x.dtype)


# --------------------------------------------------------------------------------------------
# ComA accumulator + reducers                         reference: utils/coma.py:176-487, 614-641
# --------------------------------------------------------------------------------------------
class ComAOracle:
    def __init__(self, human_res, obj_res, normal_res, spatial_grid_size, spatial_grid_thres,
                 principle_vec=(0, 0, 1), sub_principle_vec=(0, 1, 0), sigma=0.1, eps=1e-8):
        self.H, self.O, self.N = int(human_res), int(obj_res), int(normal_res)
        self.size, self.thres = float(spatial_grid_size), float(spatial_grid_thres)
        self.sigma, self.eps = float(sigma), float(eps)
        self.grid = fibonacci_sphere(self.N)                                   # f64 [N,3]
        self.p = np.asarray(principle_vec, dtype=F32)
        self.sp = np.asarray(sub_principle_vec, dtype=F32)
        self.P_h_wrt_o = np.zeros([self.H, self.O, self.N], F32)
        self.P_o_wrt_h = np.zeros([self.H, self.O, self.N], F32)
        self.nom = np.zeros([self.H, self.O], F32)
        self.den = np.zeros([self.H, self.O], F32)
        self.cnt = np.zeros([self.H, self.O], F32)
        self.used_count = 0

    # reference: utils/coma.py:279-323
    def aggregate_sample(self, human_verts, human_normals, obj_verts, obj_normals):
        hv, hn = np.asarray(human_verts).astype(F32), np.asarray(human_normals).astype(F32)
        ov, on = np.asarray(obj_verts).astype(F32), np.asarray(obj_normals).astype(F32)
        assert hv.shape == (self.H, 3) and hn.shape == (self.H, 3)
        assert ov.shape == (self.O, 3) and on.shape == (self.O, 3)
        d = np.sqrt(_sum3(np.square(hv[:, None, :] - ov[None, :, :])))         # f32 [H,O]
        self.cnt += (d < F32(self.thres)).astype(np.int64)                     # exact integers
        self.nom += negative_exp(d, self.size)
        self.den += F32(1.0)
        c1 = canonicalize(hn, on, self.p, self.sp, self.eps)                   # [H,O,3]
        c2 = canonicalize(on, hn, self.p, self.sp, self.eps).transpose(1, 0, 2)
        # f64 scores added in place into f32 grids: computed in f64, rounded once to f32
        self.P_h_wrt_o += geodesic_gaussian(self.grid, c1, self.sigma, self.eps)
        self.P_o_wrt_h += geodesic_gaussian(self.grid, c2, self.sigma, self.eps)
        self.used_count += 1

    def state(self):
        return dict(prob_grid_canon_human_wrt_obj=self.P_h_wrt_o, prob_grid_canon_obj_wrt_human=self.P_o_wrt_h,
                    contact_dist_expectation_grid_nom=self.nom, contact_dist_expectation_grid_denom=self.den,
                    significant_contact_count=self.cnt, used_count=self.used_count)

    # reference: utils/coma.py:328-330 (in place, at the head of every reducer)
    def normalize(self):
        self.P_h_wrt_o /= self.P_h_wrt_o.sum(axis=-1, keepdims=True) + F32(self.eps)
        self.P_o_wrt_h /= self.P_o_wrt_h.sum(axis=-1, keepdims=True) + F32(self.eps)

    # reference: utils/coma.py:333-366.  grid_f32=True reproduces the post-``load`` state.
    def contact_map(self, grid_f32=False):
        self.normalize()
        grid = self.grid.astype(F32) if grid_f32 else self.grid
        dots = _sum3(self.p[None, :] * grid)[None, None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            prox = self.nom / self.den
        w = (1.0 - dots) / 2.0                      # f64 while learning, f32 after load
        on_h = (self.P_h_wrt_o * w).sum(axis=-1) * prox
        on_o = (self.P_o_wrt_h * w).sum(axis=-1) * prox
        # callers see f32: to_np_torch_recursive casts on the way out (utils/misc.py:56-58)
        return on_h.astype(F32), on_o.astype(F32)

    # reference: utils/coma.py:369-383
    def significant_pairs(self, ratio):
        return self.cnt >= F32(ratio * self.used_count)

    # reference: utils/coma.py:385-438, 614-641
    def aggregated_contact(self, which, ratio, grid_f32=False):
        assert which in ("human", "obj")
        on_h, on_o = self.contact_map(grid_f32)
        pairs = self.significant_pairs(ratio)
        if which == "human":
            cols = pairs.any(axis=0)
            agg = on_h[:, cols].max(axis=-1) if cols.any() else np.zeros([self.H], F32)
            idx = np.argwhere(pairs.any(axis=0))[:, 0]
        else:
            rows = pairs.any(axis=1)
            agg = on_o[rows, :].max(axis=0) if rows.any() else np.zeros([self.O], F32)
            idx = np.argwhere(pairs.any(axis=1))[:, 0]
        return agg.astype(F32), idx.astype(np.int64), pairs

    # reference: utils/coma.py:441-487
    def nonphysical(self, n_bin=1e6):
        self.normalize()
        out = []
        for P in (self.P_h_wrt_o, self.P_o_wrt_h):
            q = (np.round(P * F32(n_bin)) / F32(n_bin)).astype(F32)
            with np.errstate(divide="ignore", invalid="ignore"):
                plogp = np.where(q == 0, F32(0), q * np.log(q))
            s = plogp.sum(axis=-1).astype(F32)
            s = s / F32(math.log(n_bin)) + F32(1.0)
            out.append(s.astype(F32))
        return out[0], out[1]


# --------------------------------------------------------------------------------------------
# A.7  occupancy                                     reference: utils/coma_occupancy.py:160-312
# --------------------------------------------------------------------------------------------
GRIDSIZE = 2.4   # hard-coded at utils/coma_occupancy.py:220


def voxel_centers(gridsize: float, R: int, center=(0, 0, 0)):
    """-> (centers f64 [3,R,R,R], index grid i64 [3,R,R,R], voxel_size, start_point)."""
    voxel = gridsize / R
    start = np.array(center) - np.array([gridsize / 2] * 3)
    idx = np.mgrid[0:R, 0:R, 0:R]
    centers = start.reshape(3, 1, 1, 1) + voxel * idx.astype(F32) + voxel / 2
    return centers, idx, voxel, start


class OccupancyOracle:
    def __init__(self, human_res, spatial_res, scale_tolerance=3.0):
        self.H, self.R = int(human_res), int(spatial_res)
        self.centers, self.idx, self.voxel, self.start = voxel_centers(GRIDSIZE, self.R)
        self.thres = self.voxel * scale_tolerance                                # f64
        self.occ = np.zeros([self.H, self.R, self.R, self.R], F32)
        self.used_count = 0

    def aggregate_sample(self, human_verts, obj_verts, h_chunk=64):
        hv = np.asarray(human_verts)
        q = (hv - np.asarray(obj_verts)[0][None]).astype(F32)                    # f64 subtract, cast f32
        for h0 in range(0, self.H, h_chunk):
            qq = q[h0:h0 + h_chunk].astype(F64)[:, :, None, None, None]
            d2 = np.square(self.centers[None] - qq)
            d = np.sqrt((d2[:, 0] + d2[:, 1]) + d2[:, 2])
            self.occ[h0:h0 + h_chunk] += (d < self.thres).astype(F32)
        self.used_count += 1

    def aggregate_windowed(self, q, margin=2):
        """q: f32 [S,H,3] already relative to the object point.  The SAME predicate as aggregate_sample, evaluated only on the
        cube of cells around each point that can satisfy it (|centre - q| < thres needs every axis within thres; `margin` extra
        cells each side) -- what makes S = 2000 samples at R = 128 checkable on a CPU.  Equality with the dense evaluation is
        asserted in tests/test_oracle_golden.py (CPU)."""
        q = np.asarray(q, F32)
        S, H = q.shape[:2]
        assert H == self.H
        ax = [self.centers[0, :, 0, 0], self.centers[1, 0, :, 0], self.centers[2, 0, 0, :]]      # f64 axis centres
        half = int(np.ceil(self.thres / self.voxel)) + margin
        W = 2 * half + 1
        qq = q.astype(F64).reshape(S * H, 3)
        base = np.floor((qq - self.start[None]) / self.voxel).astype(np.int64) - half                # [P,3] first cell of the window
        off = np.arange(W)
        rows = np.tile(np.arange(H), S)
        for p0 in range(0, S * H, 4096):
            b = base[p0:p0 + 4096]
            idx = b[:, :, None] + off[None, None, :]                                                 # [p,3,W]
            ok = (idx >= 0) & (idx < self.R)
            ic = np.clip(idx, 0, self.R - 1)
            d2 = [np.square(ax[k][ic[:, k]] - qq[p0:p0 + 4096, k, None]) for k in range(3)]          # [p,W] each
            d = np.sqrt((d2[0][:, :, None, None] + d2[1][:, None, :, None]) + d2[2][:, None, None, :])
            hit = (d < self.thres) & ok[:, 0, :, None, None] & ok[:, 1, None, :, None] & ok[:, 2, None, None, :]
            pi, xi, yi, zi = np.nonzero(hit)
            np.add.at(self.occ, (rows[p0:p0 + 4096][pi], ic[pi, 0, xi], ic[pi, 1, yi], ic[pi, 2, zi]), F32(1.0))
        self.used_count += S

    def aggregated_grid(self):
        flat = self.occ.reshape(self.H, -1)
        with np.errstate(divide="ignore", invalid="ignore"):
            flat = flat / flat.sum(axis=-1, keepdims=True)
        self.occ = flat.reshape(self.occ.shape)
        # torch.max propagates NaN (a human vertex never inside the grid -> 0/0)
        out = self.occ.max(axis=0)
        return np.where(np.isnan(self.occ).any(axis=0), F32(np.nan), out).astype(F32)


# --------------------------------------------------------------------------------------------
# A.8  nearest-vertex index map                       reference: utils/coma.py:87-96
# --------------------------------------------------------------------------------------------
def nearest_vertex(points: np.ndarray, verts: np.ndarray, chunk=256) -> np.ndarray:
    """points [P,3], verts [V,3] f64 -> i64 [P]; first minimum wins ties (np.argmin)."""
    points, verts = np.asarray(points, F64), np.asarray(verts, F64)
    out = np.empty(len(points), np.int64)
    for i0 in range(0, len(points), chunk):
        d = np.square(points[None, i0:i0 + chunk, :] - verts[:, None, :])
        out[i0:i0 + chunk] = np.argmin((d[..., 0] + d[..., 1]) + d[..., 2], axis=0)
    return out


# --------------------------------------------------------------------------------------------
# parity metrics                       reference: utils/evaluation.py:4-12
# --------------------------------------------------------------------------------------------
def max_rel_err(x, ref, floor=1e-6):
    ref = np.asarray(ref, F64)
    x = np.asarray(x, F64)
    if ref.size == 0:
        return 0.0
    if x.shape != ref.shape or (np.isnan(x) != np.isnan(ref)).any():
        return float("inf")
    top = np.nanmax(np.abs(ref)) if np.isfinite(ref).any() else 0.0
    if top == 0.0:                       # all-zero reference: absolute error
        return float(np.nanmax(np.abs(x - ref))) if np.isfinite(ref).any() else 0.0
    den = np.maximum(np.abs(ref), floor * top)
    return float(np.nanmax(np.abs(x - ref) / den))


def mae_normalised(x, ref):
    x, ref = np.asarray(x, F64).ravel(), np.asarray(ref, F64).ravel()
    return float(np.mean(np.abs(x / x.sum() - ref / ref.sum())))


# --------------------------------------------------------------------------------------------
# vertex normals (sample ingestion)                  reference: utils/coma.py:672-686 via open3d
# --------------------------------------------------------------------------------------------
def vertex_normals(verts: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """Area-weighted vertex normals, f64: un-normalised triangle cross products summed per vertex in face order, then
    normalised (zero -> (0,0,1)).  This is open3d's ComputeVertexNormals as published; open3d itself is absent from this
    image, so this function is NOT pinned against it ("parity unpinned" at this boundary)."""
    v = np.asarray(verts, F64)
    f = np.asarray(faces, np.int64)
    a, b = v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]
    tn = np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2], a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], -1)
    n = np.zeros_like(v)
    for k in range(3):
        np.add.at(n, f[:, k], tn)      # sequential in face order per column; per-vertex order = ascending face index
    # NB: np.add.at(column k) interleaves differently from the per-face loop of open3d (which adds to its three vertices
    # face by face); a vertex never appears twice in one face, so each vertex still receives its faces in ascending order
    # within a column, but columns are summed one after the other -> use the exact per-vertex order instead:
    n = np.zeros_like(v)
    order = np.lexsort((np.repeat(np.arange(len(f)), 3), f.reshape(-1)))
    vid, fid = f.reshape(-1)[order], np.repeat(np.arange(len(f)), 3)[order]
    for vi, fi in zip(vid, fid):
        n[vi] += tn[fi]
    nrm = np.sqrt((n[:, 0] ** 2 + n[:, 1] ** 2) + n[:, 2] ** 2)
    out = np.where(nrm[:, None] > 0, n / np.where(nrm > 0, nrm, 1.0)[:, None], np.array([0.0, 0.0, 1.0]))
    return out


def orientation_and_contact_targets(info, reference_object_vertex_index, contact_threshold):
    """src/application/optimize.py:190-196, restated line by line (NumPy on the exported dict)."""
    grid_prob = info["prob_grid_canon_human_wrt_obj"][:, reference_object_vertex_index, :]
    max_prob_indices = np.argmax(grid_prob, axis=1)
    relative_orientation_GT = np.array([info["canon_normal_grid"][i].reshape((3,)) for i in max_prob_indices])
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = info["contact_dist_expectation_grid_nom"] / info["contact_dist_expectation_grid_denom"]
    selected_human_indices = np.nonzero(np.max(ratio, axis=1) > contact_threshold)
    corresponding_object_indices = np.argmax(info["contact_dist_expectation_grid_nom"][selected_human_indices], axis=1)
    return max_prob_indices, relative_orientation_GT, selected_human_indices, corresponding_object_indices
