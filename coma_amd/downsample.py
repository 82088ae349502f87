"""Mesh -> down-sampled point set + nearest-vertex index map, in the pickle schema the ComA stage reads.

Mirrors the two writers of the reference, `src/coma/downsample_human.py:17-77` and `src/coma/downsample_objects.py:17-60`
(keys, dtypes and the two zero-normal filters), with the open3d pieces replaced:
  * nearest vertex  -> coma_nearest_vertex_i64 (bit-exact argmin, utils/coma.py:87-91);
  * vertex normals  -> coma_vertex_normals_f64 (area-weighted, ascending-face accumulation like open3d; parity unpinned --
    open3d is absent from the build image);
  * the point sampler: open3d's Poisson-disk elimination is third party and out of scope; points are either
    SUPPLIED (what a maintainer with open3d exports once) or drawn by the seeded area-weighted uniform sampler below.
"""
from __future__ import annotations

import numpy as np

from .coma import nearest_vertex_indices
from .ingest import vertex_normals_batch


def load_obj(pth):
    """Vertices and triangles of a Wavefront OBJ in file order (faces fan-triangulated, v/vt/vn index forms accepted)."""
    verts, faces = [], []
    with open(pth) as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                verts.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                idx = [int(tok.split("/")[0]) for tok in t[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                faces += [[idx[0], idx[k], idx[k + 1]] for k in range(1, len(idx) - 1)]
    return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64)


def sample_uniform(vertices, faces, vertex_normals, number_of_points, seed=0):
    """Seeded area-weighted uniform surface samples with barycentric-interpolated, re-normalised normals."""
    rng = np.random.default_rng(seed)
    a, b, c = (vertices[faces[:, k]] for k in range(3))
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    f = rng.choice(len(faces), size=number_of_points, p=area / area.sum())
    r1, r2 = np.sqrt(rng.random(number_of_points)), rng.random(number_of_points)
    w = np.stack([1 - r1, r1 * (1 - r2), r1 * r2], axis=1)
    pts = (w[:, :, None] * vertices[faces[f]]).sum(1)
    nrm = (w[:, :, None] * vertex_normals[faces[f]]).sum(1)
    n = np.linalg.norm(nrm, axis=1, keepdims=True)
    return pts, np.divide(nrm, n, out=np.zeros_like(nrm), where=n > 0)


def _points(vertices, faces, normals, number_of_points, points, point_normals, simplify_method, seed):
    if points is not None:
        points = np.asarray(points, dtype=np.float64)
        assert point_normals is not None and len(point_normals) == len(points), "supplied points need their normals"
        return points, np.asarray(point_normals, dtype=np.float64)
    if simplify_method != "uniform":
        raise NotImplementedError("Poisson-disk sampling is open3d's (third party): pass points=/point_normals= exported from it, "
                                  "or use simplify_method='uniform'")
    return sample_uniform(vertices, faces, normals, number_of_points, seed)


def downsample_human(vertices, faces, number_of_points, points=None, point_normals=None, simplify_method="uniform", seed=42, device="cuda"):
    """downsample_human.py:29-77 -> the dict it pickles as smplx_star_downsampled_{N}.pickle."""
    vertices, faces = np.asarray(vertices), np.asarray(faces).astype(np.int64)
    V = len(vertices)
    normals = vertex_normals_batch(vertices, faces, device=device)[0]
    if number_of_points < V:
        pts, nrm = _points(vertices.astype(np.float64), faces, normals, number_of_points, points, point_normals, simplify_method, seed)
        indices = [int(i) for i in nearest_vertex_indices(pts, vertices.astype(np.float64), device=device)]
    else:
        pts, nrm, indices = vertices.astype(np.float64), normals, list(range(V))
    indices = [i for i in indices if normals[i].sum() != 0]          # vertices without a normal are skipped (:58-65)
    return {"vertices": vertices, "faces": faces, "V": V, "F": faces.shape[0], "N": len(indices), "N_raw": len(pts),
            "downsample_indices": indices, "downsampled_pcd_points_raw": pts, "downsampled_pcd_normal_raw": nrm}


def downsample_object(supercategory, category, asset_id, vertices, faces, number_of_points, points=None, point_normals=None,
                      simplify_method="uniform", seed=42, device="cuda"):
    """downsample_objects.py:17-62 -> the dict it pickles as {asset_id}_{N}.pickle."""
    vertices, faces = np.asarray(vertices, dtype=np.float64), np.asarray(faces).astype(np.int64)
    normals = vertex_normals_batch(vertices, faces, device=device)[0]
    pts, nrm = _points(vertices, faces, normals, number_of_points, points, point_normals, simplify_method, seed)
    indices = [int(i) for i in nearest_vertex_indices(pts, vertices, device=device)]
    keep = np.array([d for d in range(len(nrm)) if nrm[d].sum() != 0], dtype=np.int64)   # zero-normal samples are dropped (:30-38)
    return {"supercategory": supercategory, "category": category, "asset_id": asset_id, "V": vertices.shape[0], "F": faces.shape[0],
            "N": len(indices), "N_raw": len(keep), "downsample_indices": indices, "downsampled_pcd_points_raw": pts[keep],
            "downsampled_pcd_normal_raw": nrm[keep], "obj_vertices_original": vertices, "obj_faces_original": faces,
            "obj_vertex_normals_original": normals}
