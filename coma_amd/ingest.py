"""Sample ingestion for the ComA stage: posed SMPL-X meshes -> (human verts / normals, object points / normals).

Mirrors ``prepare_affordance_extraction_inputs`` (reference utils/coma.py:649-791; same arguments, same returned keys)
and adds the batched device path the reference lacks: the per-sample open3d mesh build + vertex-normal computation
(:672-686) becomes one kernel over all samples of an asset (coma_vertex_normals_f64), followed by the down-sample
gather.  File formats: sample pickle ``{verts [10475,3], faces, IoU, ...}`` or a sentinel ``str``
(src/generation/compute_metrics.py:225-249); down-sample pickles (src/coma/downsample_human.py:67-77,
downsample_objects.py:46-60).
"""
from __future__ import annotations

import pickle

import numpy as np
import torch

from . import _lib
from .misc import normalize_vectors_np


def vertex_face_csr(faces, num_verts):
    """vertex -> incident faces, ascending face index (the accumulation order of open3d's ComputeVertexNormals)."""
    faces = np.asarray(faces, dtype=np.int64)
    F = len(faces)
    vid = faces.reshape(-1)
    fid = np.repeat(np.arange(F, dtype=np.int64), 3)
    order = np.lexsort((fid, vid))
    counts = np.bincount(vid, minlength=num_verts)
    off = np.zeros(num_verts + 1, dtype=np.int32)
    off[1:] = np.cumsum(counts)
    return off, fid[order].astype(np.int32)


def vertex_normals_batch(verts, faces, eps=-1.0, device="cuda"):
    """verts [S,V,3] (any float dtype), faces [F,3] -> unit vertex normals f64 [S,V,3] on the host."""
    verts = np.ascontiguousarray(np.asarray(verts, dtype=np.float64))
    if verts.ndim == 2:
        verts = verts[None]
    S, V, _ = verts.shape
    faces32 = np.ascontiguousarray(np.asarray(faces, dtype=np.int32))
    off, vf = vertex_face_csr(faces32, V)
    d = lambda a: torch.from_numpy(a).to(device)
    dv, df, do, dvf = d(verts), d(faces32), d(off), d(vf)
    out = torch.empty(S, V, 3, dtype=torch.float64, device=device)
    rc = _lib.lib().coma_vertex_normals_f64(_lib.ptr(dv, torch.float64), _lib.ptr(df, torch.int32), _lib.ptr(do, torch.int32),
                                            _lib.ptr(dvf, torch.int32), S, V, len(faces32), float(eps), _lib.ptr(out, torch.float64),
                                            _lib.stream_ptr(out.device))
    _lib.check(rc, "coma_vertex_normals_f64")
    return out.cpu().numpy()


SENTINEL_TYPES = (str,)      # failed samples are pickled strings ("NO HUMANS", "TOO LITTLE INLIERS", ...)


def load_human_sample(pth):
    with open(pth, "rb") as handle:
        data = pickle.load(handle)
    return None if isinstance(data, SENTINEL_TYPES) else data


def prepare_affordance_extraction_inputs(human_mesh_pth, human_mesh_pth_type, human_downsample_metadata, object_downsample_metadata,
                                         human_use_downsample_pcd_raw: bool, object_use_downsample_pcd_raw: bool, eps,
                                         standardize_human_scale: bool, scaler_range, camera_pth, human_params_pth,
                                         object_mesh_for_check_pth=None, interactive=False, device="cuda"):
    if human_mesh_pth_type != "pickle":
        raise NotImplementedError("only the 'pickle' sample format of the pipeline is supported (obj loading needs trimesh)")
    with open(human_mesh_pth, "rb") as handle:
        human_data = pickle.load(handle)
    human_verts_orig = np.asarray(human_data["verts"])
    human_faces_orig = np.asarray(human_data["faces"])
    normals = vertex_normals_batch(human_verts_orig, human_faces_orig, eps=-1.0, device=device)[0]
    human_vertex_normals_orig = normalize_vectors_np(normals, eps=eps)

    obj_verts_orig = object_downsample_metadata["obj_vertices_original"]
    obj_faces_orig = object_downsample_metadata["obj_faces_original"]
    obj_vertex_normals_orig = normalize_vectors_np(object_downsample_metadata["obj_vertex_normals_original"])

    hidx = human_downsample_metadata["downsample_indices"]
    oidx = object_downsample_metadata["downsample_indices"]
    assert not human_use_downsample_pcd_raw, "Human must use 'mesh' for Representation. You'll know why"
    human_verts = human_verts_orig.copy()[hidx]
    human_vertex_normals = human_vertex_normals_orig.copy()[hidx]
    assert len(human_verts) == human_downsample_metadata["N"]
    if object_use_downsample_pcd_raw:
        obj_verts = object_downsample_metadata["downsampled_pcd_points_raw"]
        obj_vertex_normals = object_downsample_metadata["downsampled_pcd_normal_raw"]
        assert len(obj_verts) == object_downsample_metadata["N_raw"]
    else:
        obj_verts = obj_verts_orig.copy()[oidx]
        obj_vertex_normals = obj_vertex_normals_orig.copy()[oidx]
        assert len(obj_verts) == object_downsample_metadata["N"]

    if standardize_human_scale:
        with open(camera_pth, "rb") as handle:
            cam_scale = pickle.load(handle)["scale"]
        with open(human_params_pth, "rb") as handle:
            hp = pickle.load(handle)
        scaler = (512 / cam_scale) * (hp["convert_data"]["z_mean"] / hp["convert_data"]["focals"][0])
        if scaler_range is not None:
            lo, hi = scaler_range
            if scaler < lo or scaler > hi:
                return None

    return dict(human_verts_orig=human_verts_orig, human_faces_orig=human_faces_orig,
                human_vertex_normals_orig=human_vertex_normals_orig, obj_verts_orig=obj_verts_orig, obj_faces_orig=obj_faces_orig,
                obj_vertex_normals_orig=obj_vertex_normals_orig, human_downsample_indices=hidx, object_downsample_indices=oidx,
                human_verts=human_verts, human_vertex_normals=human_vertex_normals, obj_verts=obj_verts,
                obj_vertex_normals=obj_vertex_normals)
