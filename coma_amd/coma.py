"""ComA accumulator + reducers on MI355X: host mirror of the reference's ``utils/coma.py`` API.

Same class / method / attribute names, argument meaning, error behaviour and pickle schema as the
reference, so ``src/coma/extract_coma.py`` / ``inference.py`` and pickles written by
either side keep working.  All arithmetic runs in libcoma_hip.so (hand-written gfx950 kernels) through
the C ABI in include/coma_hip.h; there is NO CPU fallback -- a missing library or a non-HIP device raises.

Reference map (file:line in the reference repo):
  get_uniform_points_on_sphere  utils/coma.py:18-26      negative_exp            utils/coma.py:116-119
  ComA.__init__                 utils/coma.py:177-251    register/aggregate      utils/coma.py:253-277
  aggregate (K1-K3)             utils/coma.py:279-323    normalize               utils/coma.py:328-330
  compute_contact_map           utils/coma.py:333-366    significant pairs       utils/coma.py:369-383
  aggregate_contact_for_...     utils/coma.py:385-438    nonphysical response    utils/coma.py:441-487
  export / load                 utils/coma.py:582-610    get_aggregated_contact  utils/coma.py:614-641
  simplify_mesh_and_get_indices utils/coma.py:29-98 (distance branch :87-96 -> coma_nearest_vertex_i64)
"""
from __future__ import annotations

import pickle
from copy import deepcopy
from functools import partial

import numpy as np
import torch

from . import _lib
from .misc import to_np_torch_recursive

# how many samples are staged on the device per kernel launch (inputs are 24*(H+O) bytes per sample)
MAX_SAMPLES_PER_LAUNCH = 512


def get_uniform_points_on_sphere(num_points=1000):
    """Fibonacci sphere, f64; bin k of the orientation histograms."""
    idx = np.arange(0, num_points, dtype=float) + 0.5
    phi = np.arccos(1 - 2 * idx / num_points)
    theta = np.pi * (1 + 5**0.5) * idx
    return np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)


def negative_exp(x, spatial_grid_size, spatial_grid_thres, **kwargs):
    """Proximity score exp(-d/size).  Kept as a module-level function because the ComA pickle stores a
    functools.partial of it (the fused kernel evaluates the same expression on device)."""
    return torch.exp(-x / spatial_grid_size)


# pickles written by the reference name this function ``utils.coma.negative_exp``; keep ours loadable there
negative_exp.__module__ = "utils.coma"


def _as_f32(a, shape, name):
    a = np.ascontiguousarray(np.asarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a), dtype=np.float32)
    assert a.ndim == 2 and a.shape[-1] == 3 and len(a) == shape, f"{name}: expected [{shape},3], got {a.shape}"
    return a


class ComA:
    _STATE_KEYS = ("prob_grid_canon_human_wrt_obj", "prob_grid_canon_obj_wrt_human", "contact_dist_expectation_grid_nom",
                   "contact_dist_expectation_grid_denom", "significant_contact_count")

    def __init__(self, human_res: int, obj_res: int, normal_res: int, spatial_res: int, proximity_settings=dict(),
                 principle_vec=[0, 0, 1], sub_principle_vec=[0, 1, 0], rel_dist_method: str = "dist",
                 normal_gaussian_sigma: float = 0.1, eps: float = 1e-8, device: str = "cuda"):
        self.device = device
        self.human_res = human_res
        self.obj_res = obj_res
        self.normal_res = normal_res
        self.spatial_res = spatial_res

        x, y, z = get_uniform_points_on_sphere(num_points=normal_res)
        self.canon_normal_grid = torch.tensor(np.stack([x, y, z], axis=-1)).to(device)   # f64 while learning

        if self.spatial_res == 0:
            H, O, N = self.human_res, self.obj_res, self.normal_res
            self.prob_grid_canon_human_wrt_obj = torch.zeros([H, O, N], dtype=torch.float32, device=device)
            self.prob_grid_canon_obj_wrt_human = torch.zeros([H, O, N], dtype=torch.float32, device=device)
            self.contact_dist_expectation_grid_nom = torch.zeros([H, O], dtype=torch.float32, device=device)
            self.contact_dist_expectation_grid_denom = torch.zeros([H, O], dtype=torch.float32, device=device)
            self.significant_contact_count = torch.zeros([H, O], dtype=torch.float32, device=device)
        else:
            print("Please implement the spatial grid")
            raise NotImplementedError

        self.proximity_settings = proximity_settings
        self.contact_dist_func = partial(negative_exp, **proximity_settings)
        self.cross_contact_scores_nom = torch.zeros([human_res, obj_res], dtype=torch.float32, device=device)
        self.cross_contact_scores_denom = torch.zeros([human_res, obj_res], dtype=torch.float32, device=device)

        self.cache_count = 0
        self.used_count = 0
        self.cache = dict()
        self.used = dict()

        self.principle_vec = torch.tensor(principle_vec, dtype=torch.float32).to(device)
        self.sub_principle_vec = torch.tensor(sub_principle_vec, dtype=torch.float32).to(device)

        assert rel_dist_method in ["dist", "sdf"], f"rel_dist_method: '{rel_dist_method}' not allowed"
        self.rel_dist_method = rel_dist_method
        self.normal_gaussian_sigma = normal_gaussian_sigma
        self.eps = eps

    # ------------------------------------------------------------------ sample cache
    def register_sample_to_cache(self, **kwargs):
        self.cache[f"{self.cache_count:05}"] = kwargs
        self.cache_count = len(self.cache.keys())

    def aggregate_all_samples(self):
        """All cached samples in as few launches as possible (the reference loops one sample at a time)."""
        keys = list(self.cache.keys())
        for k in keys:
            self.assert_inputs(**self.cache[k])
        for i0 in range(0, len(keys), MAX_SAMPLES_PER_LAUNCH):
            self._accumulate([self.cache[k] for k in keys[i0:i0 + MAX_SAMPLES_PER_LAUNCH]])
        for k in keys:
            self.used[f"{self.used_count:05}"] = self.cache[k]
            self.used_count = len(self.used.keys())
        self.cache = {}
        self.cache_count = 0

    def aggregate_single_sample(self, **kwargs):
        if self.spatial_res == 0:
            self.assert_inputs(**kwargs)
            self.aggregate_single_sample_for_contact(**kwargs)
        else:
            print("Please implement the spatial grid and aggregation in spatial grid")
            raise NotImplementedError

    def aggregate_single_sample_for_contact(self, human_verts, human_normals, obj_verts, obj_normals, **kwargs):
        self._accumulate([dict(human_verts=human_verts, human_normals=human_normals, obj_verts=obj_verts,
                               obj_normals=obj_normals)])

    def _accumulate(self, samples):
        if self.rel_dist_method == "sdf":
            raise NotImplementedError
        if not samples:
            return
        H, O, S = self.human_res, self.obj_res, len(samples)
        hv = np.stack([_as_f32(s["human_verts"], H, "human_verts") for s in samples])
        hn = np.stack([_as_f32(s["human_normals"], H, "human_normals") for s in samples])
        first = samples[0]
        def host(a):          # tensors of any device -> NumPy (the reference takes both through to_np_torch_recursive)
            return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        shared_obj = all((s["obj_verts"] is first["obj_verts"] and s["obj_normals"] is first["obj_normals"])
                         or (np.array_equal(host(s["obj_verts"]), host(first["obj_verts"]))
                             and np.array_equal(host(s["obj_normals"]), host(first["obj_normals"])))
                         for s in samples[1:])
        if shared_obj:
            ov = _as_f32(first["obj_verts"], O, "obj_verts")
            on = _as_f32(first["obj_normals"], O, "obj_normals")
        else:
            ov = np.stack([_as_f32(s["obj_verts"], O, "obj_verts") for s in samples])
            on = np.stack([_as_f32(s["obj_normals"], O, "obj_normals") for s in samples])
        dev = self.device
        d_hv, d_hn = torch.from_numpy(hv).to(dev), torch.from_numpy(hn).to(dev)
        d_ov, d_on = torch.from_numpy(ov).to(dev), torch.from_numpy(on).to(dev)
        self.accumulate_device(d_hv, d_hn, d_ov, d_on)

    def accumulate_device(self, human_verts, human_normals, obj_verts, obj_normals):
        """Device-resident entry: f32 [S,H,3] x2, f32 [S,O,3] or [O,3] x2 (already on the HIP device).

        Adds the S samples into the five state tensors; does not touch cache/used bookkeeping.
        """
        L = _lib.lib()
        H, O, N = self.human_res, self.obj_res, self.normal_res
        S = human_verts.shape[0]
        assert tuple(human_verts.shape) == (S, H, 3) and tuple(human_normals.shape) == (S, H, 3)
        shared = obj_verts.dim() == 2
        assert tuple(obj_verts.shape) == ((O, 3) if shared else (S, O, 3)) and obj_normals.shape == obj_verts.shape
        grid32 = self.canon_normal_grid.to(torch.float32).contiguous()
        f32 = torch.float32
        rc = L.coma_contact_accumulate_f32(
            _lib.ptr(human_verts, f32, "human_verts"), _lib.ptr(human_normals, f32, "human_normals"),
            _lib.ptr(obj_verts, f32, "obj_verts"), _lib.ptr(obj_normals, f32, "obj_normals"),
            0 if shared else 3 * O, _lib.ptr(grid32, f32), S, H, O, N,
            _lib.vec3(self.principle_vec.tolist()), _lib.vec3(self.sub_principle_vec.tolist()),
            float(self.proximity_settings["spatial_grid_size"]), float(self.proximity_settings["spatial_grid_thres"]),
            float(self.normal_gaussian_sigma), float(self.eps),
            _lib.ptr(self.prob_grid_canon_human_wrt_obj, f32, "prob_grid_canon_human_wrt_obj"),
            _lib.ptr(self.prob_grid_canon_obj_wrt_human, f32, "prob_grid_canon_obj_wrt_human"),
            _lib.ptr(self.contact_dist_expectation_grid_nom, f32), _lib.ptr(self.contact_dist_expectation_grid_denom, f32),
            _lib.ptr(self.significant_contact_count, f32), _lib.stream_ptr(human_verts.device))
        _lib.check(rc, "coma_contact_accumulate_f32")

    # ------------------------------------------------------------------ reducers
    def _contact_map_call(self, prob, want_contact):
        L = _lib.lib()
        H, O, N = self.human_res, self.obj_res, self.normal_res
        f32 = torch.float32
        grid32 = self.canon_normal_grid.to(f32).contiguous()
        out = torch.empty([H, O], dtype=f32, device=prob.device) if want_contact else None
        rc = L.coma_contact_map_f32(_lib.ptr(prob, f32, "prob grid"), _lib.ptr(grid32, f32),
                                    _lib.vec3(self.principle_vec.tolist()),
                                    _lib.ptr(self.contact_dist_expectation_grid_nom, f32),
                                    _lib.ptr(self.contact_dist_expectation_grid_denom, f32),
                                    H * O, N, float(self.eps), _lib.ptr(out, f32), _lib.stream_ptr(prob.device))
        _lib.check(rc, "coma_contact_map_f32")
        return out

    def normalize_prob_grid_for_normals(self):
        self._contact_map_call(self.prob_grid_canon_human_wrt_obj, False)
        self._contact_map_call(self.prob_grid_canon_obj_wrt_human, False)

    def compute_contact_map(self, contact_map_type: str, as_numpy: bool = True):
        self.assert_inputs(contact_map_type=contact_map_type)
        # the reference normalises BOTH grids in place before anything else (utils/coma.py:339)
        on_h = self._contact_map_call(self.prob_grid_canon_human_wrt_obj, contact_map_type in ["human", "both"])
        on_o = self._contact_map_call(self.prob_grid_canon_obj_wrt_human, contact_map_type in ["obj", "both"])
        contact_map_dict = {"human": on_h, "obj": on_o}
        if as_numpy:
            return to_np_torch_recursive(contact_map_dict, use_torch=False, device="cpu")
        return contact_map_dict

    def _pairs(self, ratio):
        L = _lib.lib()
        H, O = self.human_res, self.obj_res
        dev = self.significant_contact_count.device
        pairs = torch.empty([H, O], dtype=torch.uint8, device=dev)
        col_any = torch.empty([O], dtype=torch.uint8, device=dev)
        row_any = torch.empty([H], dtype=torch.uint8, device=dev)
        thr = float(np.float32(ratio * self.used_count))   # Python double product, compared in f32
        rc = L.coma_significant_pairs_u8(_lib.ptr(self.significant_contact_count, torch.float32), thr, H, O,
                                         _lib.ptr(pairs), _lib.ptr(col_any), _lib.ptr(row_any), _lib.stream_ptr(dev))
        _lib.check(rc, "coma_significant_pairs_u8")
        return pairs, col_any, row_any

    def significant_contact_pairs(self, significant_contact_ratio: float, as_numpy: bool = True):
        pairs = self._pairs(significant_contact_ratio)[0].to(torch.bool)
        if as_numpy:
            return to_np_torch_recursive(pairs, use_torch=False, device="cpu")
        return pairs

    def aggregate_contact_for_significant_pairs(self, contact_map_dict: dict, contact_map_type: str,
                                                significant_contact_ratio: float, as_numpy: bool = True):
        self.assert_inputs(contact_map_type=contact_map_type)
        L = _lib.lib()
        H, O = self.human_res, self.obj_res
        pairs, col_any, row_any = self._pairs(significant_contact_ratio)
        out = {"human": None, "obj": None}
        for which, code, n in (("human", 0, H), ("obj", 1, O)):
            if contact_map_type in [which, "both"]:
                cm = contact_map_dict[which]
                assert cm is not None, f"If 'contact_map_type' is '{which}' or 'both', contact_map_dict['{which}'] must not be None"
                cm = cm.to(torch.float32).contiguous()
                res = torch.empty([n], dtype=torch.float32, device=cm.device)
                rc = L.coma_masked_max_f32(_lib.ptr(cm, torch.float32), _lib.ptr(col_any), _lib.ptr(row_any), H, O, code,
                                           _lib.ptr(res), _lib.stream_ptr(cm.device))
                _lib.check(rc, "coma_masked_max_f32")
                out[which] = res
        result = {"human": out["human"], "obj": out["obj"], "significant_contact_pairs": pairs.to(torch.bool)}
        if as_numpy:
            return to_np_torch_recursive(result, use_torch=False, device="cpu")
        return result

    def compute_nonphysical_response_sphere(self, n_bin: int, nonphysical_type: str, as_numpy: bool = True):
        self.assert_inputs(nonphysical_type=nonphysical_type)
        L = _lib.lib()
        H, O, N = self.human_res, self.obj_res, self.normal_res
        f32 = torch.float32
        scores = {"human": None, "obj": None}
        for which, prob in (("human", self.prob_grid_canon_human_wrt_obj), ("obj", self.prob_grid_canon_obj_wrt_human)):
            if nonphysical_type in [which, "both"]:
                s = torch.empty([H, O], dtype=f32, device=prob.device)
                rc = L.coma_entropy_f32(_lib.ptr(prob, f32), H * O, N, float(self.eps), float(n_bin), _lib.ptr(s, f32),
                                        _lib.stream_ptr(prob.device))
                _lib.check(rc, "coma_entropy_f32")
                scores[which] = s
            else:   # still normalised in place, as in the reference
                self._contact_map_call(prob, False)
        result = {"human": scores["human"], "obj": scores["obj"], "n_bin": n_bin}
        if as_numpy:
            return to_np_torch_recursive(result, use_torch=False, device="cpu")
        return result

    # ------------------------------------------------------------------ checks (utils/coma.py:489-526)
    def assert_inputs(self, **kwargs):
        for key, n in (("human_verts", self.human_res), ("human_normals", self.human_res),
                       ("obj_verts", self.obj_res), ("obj_normals", self.obj_res)):
            if key in kwargs:
                a = kwargs[key]
                assert a.ndim == 2
                assert a.shape[-1] == 3
                assert len(a) == n
        if "contact_map_type" in kwargs:
            assert kwargs["contact_map_type"] in ["human", "obj", "both"], \
                "Only ['human'/'obj'/'both'] allowed for Argument: 'contact_map_type'"
        if "nonphysical_type" in kwargs:
            assert kwargs["nonphysical_type"] in ["human", "obj", "both"], \
                "Only ['human'/'obj'/'both'] allowed for Argument: 'nonphysical_type'"

    # ------------------------------------------------------------------ checkpoint
    def export(self, save_pth=None):
        import utils.coma  # noqa: F401  (makes utils.coma.negative_exp resolvable for pickle)
        to_export = {k: v for k, v in vars(self).items() if k not in ("cache", "used") and not k.startswith("_")}
        to_export = {k: (v.detach().clone() if isinstance(v, torch.Tensor) else deepcopy(v)) for k, v in to_export.items()}
        to_export = to_np_torch_recursive(to_export, use_torch=False, device="cpu")
        if save_pth is None:
            return to_export
        with open(save_pth, "wb") as handle:
            pickle.dump(to_export, handle, protocol=pickle.HIGHEST_PROTOCOL)

    def load(self, load_pth):
        with open(load_pth, "rb") as handle:
            loadables = pickle.load(handle)
        loadables = to_np_torch_recursive(loadables, use_torch=True, device=self.device)
        for k, v in loadables.items():
            setattr(self, k, v)

    # ------------------------------------------------------------------ multi-GPU
    def all_reduce(self, group=None):
        """Sum the partial state of all ranks (RCCL over xGMI); every rank ends with the global ComA."""
        from .dist import all_reduce_coma
        all_reduce_coma(self, group)


def get_aggregated_contact(coma: ComA, contact_map_type: str, significant_contact_ratio: float):
    assert contact_map_type in ["human", "obj"]
    contact_map_dict = coma.compute_contact_map(contact_map_type=contact_map_type, as_numpy=False)
    agg = coma.aggregate_contact_for_significant_pairs(contact_map_dict=contact_map_dict, contact_map_type=contact_map_type,
                                                       significant_contact_ratio=significant_contact_ratio, as_numpy=True)
    aggregated_contact = agg[contact_map_type]
    pairs = agg["significant_contact_pairs"]
    # NB (reference quirk, kept): for "human" the index vector lists OBJECT columns, for "obj" human rows
    indicator = np.any(pairs, axis=0 if contact_map_type == "human" else 1)
    return aggregated_contact, np.argwhere(indicator)[:, 0]


def get_nonphysical_score(coma: ComA, nonphysical_type: str):
    return coma.compute_nonphysical_response_sphere(n_bin=1e6, nonphysical_type=nonphysical_type, as_numpy=True)[nonphysical_type]


def nearest_vertex_indices(points, mesh_verts, device="cuda"):
    """argmin_v |p_i - m_v|^2 in f64 on the device -> list of python ints (duplicates kept)."""
    L = _lib.lib()
    pts = torch.tensor(np.ascontiguousarray(np.asarray(points, dtype=np.float64)), device=device)
    vts = torch.tensor(np.ascontiguousarray(np.asarray(mesh_verts, dtype=np.float64)), device=device)
    assert pts.dim() == 2 and pts.shape[1] == 3 and vts.dim() == 2 and vts.shape[1] == 3
    idx = torch.empty([pts.shape[0]], dtype=torch.int64, device=pts.device)
    rc = L.coma_nearest_vertex_i64(_lib.ptr(pts, torch.float64), _lib.ptr(vts, torch.float64), pts.shape[0], vts.shape[0],
                                   _lib.ptr(idx, torch.int64), _lib.stream_ptr(pts.device))
    _lib.check(rc, "coma_nearest_vertex_i64")
    return idx.cpu().numpy()


def simplify_mesh_and_get_indices(mesh, number_of_points: int, simplify_method="poisson_disk",
                                  mesh_index_find_method="distance-based", debug=False, device="cuda"):
    """Down-sample a mesh to a point cloud (open3d, third party) and map each point to its nearest vertex.

    Only the distance branch of the reference is on the accelerated path; the ray-casting branch of the
    reference drops into an interactive shell and is not reproduced.
    """
    if simplify_method == "poisson_disk":
        pcd = mesh.sample_points_poisson_disk(number_of_points=number_of_points)
    elif simplify_method == "uniform":
        pcd = mesh.sample_points_uniformly(number_of_points=number_of_points)
    else:
        raise NotImplementedError
    if mesh_index_find_method != "distance-based":
        raise NotImplementedError
    idx = nearest_vertex_indices(np.asarray(pcd.points), np.asarray(mesh.vertices), device=device)
    return list(idx), pcd
