"""Voxel-occupancy accumulator on MI355X: host mirror of the reference's ``utils/coma_occupancy.py``.

Reference map: load_voxelgrid utils/coma_occupancy.py:160-183; ComA_Occupancy.__init__ :190-249;
aggregate_single_sample_for_occupancy :272-295; normalize/return_aggregated_spatial_grids :297-312;
export/load :315-343.  Same names/attributes/pickle keys; the splat and the reducer run in
libcoma_hip.so (coma_occupancy_splat / coma_occupancy_reduce), bit-exact counts.  No CPU fallback.
"""
from __future__ import annotations

import os
import pickle
from copy import deepcopy

import numpy as np
import torch

from . import _lib
from .misc import get_3d_indexgrid_ijk, to_np_torch_recursive

MAX_SAMPLES_PER_LAUNCH = 4096


def load_voxelgrid(gridsize=3.0, resolution=24, center=[0, 0, 0]):
    """Voxel centres [3,R,R,R] (f64), index grid and metadata.  Note: ``voxel_size * index`` is formed
    in f32 (the index grid is cast to f32 and the Python scalar does not promote it) before the f64
    additions -- the centres are therefore not exactly start + voxel*(i+1/2); reproduced on purpose."""
    length_x = length_y = length_z = gridsize
    N_x = N_y = N_z = resolution
    voxel_size = gridsize / resolution
    center = np.array(center)
    start_point = center - np.array([length_x / 2, length_y / 2, length_z / 2])
    indexgrid = get_3d_indexgrid_ijk(N_x, N_y, N_z)
    canon_grid = start_point.reshape(3, 1, 1, 1) + voxel_size * indexgrid.astype(np.float32) + voxel_size / 2
    grid_metadata = dict(length_x=length_x, length_y=length_y, length_z=length_z, N_x=N_x, N_y=N_y, N_z=N_z,
                         start_point=start_point, voxel_size=voxel_size)
    return canon_grid, indexgrid, grid_metadata


class ComA_Occupancy:
    selected_obj_idxs = [0]

    def __init__(self, scale_tolerance: float, human_res: int, obj_res: int, normal_res: int, spatial_res: int,
                 proximity_settings=dict(), principle_vec=[0, 0, 1], sub_principle_vec=[0, 1, 0],
                 rel_dist_method: str = "dist", normal_gaussian_sigma: float = 0.1, selected_obj_idx: int = None,
                 eps: float = 1e-8, device: str = "cuda"):
        self.device = device
        self.human_res = human_res
        self.obj_res = obj_res
        self.normal_res = normal_res
        self.spatial_res = spatial_res
        assert normal_res == 0, "In this version, normal res is 0."

        self.spatial_grid, self.spatial_indexgrid, self.spatial_grid_metadata = load_voxelgrid(
            gridsize=2.4, resolution=self.spatial_res, center=[0, 0, 0])
        self.N_x = self.spatial_grid_metadata["N_x"]
        self.N_y = self.spatial_grid_metadata["N_y"]
        self.N_z = self.spatial_grid_metadata["N_z"]
        self.spatial_grid = torch.from_numpy(self.spatial_grid).to(device)   # f64 [3,R,R,R]

        # Per-vertex counts.  Samples aggregated into a still-pristine grid are only staged on the device (_pending); the
        # first thing that needs the grid -- export, an outside read of the attribute, or return_aggregated_spatial_grids --
        # runs splat + row sums + max over humans as ONE pass that writes the grid once (coma_occupancy_fused, SURVEY.md 8d
        # structure B).  The reference's in-place normalisation by the reducer is applied lazily (_needs_norm) the next time
        # the grid itself is looked at.
        self._grid = torch.zeros([self.human_res, self.N_x, self.N_y, self.N_z], dtype=torch.float32, device=device)
        self._pending = []          # staged q tensors [S,H,3]
        self._pristine = True       # nothing has been accumulated into _grid and nobody outside has seen it
        self._field_all = None      # max over ALL humans computed by the fused pass, valid until the grid is exposed
        self._needs_norm = False    # a reduction has happened: outside readers must see counts / row sums
        self._zero_pending = False  # reset() was called: the grid is logically zero, the memset happens only if somebody needs it
        self.cache_count = 0
        self.used_count = 0
        self.cache = dict()
        self.used = dict()

        self.principle_vec = torch.tensor(principle_vec, dtype=torch.float32).to(device)
        self.sub_principle_vec = torch.tensor(sub_principle_vec, dtype=torch.float32).to(device)

        assert rel_dist_method in ["dist", "sdf"], f"rel_dist_method: '{rel_dist_method}' not allowed"
        self.rel_dist_method = rel_dist_method
        self.rel_dist_thres = self.spatial_grid_metadata["voxel_size"] * scale_tolerance
        self.normal_gaussian_sigma = normal_gaussian_sigma
        self.eps = eps
        self.debug_obj_vert = None
        self.debug_obj_normal = None

    @property
    def spatial_occupancy_grids(self):
        self._materialize()
        self._field_all, self._pristine = None, False      # the caller may modify it in place
        return self._grid

    @spatial_occupancy_grids.setter
    def spatial_occupancy_grids(self, value):
        self._pending, self._pristine, self._field_all, self._needs_norm, self._zero_pending = [], False, None, False, False
        self._grid = value

    def reset(self):
        """Back to the state after construction (extension: lets a long-lived object be re-used).  The 4 H R^3-byte memset is deferred:
        the fused pass writes every cell of the grid anyway; any other reader or writer zeroes it first (_materialize)."""
        self._zero_pending = True
        self._pending, self._pristine, self._field_all, self._needs_norm = [], True, None, False
        self.cache, self.used, self.cache_count, self.used_count = dict(), dict(), 0, 0

    def _materialize(self):
        """Make _grid hold what the reference's attribute would hold right now."""
        if self._zero_pending and not (self._pending and self._pristine and self._fusable()):
            self._grid.zero_()
            self._zero_pending = False
        if self._pending:
            if self._pristine and self._fusable():
                self._field_all = self._fused(None)
            else:
                self._flush()
        if self._needs_norm:
            self._needs_norm = False
            self._classic_reduce(None)

    def register_sample_to_cache(self, **kwargs):
        self.cache[f"{self.cache_count:05}"] = kwargs
        self.cache_count = len(self.cache.keys())

    def aggregate_all_samples(self):
        keys = list(self.cache.keys())
        for i0 in range(0, len(keys), MAX_SAMPLES_PER_LAUNCH):
            self._accumulate([self.cache[k] for k in keys[i0:i0 + MAX_SAMPLES_PER_LAUNCH]])
        for k in keys:
            self.used[f"{self.used_count:05}"] = self.cache[k]
            self.used_count = len(self.used.keys())
        self.cache = {}
        self.cache_count = 0

    def aggregate_single_sample(self, **kwargs):
        self.aggregate_single_sample_for_occupancy(**kwargs)

    def aggregate_single_sample_for_occupancy(self, human_verts, human_normals, obj_verts, obj_normals, **kwargs):
        self._accumulate([dict(human_verts=human_verts, human_normals=human_normals, obj_verts=obj_verts,
                               obj_normals=obj_normals)])

    def _accumulate(self, samples):
        if not samples:
            return
        assert list(self.selected_obj_idxs) == [0], "only object point 0 is supported (as shipped in the reference)"
        def host(a):          # tensors of any device -> NumPy (the reference takes both through to_np_torch_recursive)
            return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        qs = []
        for s in samples:
            obj_vert = host(s["obj_verts"])[0]
            obj_normal = host(s["obj_normals"])[0]
            # the reference asserts that the object point and its normal never change across samples
            if self.debug_obj_vert is None:
                self.debug_obj_vert = obj_vert
            else:
                assert np.allclose(self.debug_obj_vert, obj_vert)
            if self.debug_obj_normal is None:
                self.debug_obj_normal = obj_normal
            else:
                assert np.allclose(self.debug_obj_normal, obj_normal)
            hv = host(s["human_verts"])
            assert hv.shape[0] == self.human_res
            qs.append((hv - obj_vert[None]).astype(np.float32))      # subtract in the input dtype, then f32
        q = torch.from_numpy(np.ascontiguousarray(np.stack(qs))).to(self.device)
        self.accumulate_device(q)

    def _axis_centers(self):
        c = getattr(self, "_centers", None)
        if c is None:
            g = self.spatial_grid.to(torch.float64)
            c = self._centers = torch.stack([g[0, :, 0, 0], g[1, 0, :, 0], g[2, 0, 0, :]]).contiguous()
        return c

    def accumulate_device(self, q, lazy=True):
        """q: f32 [S,H,3] on the HIP device, already relative to object point 0."""
        assert tuple(q.shape[1:]) == (self.human_res, 3)
        if lazy and self._pristine and self._fusable(extra=q.shape[0]):
            self._pending.append(q)
            return
        self._materialize()
        self._field_all = None
        self._splat(q)

    def _window(self):
        """Candidate cells per axis a sample can reach (what coma_occupancy_fused is told; it rounds up to a power of two)."""
        voxel, thres = float(self.spatial_grid_metadata["voxel_size"]), float(self.rel_dist_thres)
        return int(np.ceil(2.0 * thres / voxel - 1e-9)) + 2

    def _fusable(self, extra=0):
        """Can the fused pass take the staged samples (+ `extra` more)?  Mirrors every size check of coma_occupancy_fused:
        cubic grid, R*R % 4 == 0, an x-plane fits the LDS slab, window <= 16 cells (scale_tolerance is a free CLI float) and
        fewer than 65536 samples per call (16-bit counters)."""
        R = self.spatial_res
        w = self._window()
        w2 = w if w <= 2 else 1 << (w - 1).bit_length()
        staged = sum(int(p.shape[0]) for p in self._pending) + int(extra)
        return (self.N_x == self.N_y == self.N_z == R and (R * R) % 4 == 0 and R * R <= 20480 and R <= 255 and w2 <= 16
                and staged < 65536)

    def _flush(self):
        pend, self._pending = self._pending, []
        for q in pend:
            self._splat(q)

    def _splat(self, q):
        self._pristine = False
        L = _lib.lib()
        S, H, R = q.shape[0], self.human_res, self.spatial_res
        assert tuple(q.shape) == (S, H, 3) and self.N_x == self.N_y == self.N_z == R
        centers = self._axis_centers()
        rc = L.coma_occupancy_splat(_lib.ptr(q, torch.float32, "q"), S, H, R, _lib.ptr(centers, torch.float64),
                                    float(self.spatial_grid_metadata["voxel_size"]), float(self.rel_dist_thres),
                                    _lib.ptr(self._grid, torch.float32, "spatial_occupancy_grids"), _lib.stream_ptr(q.device))
        _lib.check(rc, "coma_occupancy_splat")

    @staticmethod
    def _sqrt_cut(thres):
        """Smallest double x with sqrt(x) >= thres: `(dx^2+dy^2)+dz^2 < x` is then bit-for-bit `sqrt(..) < thres` (sqrt is
        correctly rounded, hence monotone)."""
        y = np.float64(thres) * np.float64(thres)
        while np.sqrt(y) >= thres:
            y = np.nextafter(y, np.float64(0.0))
        while np.sqrt(y) < thres:
            y = np.nextafter(y, np.float64(np.inf))
        return float(y)

    def _cut(self, thres):
        if getattr(self, "_cut_cache", (None, None))[0] != thres:
            self._cut_cache = (thres, self._sqrt_cut(thres))
        return self._cut_cache[1]

    def _fused(self, sel):
        """All staged samples -> raw counts (written once) + row sums + max of counts / row sums over the selected humans."""
        L = _lib.lib()
        q = self._pending[0] if len(self._pending) == 1 else torch.cat(self._pending, dim=0)
        q = q.contiguous()
        S, H, R = int(q.shape[0]), self.human_res, self.spatial_res
        dev = self._grid.device
        voxel, thres = float(self.spatial_grid_metadata["voxel_size"]), float(self.rel_dist_thres)
        window = self._window()
        nbytes = int(L.coma_occupancy_fused_workspace_bytes(S, H, R, window))
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)      # kept: the next call re-uses it
        rowsum = torch.empty([H], dtype=torch.float32, device=dev)
        out = torch.empty([R, R, R], dtype=torch.float32, device=dev)
        centers = self._axis_centers()
        rc = L.coma_occupancy_fused(_lib.ptr(q, torch.float32, "q"), S, H, R, _lib.ptr(centers, torch.float64), voxel, thres,
                                    self._cut(thres), window, _lib.ptr(sel), 1, _lib.ptr(self._grid, torch.float32),
                                    _lib.ptr(rowsum), _lib.ptr(out), _lib.ptr(ws), nbytes, _lib.stream_ptr(dev))
        _lib.check(rc, "coma_occupancy_fused")
        self._pending, self._pristine = [], False          # only once the pass has been accepted: a refusal loses nothing
        self._zero_pending = False                         # every cell of every row has just been written
        return out

    def _reduce(self, human_indices):
        H = self.human_res
        dev = self._grid.device
        sel = None
        if human_indices is not None:
            sel = torch.zeros([H], dtype=torch.uint8, device=dev)
            sel[torch.as_tensor(list(human_indices), dtype=torch.long, device=dev)] = 1
        if self._pending and self._pristine and self._fusable():
            out = self._fused(sel)
            self._needs_norm = True
            return out
        if sel is None and self._field_all is not None and not self._needs_norm:
            out, self._field_all = self._field_all, None
            self._needs_norm = True
            return out
        self._materialize()
        self._field_all = None
        return self._classic_reduce(sel)

    def reduce_keep_raw(self, human_indices=None, want_raw=True):
        """(raw per-vertex counts, field) of this object's rows: the reduction of return_aggregated_spatial_grids plus the RAW
        grid the reference would have exported before it (used by the row-sharded multi-GPU path, coma_amd/dist.py).  When the
        samples are still staged this is the one fused pass -- the counts it leaves in place are raw until somebody looks at the
        attribute -- otherwise the grid is cloned before the in-place normalisation.  An empty selection gives a -inf field."""
        fused = bool(self._pending) and self._pristine and self._fusable()
        raw = None
        if want_raw and not fused:
            raw = self.spatial_occupancy_grids.clone()
        if human_indices is not None and len(human_indices) == 0:
            self.normalize_prob_grid_for_spatials()
            field = torch.full([self.N_x, self.N_y, self.N_z], float("-inf"), dtype=torch.float32, device=self._grid.device)
        else:
            field = self._reduce(human_indices)
        if want_raw and fused:
            assert self._needs_norm, "the fused pass leaves raw counts"
            raw = self._grid
        return raw, field

    def _classic_reduce(self, sel):
        L = _lib.lib()
        H = self.human_res
        R3 = self.N_x * self.N_y * self.N_z
        g = self._grid
        dev = g.device
        rowsum = torch.empty([H], dtype=torch.float32, device=dev)
        out = torch.empty([self.N_x, self.N_y, self.N_z], dtype=torch.float32, device=dev)
        rc = L.coma_occupancy_reduce(_lib.ptr(g, torch.float32), _lib.ptr(sel), H, R3, _lib.ptr(rowsum), _lib.ptr(out),
                                     _lib.stream_ptr(dev))
        _lib.check(rc, "coma_occupancy_reduce")
        return out

    def normalize_prob_grid_for_spatials(self):
        self._reduce(None)

    def normalize_prob_grid_for_spatials_v2(self):
        self.spatial_occupancy_grids = self.spatial_occupancy_grids / self.used_count

    def return_aggregated_spatial_grids(self, human_indices=None):
        """Normalise every row in place (0/0 -> NaN for a vertex never inside the grid, as in the
        reference) and return max over the selected human vertices, [R,R,R] f32 on the device."""
        return self._reduce(human_indices)

    @staticmethod
    def shard_path(save_pth, rank):
        """`.../key:total.pickle` -> `.../key:total_rank{rank}.pickle` (row-sliced export of a multi-GPU run)."""
        stem, ext = os.path.splitext(save_pth)
        return f"{stem}_rank{rank}{ext}"

    def export(self, save_pth=None, shard=None):
        """The reference's export (utils/coma_occupancy.py:315-330): every public attribute except the caches, tensors as NumPy.
        shard = (rank, world, total_rows) (an addition, SURVEY.md 7 "Memory at config 5"): this object holds human rows
        `shard_slice(total_rows, rank, world)` of a row-sharded run; the SAME keys are written to `shard_path(save_pth, rank)` with
        `spatial_occupancy_grids` = this rank's rows, `human_res` = total_rows, plus one extra key `row_shard` = (rank, world, lo,
        hi) -- no rank ever holds or pickles the full [H, R, R, R] grid (88 GB at config 5).  `load` accepts either form and
        `assemble_shards` rebuilds the single-file dict bit for bit."""
        to_export = {k: v for k, v in vars(self).items() if k not in ("cache", "used") and not k.startswith("_")}
        self._materialize()
        to_export["spatial_occupancy_grids"] = self._grid
        # device tensors are copied by .cpu().numpy() below (no 11 GB device-side clone first); a tensor that already lives on the CPU
        # would be aliased by .numpy(), so it is cloned here
        to_export = {k: ((v.detach().clone() if v.device.type == "cpu" else v) if isinstance(v, torch.Tensor) else deepcopy(v))
                     for k, v in to_export.items()}
        to_export = to_np_torch_recursive(to_export, use_torch=False, device="cpu")
        if shard is not None:
            from .dist import shard_slice
            rank, world, total = (int(x) for x in shard)
            lo, hi = shard_slice(total, rank, world)
            assert hi - lo == self.human_res, f"row shard {rank}/{world} of {total} has {hi - lo} rows, this object {self.human_res}"
            to_export["human_res"] = total
            to_export["row_shard"] = (rank, world, lo, hi)
            if save_pth is not None:
                save_pth = self.shard_path(save_pth, rank)
        if save_pth is None:
            return to_export
        with open(save_pth, "wb") as handle:
            pickle.dump(to_export, handle, protocol=pickle.HIGHEST_PROTOCOL)

    @classmethod
    def shard_files(cls, load_pth):
        """The row-shard files of `load_pth`, ordered by rank, or [] (none / incomplete set)."""
        import glob
        import re
        stem, ext = os.path.splitext(load_pth)
        found = {}
        for f in glob.glob(f"{glob.escape(stem)}_rank*{ext}"):
            m = re.fullmatch(re.escape(stem) + r"_rank(\d+)" + re.escape(ext), f)
            if m:
                found[int(m.group(1))] = f
        return [found[r] for r in range(len(found))] if found and sorted(found) == list(range(len(found))) else []

    @classmethod
    def assemble_shards(cls, files):
        """Row-shard pickles -> the dict a single-process export would have written (same keys, same bits)."""
        parts = []
        for f in files:
            with open(f, "rb") as handle:
                parts.append(pickle.load(handle))
        world = len(parts)
        for r, d in enumerate(parts):
            rank, w, lo, hi = d["row_shard"]
            assert (rank, w) == (r, world), f"{files[r]}: shard {rank}/{w}, expected {r}/{world}"
        out = {k: v for k, v in parts[0].items() if k != "row_shard"}
        out["spatial_occupancy_grids"] = np.concatenate([d["spatial_occupancy_grids"] for d in parts], axis=0)
        assert out["spatial_occupancy_grids"].shape[0] == out["human_res"]
        return out

    def load(self, load_pth, shard=None):
        """The reference's load (:333-343).  If `load_pth` does not exist but its row-shard files do (see export), they are
        re-assembled; with shard = (rank, world) and a shard set written by the same world size only this rank's file is read and
        the object becomes that row shard (human_res = its rows)."""
        if os.path.exists(load_pth):
            with open(load_pth, "rb") as handle:
                loadables = pickle.load(handle)
            if shard is not None:
                from .dist import shard_slice
                lo, hi = shard_slice(int(loadables["human_res"]), int(shard[0]), int(shard[1]))
                loadables["spatial_occupancy_grids"] = loadables["spatial_occupancy_grids"][lo:hi]
                loadables["human_res"] = hi - lo
        else:
            files = self.shard_files(load_pth)
            if not files:
                raise FileNotFoundError(f"{load_pth}: neither the file nor a complete set of row shards "
                                        f"({self.shard_path(load_pth, 0)}, ...) exists")
            if shard is not None and int(shard[1]) == len(files):
                with open(files[int(shard[0])], "rb") as handle:
                    loadables = pickle.load(handle)
                rank, world, lo, hi = loadables.pop("row_shard")
                loadables["human_res"] = hi - lo
            else:
                loadables = self.assemble_shards(files)
                if shard is not None:
                    from .dist import shard_slice
                    lo, hi = shard_slice(int(loadables["human_res"]), int(shard[0]), int(shard[1]))
                    loadables["spatial_occupancy_grids"] = loadables["spatial_occupancy_grids"][lo:hi]
                    loadables["human_res"] = hi - lo
        loadables = to_np_torch_recursive(loadables, use_torch=True, device=self.device)
        for k, v in loadables.items():
            setattr(self, k, v)
        self._centers = None            # derived from spatial_grid
