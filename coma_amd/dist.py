"""Multi-GPU layer of the ComA path: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in CPU tests).  SURVEY.md 8e:

  * contact/orientation: samples are sharded across ranks, every state tensor is a plain sum over
    samples, so ONE all-reduce(SUM) of the five tensors (2*H*O*N + 3*H*O floats) at the end rebuilds the
    single-process ComA.  The two [H,O,N] histograms are reduced in place (no 3.8 GB staging copy);
    the three [H,O] maps travel as one flat bucket.  used_count is summed as an int.
  * occupancy: human vertices (rows) are sharded, each rank reduces its rows locally and the [R,R,R]
    result is combined with all-reduce(MAX) (NaN-propagating like the single-process reducer).

The reference has no collective at all (its workers only share files); this step is new by design.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_slice(n_items: int, rank: int, world: int):
    """Balanced contiguous slice of ``n_items`` work items for ``rank`` (sizes differ by at most 1)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def reference_slice(n_items: int, rank: int, world: int):
    """The reference's per-GPU slice arithmetic (src/generation/inpaint.py:271-274): sub_length =
    len // n + 1, so 512 items over 8 ranks give 65,65,...,57.  Kept for the inpainting work list."""
    sub = n_items // world + 1
    return min(rank * sub, n_items), min((rank + 1) * sub, n_items)


def all_reduce_coma(coma, group=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    big = [coma.prob_grid_canon_human_wrt_obj, coma.prob_grid_canon_obj_wrt_human]
    small = [coma.contact_dist_expectation_grid_nom, coma.contact_dist_expectation_grid_denom,
             coma.significant_contact_count]
    works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t in big]
    flat = torch.cat([t.reshape(-1) for t in small])
    works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
    cnt = torch.tensor([coma.used_count], dtype=torch.int64, device=flat.device)
    works.append(dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    off = 0
    for t in small:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    coma.used_count = int(cnt.item())


def all_reduce_max_nan(t: torch.Tensor, group=None):
    """all-reduce(MAX) that propagates NaN the way torch.max does on one device."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    nan = torch.isnan(t).to(torch.float32)
    clean = torch.nan_to_num(t, nan=float("-inf"))
    dist.all_reduce(clean, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(nan, op=dist.ReduceOp.MAX, group=group)
    t.copy_(torch.where(nan > 0, torch.full_like(clean, float("nan")), clean))
    return t


def gather_rows_to_rank0(local: torch.Tensor, n_rows: int, group=None, dst: int = 0):
    """Row-sharded state (rank r holds rows shard_slice(n_rows, r, world)) -> the full [n_rows, ...] tensor on rank `dst`
    (None on the others).  Used once, for the exported per-vertex occupancy grid; the reduction itself never moves rows.
    Every shard is received straight into its slice of ONE preallocated tensor (point-to-point, no padding, no concatenation):
    rank `dst` holds the full grid plus its own shard and nothing else (the grid is 88 GB at config 5)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    local = local.contiguous()
    if rank != dst:
        if local.shape[0] > 0:
            dist.send(local, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        return None
    full = torch.empty((n_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        lo, hi = shard_slice(n_rows, r, world)
        if hi == lo:
            continue
        if r == dst:
            full[lo:hi].copy_(local)
        else:
            dist.recv(full[lo:hi], src=dist.get_global_rank(group, r) if group is not None else r, group=group)
    return full


def occupancy_rows_reduce(occ_shard, n_rows: int, human_indices=None, group=None, gather: bool = True):
    """SURVEY.md 8e-3 for a row-sharded ComA_Occupancy: normalise and max-reduce the local rows (ONE fused pass when the samples
    are still staged: splat + row sums + max, raw counts left in place), all-reduce(MAX, NaN-propagating) the [R,R,R] field, and
    -- for the export -- gather the raw per-vertex counts to rank 0.  `human_indices` are GLOBAL row indices.
    Returns (full raw grid on rank 0 else None [None everywhere when gather=False], field on every rank)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_slice(n_rows, rank, world)
    local_sel = None if human_indices is None else [i - lo for i in human_indices if lo <= i < hi]
    raw, field = occ_shard.reduce_keep_raw(local_sel, want_raw=gather)
    all_reduce_max_nan(field, group)
    full = gather_rows_to_rank0(raw, n_rows, group) if gather else None
    if full is not None and full is raw:
        full = full.clone()                                    # single process: the grid is normalised in place later
    return full, field


# ------------------------------------------------------------------ K4 reducers, row-parallel
def row_view(coma, lo: int, hi: int):
    """A ComA over human rows [lo, hi) of `coma` that SHARES its state (dim-0 slices are contiguous views): the in-place
    normalisation of the reducers then touches this rank's rows only."""
    import copy
    v = copy.copy(coma)
    v.human_res = hi - lo
    for k in type(coma)._STATE_KEYS:
        setattr(v, k, getattr(coma, k)[lo:hi])
    return v


def _gather_rows(local: torch.Tensor, n_rows: int, group=None):
    """Row shards (shard_slice) -> the full tensor on EVERY rank (padded all_gather: the vectors here are H or H*O floats)."""
    world = dist.get_world_size(group)
    per = -(-n_rows // world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][:shard_slice(n_rows, r, world)[1] - shard_slice(n_rows, r, world)[0]] for r in range(world)], dim=0)


def aggregated_contact_row_parallel(coma, contact_map_type: str, significant_contact_ratio: float, group=None):
    """`get_aggregated_contact` (utils/coma.py:614-641) on an all-reduced ComA with the K4 work sharded by human rows: every rank
    normalises / reduces rows shard_slice(H, rank, world) of the two [H,O,N] grids and the small results are combined --
    "human": the object-column mask is OR-ed over ranks (all-reduce MAX of a u8 vector), the [H] vector is all-gathered;
    "obj": the per-shard column maxima are MAX-all-reduced (NaN-propagating), the row mask is all-gathered.
    Returns (aggregated contact f32 NumPy vector, i64 index vector) on every rank, equal to the single-process call."""
    import numpy as np
    from . import _lib
    assert contact_map_type in ["human", "obj"]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        from .coma import get_aggregated_contact
        return get_aggregated_contact(coma, contact_map_type, significant_contact_ratio)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    H, O = coma.human_res, coma.obj_res
    lo, hi = shard_slice(H, rank, world)
    dev = coma.significant_contact_count.device
    if hi > lo:
        view = row_view(coma, lo, hi)
        cm = view.compute_contact_map(contact_map_type=contact_map_type, as_numpy=False)[contact_map_type].contiguous()
        _, col_any, row_any = view._pairs(significant_contact_ratio)
    else:                                                    # more ranks than rows
        cm = torch.zeros([0, O], dtype=torch.float32, device=dev)
        col_any, row_any = torch.zeros([O], dtype=torch.uint8, device=dev), torch.zeros([0], dtype=torch.uint8, device=dev)
    L = _lib.lib()
    if contact_map_type == "human":
        col = col_any.to(torch.int32)
        dist.all_reduce(col, op=dist.ReduceOp.MAX, group=group)                 # object points with a significant contact anywhere
        col_any = col.to(torch.uint8)
        res = torch.zeros([hi - lo], dtype=torch.float32, device=dev)
        if hi > lo:
            rc = L.coma_masked_max_f32(_lib.ptr(cm, torch.float32), _lib.ptr(col_any), _lib.ptr(row_any), hi - lo, O, 0,
                                       _lib.ptr(res), _lib.stream_ptr(dev))
            _lib.check(rc, "coma_masked_max_f32")
        agg = _gather_rows(res, H, group)
        index = np.argwhere(col_any.cpu().numpy() > 0)[:, 0]                    # reference quirk: object columns for "human"
    else:
        res = torch.full([O], float("-inf"), dtype=torch.float32, device=dev)
        if hi > lo and bool(row_any.any()):
            rc = L.coma_masked_max_f32(_lib.ptr(cm, torch.float32), _lib.ptr(col_any), _lib.ptr(row_any), hi - lo, O, 1,
                                       _lib.ptr(res), _lib.stream_ptr(dev))
            _lib.check(rc, "coma_masked_max_f32")
        all_reduce_max_nan(res, group)
        rows = _gather_rows(row_any, H, group)
        if not bool(rows.any()):
            res = torch.zeros([O], dtype=torch.float32, device=dev)             # nothing significant anywhere (utils/coma.py:424-425)
        agg = res
        index = np.argwhere(rows.cpu().numpy() > 0)[:, 0]                       # ... and human rows for "obj"
    return agg.cpu().numpy(), index


def nonphysical_score_row_parallel(coma, nonphysical_type: str, group=None):
    """`get_nonphysical_score` (entropy response, utils/coma.py:441-487) with the rows sharded the same way -> [H,O] f32 NumPy."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        from .coma import get_nonphysical_score
        return get_nonphysical_score(coma, nonphysical_type)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    H, O = coma.human_res, coma.obj_res
    lo, hi = shard_slice(H, rank, world)
    dev = coma.significant_contact_count.device
    if hi > lo:
        s = row_view(coma, lo, hi).compute_nonphysical_response_sphere(n_bin=1e6, nonphysical_type=nonphysical_type, as_numpy=False)[nonphysical_type]
    else:
        s = torch.zeros([0, O], dtype=torch.float32, device=dev)
    return _gather_rows(s.contiguous(), H, group).cpu().numpy()
