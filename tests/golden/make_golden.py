#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference on CPU.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py

The reference's ComA modules import open3d / trimesh / cv2 / easydict at module level but never touch
them inside the arithmetic we pin, so four empty stub modules are injected.  Nothing
from the reference is copied: the outputs are *data* (inputs + expected outputs).  The same script
also asserts that oracle/coma_oracle.py reproduces every vector, which is what "pins" the oracle.
"""
import os
import sys
import types
import copy
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def _import_reference():
    for m in ["open3d", "trimesh", "cv2"]:
        sys.modules.setdefault(m, types.ModuleType(m))
    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        pass

    ed.EasyDict = EasyDict
    sys.modules.setdefault("easydict", ed)
    # the reference package is called `utils`; make sure OUR utils/ does not shadow it
    sys.path = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
    sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        del sys.modules[k]
    import utils.coma as rc
    import utils.coma_occupancy as ro
    import utils.misc as rm
    assert rc.__file__.startswith(REF)
    return rc, ro, rm


def main():
    rc, ro, rm = _import_reference()
    sys.path.insert(0, ROOT)
    from oracle import coma_oracle as orc
    from tests.synth import make_samples

    out = {}
    chk = []

    def check(name, a, b, exact=True, rtol=0.0):
        a, b = np.asarray(a), np.asarray(b)
        if exact:
            ok = a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
        else:
            ok = a.shape == b.shape and orc.max_rel_err(a, b) <= rtol
        chk.append((name, ok))
        print(f"  oracle vs reference  {name:48s} {'OK' if ok else 'MISMATCH'}"
              + ("" if exact else f"  (max rel {orc.max_rel_err(a, b):.2e})"))

    # ---------------- G1 sphere ----------------
    x, y, z = rc.get_uniform_points_on_sphere(250)
    out["g1_sphere250"] = np.stack([x, y, z], -1)
    check("G1 sphere", orc.fibonacci_sphere(250), out["g1_sphere250"])

    # ---------------- G2 canonicalise / G3 scores ----------------
    rng = np.random.default_rng(2)
    a = rng.normal(size=(7, 3)).astype(np.float32)
    b = rng.normal(size=(5, 3)).astype(np.float32)
    b[0] = [0, 0, -1]          # exactly opposite of p = z  -> mirrored branch
    b[1] = [0, 0, 1]           # identical to p
    b[2] = [0, 0, -3.5]        # opposite after normalisation
    out["g2_a"], out["g2_b"] = a, b
    for tag, p, sp in [("z", [0, 0, 1], [0, 1, 0]), ("x", [1, 0, 0], [0, 1, 0])]:
        for eps in (1e-10, 1e-8):
            r = rc.canonicalize_a_wrt_b_to_p(torch.tensor(a), torch.tensor(b), torch.tensor(p, dtype=torch.float32),
                                             torch.tensor(sp, dtype=torch.float32), eps=eps).numpy()
            key = f"g2_canon_{tag}_{eps:g}"
            out[key] = r
            mine = orc.canonicalize(a, b, np.array(p, np.float32), np.array(sp, np.float32), eps)
            check(key, mine, r, exact=False, rtol=2e-6)
    grid = torch.tensor(out["g1_sphere250"])
    for sigma in (0.2, 0.25):
        c = torch.tensor(out["g2_canon_z_1e-10"])
        r = rc.geodesic_gaussian_scores(grid, c, sigma, 1e-10).numpy()
        out[f"g3_scores_{sigma:g}"] = r
        mine = orc.geodesic_gaussian(out["g1_sphere250"], out["g2_canon_z_1e-10"], sigma, 1e-10)
        assert r.dtype == np.float64 and mine.dtype == np.float64
        check(f"G3 scores sigma={sigma}", mine, r, exact=False, rtol=1e-9)

    # ---------------- G4-G7 ComA state + reducers ----------------
    def run_coma(H, O, N, S, seed, size, thres, sigma, eps, const_obj=True):
        ref = rc.ComA(H, O, N, 0, proximity_settings=dict(spatial_grid_size=size, spatial_grid_thres=thres),
                      principle_vec=[0, 0, 1], sub_principle_vec=[0, 1, 0], rel_dist_method="dist",
                      normal_gaussian_sigma=sigma, eps=eps, device="cpu")
        mine = orc.ComAOracle(H, O, N, size, thres, sigma=sigma, eps=eps)
        samples = make_samples(H, O, S, seed, thres, const_obj)
        for smp in samples:
            ref.register_sample_to_cache(**copy.deepcopy(smp))
            mine.aggregate_sample(**smp)
        ref.aggregate_all_samples()
        return ref, mine, samples

    ref, mine, samples = run_coma(32, 8, 250, 4, seed=4, size=0.07, thres=0.03, sigma=0.25, eps=1e-10)
    for i, s in enumerate(samples):
        for k, v in s.items():
            out[f"g4_in{i}_{k}"] = v
    exp = ref.export()
    out["g4_export_keys"] = np.array(sorted(exp.keys()))
    out["g4_export_dtypes"] = np.array([f"{k}:{getattr(exp[k], 'dtype', type(exp[k]).__name__)}" for k in sorted(exp.keys())])
    st = mine.state()
    for k in ["prob_grid_canon_human_wrt_obj", "prob_grid_canon_obj_wrt_human", "contact_dist_expectation_grid_nom",
              "contact_dist_expectation_grid_denom", "significant_contact_count"]:
        out[f"g4_{k}"] = exp[k]
        check(f"G4 {k}", st[k], exp[k], exact=k.endswith(("denom", "count")), rtol=1e-5)
    out["g4_used_count"] = np.int64(exp["used_count"])
    out["g4_canon_normal_grid_f32"] = exp["canon_normal_grid"]

    # reducers (each on a fresh deep copy: they normalise in place)
    r5 = copy.deepcopy(ref)
    cm = r5.compute_contact_map("both", as_numpy=True)
    out["g5_contact_map_human"], out["g5_contact_map_obj"] = cm["human"], cm["obj"]
    m5 = copy.deepcopy(mine)
    mh, mo = m5.contact_map()
    check("G5 contact_map human", mh, cm["human"], exact=False, rtol=1e-5)
    check("G5 contact_map obj", mo, cm["obj"], exact=False, rtol=1e-5)
    for ratio in (0.1, 0.3, 0.75):
        pairs = copy.deepcopy(ref).significant_contact_pairs(ratio, as_numpy=True)
        out[f"g5_pairs_{ratio:g}"] = pairs
        check(f"G5 pairs ratio={ratio}", mine.significant_pairs(ratio), pairs)
        for which in ("human", "obj"):
            agg, idx = rc.get_aggregated_contact(copy.deepcopy(ref), which, ratio)
            out[f"g5_agg_{which}_{ratio:g}"], out[f"g5_idx_{which}_{ratio:g}"] = agg, idx
            a2, i2, _ = copy.deepcopy(mine).aggregated_contact(which, ratio)
            check(f"G5 aggregated {which} ratio={ratio}", a2, agg, exact=False, rtol=1e-5)
            check(f"G5 index vector {which} ratio={ratio}", i2, idx)
    np6 = copy.deepcopy(ref).compute_nonphysical_response_sphere(n_bin=1e6, nonphysical_type="both", as_numpy=True)
    out["g6_nonphys_human"], out["g6_nonphys_obj"] = np6["human"], np6["obj"]
    nh, no = copy.deepcopy(mine).nonphysical(1e6)
    check("G6 nonphysical human", nh, np6["human"], exact=False, rtol=2e-5)
    check("G6 nonphysical obj", no, np6["obj"], exact=False, rtol=2e-5)

    # G7: export -> load round trip (post-load dtypes; reducers after load use an f32 bin grid)
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "coma.pickle")
        ref.export(pth)
        ref2 = rc.ComA(32, 8, 250, 0, proximity_settings=dict(spatial_grid_size=0.07, spatial_grid_thres=0.03),
                       normal_gaussian_sigma=0.25, eps=1e-10, device="cpu")
        ref2.load(pth)
    out["g7_loaded_dtypes"] = np.array([f"{k}:{v.dtype}" for k, v in sorted(vars(ref2).items()) if isinstance(v, torch.Tensor)])
    agg, idx = rc.get_aggregated_contact(ref2, "human", 0.1)
    out["g7_agg_human_after_load"], out["g7_idx_human_after_load"] = agg, idx
    a2, i2, _ = copy.deepcopy(mine).aggregated_contact("human", 0.1, grid_f32=True)
    check("G7 aggregated human after load", a2, agg, exact=False, rtol=1e-5)
    check("G7 index after load", i2, idx)

    # varying objects across samples + other preset values + non-250 N
    refv, minev, samplesv = run_coma(12, 5, 70, 3, seed=14, size=0.15, thres=0.05, sigma=0.2, eps=1e-10, const_obj=False)
    for i, s in enumerate(samplesv):
        for k, v in s.items():
            out[f"g4v_in{i}_{k}"] = v
    ev = refv.export()
    for k in ["prob_grid_canon_human_wrt_obj", "prob_grid_canon_obj_wrt_human", "contact_dist_expectation_grid_nom",
              "significant_contact_count"]:
        out[f"g4v_{k}"] = ev[k]
        check(f"G4v {k}", minev.state()[k], ev[k], exact=k.endswith("count"), rtol=1e-5)

    # mid-size case: H=256,O=180,N=250,S=3 -> row sums + sampled rows (full grids would be 92 MB)
    refm, minem, samplesm = run_coma(256, 180, 250, 3, seed=24, size=0.07, thres=0.03, sigma=0.25, eps=1e-10)
    out["g4m_seed"] = np.int64(24)
    em = refm.export()
    out["g4m_nom"] = em["contact_dist_expectation_grid_nom"]
    out["g4m_count_u8"] = em["significant_contact_count"].astype(np.uint8)
    check("G4m count", minem.cnt, em["significant_contact_count"])
    check("G4m nom", minem.nom, em["contact_dist_expectation_grid_nom"], exact=False, rtol=1e-5)
    rows = np.random.default_rng(99).integers(0, 256 * 180, size=48)
    out["g4m_rows"] = rows
    for tag, k in (("h", "prob_grid_canon_human_wrt_obj"), ("o", "prob_grid_canon_obj_wrt_human")):
        out[f"g4m_rowsum_{tag}"] = em[k].astype(np.float64).sum(-1).astype(np.float32)
        out[f"g4m_rows_{tag}"] = em[k].reshape(-1, 250)[rows]
        check(f"G4m rows {tag}", minem.state()[k].reshape(-1, 250)[rows], out[f"g4m_rows_{tag}"], exact=False, rtol=1e-5)

    # ---------------- G8-G10 occupancy ----------------
    for R in (4, 30):
        g, ig, md = ro.load_voxelgrid(2.4, R, [0, 0, 0])
        assert g.dtype == np.float64
        out[f"g8_axis_{R}"] = np.stack([g[0, :, 0, 0], g[1, 0, :, 0], g[2, 0, 0, :]])
        out[f"g8_voxel_{R}"] = np.float64(md["voxel_size"])
        c, _, vox, _ = orc.voxel_centers(2.4, R)
        check(f"G8 voxel grid R={R}", c, g)
    H, R, S = 16, 8, 4
    rng = np.random.default_rng(9)
    refo = ro.ComA_Occupancy(scale_tolerance=3.0, human_res=H, obj_res=3, normal_res=0, spatial_res=R, device="cpu")
    mineo = orc.OccupancyOracle(H, R, 3.0)
    ov = rng.normal(scale=0.1, size=(3, 3))
    on = np.tile(np.array([[0, 0, 1.0]]), (3, 1))
    thres = refo.rel_dist_thres
    for s in range(S):
        hv = rng.uniform(-1.3, 1.3, size=(H, 3)) + ov[0]
        if s == 0:   # a vertex whose distance to a voxel centre is within 3 f32 ulps of the threshold sphere
            ctr = np.array([refo.spatial_grid[c, 3, 4, 2].item() for c in range(3)])
            dirn = np.array([1.0, 0, 0])
            hv[0] = ctr + dirn * thres * (1 - 2e-7) + ov[0]
            hv[1] = ctr + dirn * thres * (1 + 2e-7) + ov[0]
        hv[2] = np.array([0.31, -0.2, 0.5]) + ov[0]       # make sure rows 0..2 are never empty
        out[f"g9_in{s}_human_verts"] = hv
        smp = dict(human_verts=hv, human_normals=np.zeros_like(hv), obj_verts=ov, obj_normals=on)
        refo.register_sample_to_cache(**copy.deepcopy(smp))
        mineo.aggregate_sample(hv, ov)
    out["g9_obj_verts"], out["g9_obj_normals"] = ov, on
    refo.aggregate_all_samples()
    out["g9_counts"] = refo.spatial_occupancy_grids.numpy().copy()
    check("G9 occupancy counts", mineo.occ, out["g9_counts"])
    eo = refo.export()
    out["g9_export_keys"] = np.array(sorted(eo.keys()))
    out["g10_grid_with_nan"] = copy.deepcopy(refo).return_aggregated_spatial_grids().numpy()
    check("G10 aggregated grid (NaN rows)", copy.deepcopy(mineo).aggregated_grid(), out["g10_grid_with_nan"])
    # same, restricted to never-empty rows
    sel = [h for h in range(H) if out["g9_counts"][h].sum() > 0]
    out["g10_sel"] = np.array(sel)
    out["g10_grid_sel"] = copy.deepcopy(refo).return_aggregated_spatial_grids(human_indices=sel).numpy()

    # G10b: a vertex that never falls inside the grid -> its row is 0/0 = NaN -> torch.max poisons the grid
    refn = ro.ComA_Occupancy(scale_tolerance=3.0, human_res=4, obj_res=1, normal_res=0, spatial_res=8, device="cpu")
    minen = orc.OccupancyOracle(4, 8, 3.0)
    rngn = np.random.default_rng(10)
    for s in range(2):
        hvn = rngn.uniform(-1.0, 1.0, size=(4, 3))
        hvn[3] = [5.0, 5.0, 5.0]
        out[f"g10b_in{s}_human_verts"] = hvn
        refn.register_sample_to_cache(human_verts=hvn, human_normals=np.zeros((4, 3)), obj_verts=np.zeros((1, 3)),
                                      obj_normals=np.ones((1, 3)))
        minen.aggregate_sample(hvn, np.zeros((1, 3)))
    refn.aggregate_all_samples()
    out["g10b_counts"] = refn.spatial_occupancy_grids.numpy().copy()
    out["g10b_grid"] = copy.deepcopy(refn).return_aggregated_spatial_grids().numpy()
    out["g10b_grid_sel012"] = copy.deepcopy(refn).return_aggregated_spatial_grids(human_indices=[0, 1, 2]).numpy()
    check("G10b counts", minen.occ, out["g10b_counts"])
    check("G10b NaN grid", copy.deepcopy(minen).aggregated_grid(), out["g10b_grid"])

    # ---------------- G7b: pickles WRITTEN BY THE REFERENCE (file-format fixtures) ----------------
    refs, _, smps = run_coma(6, 4, 16, 2, seed=77, size=0.07, thres=0.03, sigma=0.25, eps=1e-10)
    refs.export(os.path.join(HERE, "ref_coma_small.pickle"))
    aggs, idxs = rc.get_aggregated_contact(copy.deepcopy(refs), "human", 0.1)
    out["g7b_agg_human"], out["g7b_idx_human"] = aggs, idxs
    refos = ro.ComA_Occupancy(scale_tolerance=3.0, human_res=5, obj_res=2, normal_res=0, spatial_res=6, device="cpu")
    rngs = np.random.default_rng(78)
    for _ in range(3):
        refos.register_sample_to_cache(human_verts=rngs.uniform(-0.4, 0.4, size=(5, 3)), human_normals=np.zeros((5, 3)),
                                       obj_verts=np.zeros((2, 3)), obj_normals=np.ones((2, 3)))
    refos.aggregate_all_samples()
    refos.export(os.path.join(HERE, "ref_occupancy_small.pickle"))
    out["g7b_occ_grid"] = copy.deepcopy(refos).return_aggregated_spatial_grids().numpy()

    # ---------------- presets: the reference's hyper-parameter tables as data ----------------
    import importlib.util, json
    def _load(path, name):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    rq = _load(os.path.join(REF, "constants/coma/qual.py"), "_ref_qual").QUAL_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT
    rn = _load(os.path.join(REF, "constants/coma/quant.py"), "_ref_quant").QUANT_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT
    with open(os.path.join(HERE, "presets.json"), "w") as fh:
        json.dump({"qual": rq, "quant": rn}, fh, indent=1, sort_keys=True)

    # ---------------- G11 nearest vertex ----------------
    rng = np.random.default_rng(11)
    verts = rng.normal(size=(500, 3))
    verts[123] = verts[45]                      # duplicated vertex -> tie, first index must win
    pts = verts[rng.integers(0, 500, size=64)] + rng.normal(scale=1e-3, size=(64, 3))
    pts[5] = verts[123]
    pts[6] = 0.5 * (verts[10] + verts[20])      # equidistant in exact arithmetic

    class _Pcd:
        points = pts

    class _Mesh:
        vertices = verts

        def sample_points_poisson_disk(self, number_of_points):
            return _Pcd()

    idx, _ = rc.simplify_mesh_and_get_indices(_Mesh(), 64)
    out["g11_points"], out["g11_verts"], out["g11_idx"] = pts, verts, np.asarray(idx, np.int64)
    check("G11 nearest vertex", orc.nearest_vertex(pts, verts), out["g11_idx"])

    # ---------------- G12 dtype table of to_np_torch_recursive ----------------
    table = []
    for dt in [np.float64, np.float32, np.float16, np.int64, np.int32, np.int16, np.uint8, np.bool_]:
        t = rm.to_np_torch_recursive(np.zeros(2, dt), use_torch=True, device="cpu")
        n = rm.to_np_torch_recursive(torch.zeros(2, dtype=t.dtype), use_torch=False, device="cpu")
        table.append(f"{np.dtype(dt).name}->{t.dtype}->{n.dtype}")
    out["g12_dtype_table"] = np.array(table)

    # ---------------- G17 the optimisation app's reading of the state (src/application/optimize.py:190-196) ----------------
    # those lines sit inside optimize_smpl (which needs smplx / COAP / VPoser); they are executed here verbatim, read from
    # the reference file at generation time, on states exported by the reference's own ComA
    import textwrap
    with open(os.path.join(REF, "src", "application", "optimize.py")) as fh:
        lines = fh.read().split("\n")[189:196]          # file lines 190-196
    assert lines[0].lstrip().startswith("grid_prob = affordance_info") and "corresponding_object_indices" in lines[-1], lines
    consumer_src = textwrap.dedent("\n".join(lines))

    def run_consumer(info, o_ref, thr):
        ns = dict(np=np, affordance_info=info, reference_object_vertex_index=o_ref, contact_threshold=thr)
        with np.errstate(divide="ignore", invalid="ignore"):
            exec(consumer_src, ns)
        return (ns["max_prob_indices"], ns["relative_orientation_GT"], ns["selected_human_indices"][0],
                ns["corresponding_object_indices"])

    tricky = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in exp.items()}
    P = tricky["prob_grid_canon_human_wrt_obj"]
    P[3, 5, :] = 0.0                                  # all-equal row -> first index
    P[4, 5, 17] = P[4, 5, 200] = P[4, 5].max() * 2     # tie of two maxima -> the first
    P[6, 5, 9] = np.nan                                # NaN counts as the maximum
    tricky["contact_dist_expectation_grid_denom"][7, :] = 0.0     # 0/0 and x/0 in the ratio
    tricky["contact_dist_expectation_grid_nom"][7, :3] = 0.0
    tricky["contact_dist_expectation_grid_nom"][8, 2] = tricky["contact_dist_expectation_grid_nom"][8, 6] = 99.0   # tie
    for k in ["prob_grid_canon_human_wrt_obj", "contact_dist_expectation_grid_nom", "contact_dist_expectation_grid_denom"]:
        out[f"g17t_{k}"] = tricky[k]
    ratio = exp["contact_dist_expectation_grid_nom"] / exp["contact_dist_expectation_grid_denom"]
    thr_mid = float(np.median(ratio.max(1)))
    out["g17_thresholds"] = np.array([0.0, thr_mid, 10.0])
    for tag, info in (("", exp), ("t", tricky)):
        for o_ref in (0, 5):
            for ti, thr in enumerate(out["g17_thresholds"]):
                got = run_consumer(info, o_ref, float(thr))
                mine17 = orc.orientation_and_contact_targets(info, o_ref, float(thr))
                for name, a, b in zip(("argmax", "orientation", "selected", "objects"), got,
                                      (mine17[0], mine17[1], mine17[2][0], mine17[3])):
                    out[f"g17{tag}_{name}_o{o_ref}_t{ti}"] = np.asarray(a)
                    check(f"G17{tag} {name} o={o_ref} thr#{ti}", np.asarray(b), np.asarray(a))

    bad = [n for n, ok in chk if not ok]
    if bad:
        raise SystemExit(f"oracle does not reproduce the reference on: {bad}")
    np.savez_compressed(os.path.join(HERE, "coma_golden.npz"), **out)
    sz = os.path.getsize(os.path.join(HERE, "coma_golden.npz"))
    print(f"wrote coma_golden.npz  ({sz/1e6:.2f} MB, {len(out)} arrays); oracle pinned on {len(chk)} checks")


if __name__ == "__main__":
    main()
