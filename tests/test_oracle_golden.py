"""CPU: the oracle (oracle/coma_oracle.py) against the golden vectors captured from the real reference."""
import copy

import numpy as np
import pytest

from oracle import coma_oracle as orc


def _samples(g, prefix, S):
    return [dict(human_verts=g[f"{prefix}_in{i}_human_verts"], human_normals=g[f"{prefix}_in{i}_human_normals"],
                 obj_verts=g[f"{prefix}_in{i}_obj_verts"], obj_normals=g[f"{prefix}_in{i}_obj_normals"]) for i in range(S)]


def _run(g, prefix, S, H, O, N, size, thres, sigma):
    m = orc.ComAOracle(H, O, N, size, thres, sigma=sigma, eps=1e-10)
    for s in _samples(g, prefix, S):
        m.aggregate_sample(**s)
    return m


def test_sphere(golden):
    assert np.array_equal(orc.fibonacci_sphere(250), golden["g1_sphere250"])


def test_canonicalize_including_antiparallel_and_nonz_p(golden):
    a, b = golden["g2_a"], golden["g2_b"]
    for tag, p in (("z", [0, 0, 1]), ("x", [1, 0, 0])):
        for eps in (1e-10, 1e-8):
            r = orc.canonicalize(a, b, np.array(p, np.float32), np.array([0, 1, 0], np.float32), eps)
            assert orc.max_rel_err(r, golden[f"g2_canon_{tag}_{eps:g}"]) <= 2e-6


def test_geodesic_gaussian_is_f64(golden):
    for sigma in (0.2, 0.25):
        r = orc.geodesic_gaussian(golden["g1_sphere250"], golden["g2_canon_z_1e-10"], sigma, 1e-10)
        assert r.dtype == np.float64
        assert orc.max_rel_err(r, golden[f"g3_scores_{sigma:g}"], floor=0) <= 1e-9


def test_state_after_four_samples(golden):
    m = _run(golden, "g4", 4, 32, 8, 250, 0.07, 0.03, 0.25)
    st = m.state()
    assert np.array_equal(st["significant_contact_count"], golden["g4_significant_contact_count"])
    assert st["significant_contact_count"].sum() > 0
    assert np.array_equal(st["contact_dist_expectation_grid_denom"], golden["g4_contact_dist_expectation_grid_denom"])
    for k in ("prob_grid_canon_human_wrt_obj", "prob_grid_canon_obj_wrt_human", "contact_dist_expectation_grid_nom"):
        assert orc.max_rel_err(st[k], golden[f"g4_{k}"]) <= 1e-5
    assert m.used_count == int(golden["g4_used_count"])


def test_reducers_and_index_vectors(golden):
    m = _run(golden, "g4", 4, 32, 8, 250, 0.07, 0.03, 0.25)
    h, o = copy.deepcopy(m).contact_map()
    assert orc.max_rel_err(h, golden["g5_contact_map_human"]) <= 1e-5
    assert orc.max_rel_err(o, golden["g5_contact_map_obj"]) <= 1e-5
    for ratio in (0.1, 0.3, 0.75):
        assert np.array_equal(m.significant_pairs(ratio), golden[f"g5_pairs_{ratio:g}"])
        for which in ("human", "obj"):
            agg, idx, _ = copy.deepcopy(m).aggregated_contact(which, ratio)
            assert orc.max_rel_err(agg, golden[f"g5_agg_{which}_{ratio:g}"]) <= 1e-5
            assert idx.dtype == np.int64 and np.array_equal(idx, golden[f"g5_idx_{which}_{ratio:g}"])
    nh, no = copy.deepcopy(m).nonphysical(1e6)
    assert orc.max_rel_err(nh, golden["g6_nonphys_human"]) <= 2e-5
    assert orc.max_rel_err(no, golden["g6_nonphys_obj"]) <= 2e-5
    agg, idx, _ = copy.deepcopy(m).aggregated_contact("human", 0.1, grid_f32=True)
    assert orc.max_rel_err(agg, golden["g7_agg_human_after_load"]) <= 1e-5


def test_varying_objects(golden):
    m = _run(golden, "g4v", 3, 12, 5, 70, 0.15, 0.05, 0.2)
    assert np.array_equal(m.cnt, golden["g4v_significant_contact_count"])
    assert orc.max_rel_err(m.P_h_wrt_o, golden["g4v_prob_grid_canon_human_wrt_obj"]) <= 1e-5
    assert orc.max_rel_err(m.P_o_wrt_h, golden["g4v_prob_grid_canon_obj_wrt_human"]) <= 1e-5


def test_voxel_grid_and_occupancy(golden):
    for R in (4, 30):
        c, _, vox, _ = orc.voxel_centers(2.4, R)
        assert c.dtype == np.float64 and vox == float(golden[f"g8_voxel_{R}"])
        assert np.array_equal(np.stack([c[0, :, 0, 0], c[1, 0, :, 0], c[2, 0, 0, :]]), golden[f"g8_axis_{R}"])
    m = orc.OccupancyOracle(16, 8, 3.0)
    for s in range(4):
        m.aggregate_sample(golden[f"g9_in{s}_human_verts"], golden["g9_obj_verts"])
    assert np.array_equal(m.occ, golden["g9_counts"])
    # sample 0 holds two vertices placed 3 f32 ulps inside / outside the threshold sphere of voxel (3,4,2)
    m0 = orc.OccupancyOracle(16, 8, 3.0)
    m0.aggregate_sample(golden["g9_in0_human_verts"], golden["g9_obj_verts"])
    assert m0.occ[0, 3, 4, 2] == 1 and m0.occ[1, 3, 4, 2] == 0
    out = copy.deepcopy(m).aggregated_grid()
    assert np.array_equal(out, golden["g10_grid_with_nan"], equal_nan=True)
    # a vertex that is never inside the grid makes its row 0/0 and poisons the whole grid (reference quirk)
    mn = orc.OccupancyOracle(4, 8, 3.0)
    for s in range(2):
        mn.aggregate_sample(golden[f"g10b_in{s}_human_verts"], np.zeros((1, 3)))
    assert np.array_equal(mn.occ, golden["g10b_counts"]) and golden["g10b_counts"][3].sum() == 0
    assert np.isnan(golden["g10b_grid"]).all() and np.isnan(mn.aggregated_grid()).all()
    assert not np.isnan(golden["g10b_grid_sel012"]).any()


def test_windowed_occupancy_oracle_equals_dense(golden):
    """OccupancyOracle.aggregate_windowed (used to check the config-5-size GPU run) == the dense evaluation, bit for bit,
    including points outside / on the edge of the grid and the golden near-threshold vertices."""
    for R, H, S, span in ((8, 16, 4, 1.4), (30, 12, 6, 1.3), (37, 5, 3, 1.25)):
        rng = np.random.default_rng(R)
        q = rng.uniform(-span, span, size=(S, H, 3))
        if R == 8:
            q[:4] = np.stack([golden[f"g9_in{s}_human_verts"] for s in range(4)])
        dense, win = orc.OccupancyOracle(H, R, 3.0), orc.OccupancyOracle(H, R, 3.0)
        for s in range(S):
            dense.aggregate_sample(q[s], np.zeros((1, 3)))
        win.aggregate_windowed(q.astype(np.float32))
        assert dense.occ.sum() > 0 and np.array_equal(dense.occ, win.occ)


def test_nearest_vertex_first_minimum(golden):
    idx = orc.nearest_vertex(golden["g11_points"], golden["g11_verts"])
    assert np.array_equal(idx, golden["g11_idx"])
    assert golden["g11_idx"][5] == 45        # duplicated vertex 45/123: first index wins


def _g17_info(golden, tricky):
    pre = "g17t_" if tricky else "g4_"
    return {"prob_grid_canon_human_wrt_obj": golden[pre + "prob_grid_canon_human_wrt_obj"],
            "contact_dist_expectation_grid_nom": golden[pre + "contact_dist_expectation_grid_nom"],
            "contact_dist_expectation_grid_denom": golden[pre + "contact_dist_expectation_grid_denom"],
            "canon_normal_grid": golden["g4_canon_normal_grid_f32"]}


@pytest.mark.parametrize("tricky", [False, True])
def test_optimisation_app_targets(golden, tricky):
    """G17: src/application/optimize.py:190-196 executed by the generator on reference-exported states (plain, and with
    ties / an all-equal row / NaN / zero denominators injected)."""
    info, tag = _g17_info(golden, tricky), "g17t" if tricky else "g17"
    for o_ref in (0, 5):
        for ti, thr in enumerate(golden["g17_thresholds"]):
            am, ori, sel, obj = orc.orientation_and_contact_targets(info, o_ref, float(thr))
            assert np.array_equal(am, golden[f"{tag}_argmax_o{o_ref}_t{ti}"])
            assert np.array_equal(ori, golden[f"{tag}_orientation_o{o_ref}_t{ti}"], equal_nan=True)
            assert np.array_equal(sel[0], golden[f"{tag}_selected_o{o_ref}_t{ti}"])
            assert np.array_equal(obj, golden[f"{tag}_objects_o{o_ref}_t{ti}"])


def test_torch_cpu_port_agrees_with_the_pinned_numpy_oracle():
    """oracle/coma_oracle_torch.py (bench.py's all-threads CPU baseline) against the NumPy oracle, which G1-G7 pin to the reference."""
    from oracle import coma_oracle as orc
    from oracle import coma_oracle_torch as ot
    from tests.synth import make_samples
    H, O, N = 40, 12, 250
    a = orc.ComAOracle(H, O, N, 0.07, 0.03, sigma=0.25, eps=1e-10)
    b = ot.ComATorch(H, O, N, 0.07, 0.03, sigma=0.25, eps=1e-10)
    for smp in make_samples(H, O, 3, seed=3, thres=0.03):
        a.aggregate_sample(**smp)
        b.aggregate_sample(**smp)
    assert np.array_equal(a.cnt, b.cnt.numpy()) and np.array_equal(a.den, b.den.numpy())
    assert orc.max_rel_err(b.nom.numpy(), a.nom) <= 1e-5
    assert orc.max_rel_err(b.P_h_wrt_o.numpy(), a.P_h_wrt_o) <= 1e-4 and orc.max_rel_err(b.P_o_wrt_h.numpy(), a.P_o_wrt_h) <= 1e-4
    assert np.allclose(ot.fibonacci_sphere(250).numpy(), orc.fibonacci_sphere(250), rtol=0, atol=1e-15)
