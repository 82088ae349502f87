"""GPU: the inference CLI surface (src/coma/inference.py) end to end on synthetic pickles in the reference's formats:
down-sample pickles (src/coma/downsample_human.py:67-77, downsample_objects.py:46-60), a ComA pickle, and the four
output artefacts with their normalisations (x/x.max(), min-max, 0.7*field/max) checked against the oracle."""
import copy
import os
import pickle

import numpy as np
import pytest

from oracle import coma_oracle as orc
from tests.synth import make_samples

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pickles(tmp_path, H, O):
    rng = np.random.default_rng(0)
    hp, op = tmp_path / "smplx.pickle", tmp_path / "asset.pickle"
    pickle.dump(dict(N=H, N_raw=H, downsample_indices=list(range(H))), open(hp, "wb"))
    pickle.dump(dict(N=O, N_raw=O, downsample_indices=list(range(O)), downsampled_pcd_points_raw=rng.normal(size=(O, 3)),
                     downsampled_pcd_normal_raw=rng.normal(size=(O, 3))), open(op, "wb"))
    return str(hp), str(op)


def test_contact_and_orientation_outputs(tmp_path, hip_lib):
    from constants.coma.qual import QUAL_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT as Q
    from src.coma.inference import inference
    from utils.coma import ComA
    H, O, N, S = 40, 12, 250, 6
    hp_path, op_path = _pickles(tmp_path, H, O)
    for key, fname in (("qual:backpack_human_contact", "human_contact.npy"), ("qual:backpack_object_contact", "object_contact.ply"),
                       ("qual:backpack_orientation", "orientational_tendency.npy")):
        hp = Q[key]
        samples = make_samples(H, O, S, seed=3, thres=hp["spatial_grid_thres"])
        coma = ComA(H, O, N, 0, proximity_settings=dict(spatial_grid_size=hp["spatial_grid_size"], spatial_grid_thres=hp["spatial_grid_thres"]),
                    normal_gaussian_sigma=hp["normal_gaussian_sigma"], eps=hp["eps"], device=DEV)
        ref = orc.ComAOracle(H, O, N, hp["spatial_grid_size"], hp["spatial_grid_thres"], sigma=hp["normal_gaussian_sigma"], eps=hp["eps"])
        for s in samples:
            coma.register_sample_to_cache(**copy.deepcopy(s))
            ref.aggregate_sample(**s)
        coma.aggregate_all_samples()
        cp = str(tmp_path / f"{key.replace(':', '_')}.pickle")
        coma.export(cp)
        out = inference("BEHAVE", "backpack", cp, hp_path, op_path, None, key, hp, str(tmp_path / "out"), device=DEV)
        path = os.path.join(out, fname)
        assert os.path.exists(path)
        if fname == "human_contact.npy":
            agg, _, _ = ref.aggregated_contact("human", hp["significant_contact_ratio"], grid_f32=True)
            got = np.load(path)
            assert got.shape == (H,) and orc.max_rel_err(got, agg / agg.max()) <= 1e-3 and got.max() == 1.0
        elif fname == "orientational_tendency.npy":
            s = ref.nonphysical(1e6)[0][:, 0]
            got = np.load(path)
            # min-max over a narrow score range amplifies errors by 1/(max-min) ~ 30x -> absolute tolerance on [0,1]
            assert float(np.abs(got - (s - s.min()) / (s.max() - s.min())).max()) <= 2e-3 and got.min() == 0.0 and got.max() == 1.0
        else:
            lines = open(path).read().splitlines()
            assert lines[0] == "ply" and f"element vertex {O}" in lines and len(lines) == 13 + O


def test_occupancy_output(tmp_path, hip_lib):
    from constants.coma.qual import QUAL_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT as Q
    from src.coma.inference import inference
    from utils.coma_occupancy import ComA_Occupancy
    hp = dict(Q["qual:backpack_occupancy"], spatial_res=10)
    H, O = 12, 3
    hp_path, op_path = _pickles(tmp_path, H, O)
    occ = ComA_Occupancy(scale_tolerance=3.0, human_res=H, obj_res=O, normal_res=0, spatial_res=10, device=DEV)
    ref = orc.OccupancyOracle(H, 10, 3.0)
    rng = np.random.default_rng(1)
    for _ in range(4):
        hv = rng.uniform(-1.0, 1.0, size=(H, 3))
        occ.register_sample_to_cache(human_verts=hv, human_normals=hv, obj_verts=np.zeros((O, 3)), obj_normals=np.ones((O, 3)))
        ref.aggregate_sample(hv, np.zeros((O, 3)))
    occ.aggregate_all_samples()
    cp = str(tmp_path / "occ.pickle")
    occ.export(cp)
    out = inference("BEHAVE", "backpack", cp, hp_path, op_path, None, "qual:backpack_occupancy", hp, str(tmp_path / "out"), device=DEV)
    d = np.load(os.path.join(out, "occupancy.npy"), allow_pickle=True).item()
    field = ref.aggregated_grid()
    assert np.array_equal(d["prob_field"], 0.7 * (field / field.max()))
    assert d["spatial_grid_metadata"]["N_x"] == 10 and abs(d["spatial_grid_metadata"]["voxel_size"] - 0.24) < 1e-12


def _mesh(rng, V=80):
    """Random closed-ish triangle soup over V vertices on a bumpy sphere (every vertex used)."""
    pts = rng.normal(size=(V, 3))
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    from scipy.spatial import ConvexHull
    hull = ConvexHull(pts)
    return pts * 0.4, hull.simplices.astype(np.int64)


def test_vertex_normals_kernel_vs_oracle(hip_lib):
    from coma_amd.ingest import vertex_normals_batch
    rng = np.random.default_rng(0)
    base, faces = _mesh(rng)
    verts = np.stack([base + rng.normal(scale=0.01, size=base.shape) for _ in range(3)])
    got = vertex_normals_batch(verts, faces, device=DEV)
    for s in range(3):
        ref = orc.vertex_normals(verts[s], faces)
        assert np.abs(got[s] - ref).max() <= 1e-14
    assert np.allclose(np.linalg.norm(got, axis=-1), 1.0)


def test_extract_coma_cli_end_to_end(tmp_path, hip_lib):
    """results/ tree in the reference's formats -> extract_coma -> pickle + human_contact.npy, checked against the oracle
    fed with oracle vertex normals; a sentinel sample and a post-filtered sample must be ignored."""
    import json
    from constants.coma.qual import QUAL_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT as Q
    from src.coma.extract_coma import run_affordance_extraction
    rng = np.random.default_rng(5)
    key = "qual:backpack_human_contact"
    hp = dict(Q[key], human_res="20", enable_postfilter=True)
    base, faces = _mesh(rng)
    V, H, O = len(base), 20, 6
    hidx = list(rng.choice(V, size=H, replace=False))
    opts = base[hidx[:O]] / 0.4                                    # object points hugging six of the sampled body vertices
    obj_meta = dict(N=O, N_raw=O, downsample_indices=list(range(O)), downsampled_pcd_points_raw=opts * 0.41,
                    downsampled_pcd_normal_raw=-opts.copy(), obj_vertices_original=opts * 0.41, obj_faces_original=np.zeros((1, 3), int),
                    obj_vertex_normals_original=-opts.copy())
    (tmp_path / "mesh").mkdir()
    pickle.dump(dict(N=H, N_raw=H, downsample_indices=hidx), open(tmp_path / "mesh" / "smplx_star_downsampled_20.pickle", "wb"))
    ad = tmp_path / "asset_ds" / "BEHAVE" / "backpack"
    ad.mkdir(parents=True)
    pickle.dump(obj_meta, open(ad / "behave_asset_180.pickle", "wb"))
    prompt = "1 person wears the backpack"
    sd = tmp_path / "hs" / "BEHAVE" / "backpack" / "behave_asset" / "view:00000" / "00001" / prompt
    sd.mkdir(parents=True)
    ref = orc.ComAOracle(H, O, hp["normal_res"], hp["spatial_grid_size"], hp["spatial_grid_thres"], sigma=hp["normal_gaussian_sigma"], eps=hp["eps"])
    listed = []
    for i in range(5):
        verts = base + rng.normal(scale=0.004, size=base.shape)
        payload = "TOO LITTLE INLIERS" if i == 3 else dict(verts=verts, faces=faces, IoU=0.9, num_inliers=5)
        pickle.dump(payload, open(sd / f"{i:06}.pickle", "wb"))
        if i in (0, 1, 2):                      # sample 4 is valid but not listed by the post-filter, sample 3 is a sentinel
            listed.append(["view:00000", "00001", prompt, f"{i:06}"])
            n = orc.vertex_normals(verts, faces)
            n = n / (np.sqrt((n**2).sum(-1, keepdims=True)) + hp["eps"])
            ref.aggregate_sample(verts[hidx], n[hidx], obj_meta["downsampled_pcd_points_raw"], obj_meta["downsampled_pcd_normal_raw"])
    pf = tmp_path / "pf" / "BEHAVE" / "backpack" / "behave_asset"
    pf.mkdir(parents=True)
    json.dump(listed, open(pf / f"{prompt}.json", "w"))
    done = run_affordance_extraction(None, None, None, str(tmp_path / "cam"), str(tmp_path / "params"), str(tmp_path / "asset_ds"),
                                     str(tmp_path / "pf"), str(tmp_path / "hs"), str(tmp_path / "coma"), str(tmp_path / "aff"),
                                     str(tmp_path / "mesh"), hp, key, 3.0, False, device=DEV)
    assert len(done) == 1
    _, save_pth, out = done[0]
    state = pickle.load(open(save_pth, "rb"))
    assert state["used_count"] == 3
    assert np.array_equal(state["significant_contact_count"], ref.cnt)
    assert orc.max_rel_err(state["prob_grid_canon_human_wrt_obj"], ref.P_h_wrt_o) <= 1e-3       # exported BEFORE normalisation
    agg, _, _ = ref.aggregated_contact("human", hp["significant_contact_ratio"])
    got = np.load(os.path.join(out, "human_contact.npy"))
    exp = agg / agg.max() if agg.max() > 0 else agg
    assert agg.max() > 0 and ref.cnt.sum() > 0
    assert np.allclose(got, exp, atol=2e-3)
    meta = json.load(open(save_pth.replace(".pickle", ".json")))
    assert meta["H"] == H and meta["O"] == O and len(meta["input_human_pths"]) == 3


def test_downsample_writers_schema_and_index_map(tmp_path, hip_lib):
    """src/coma/downsample_{human,objects}.py: the reference's pickle schema (downsample_human.py:67-77,
    downsample_objects.py:46-60) from a mesh + sampled points, index map = first-minimum nearest vertex (oracle)."""
    import pickle, types
    from oracle import coma_oracle as orc
    from src.coma import downsample_human as dh, downsample_objects as do
    rng = np.random.default_rng(0)
    # a closed triangulated box surface with a few hundred vertices
    g = np.linspace(-0.5, 0.5, 9)
    verts, faces = [], []
    def quad_grid(fix_axis, val):
        base = len(verts)
        for a in g:
            for b in g:
                p = [0.0, 0.0, 0.0]
                ax = [i for i in range(3) if i != fix_axis]
                p[fix_axis], p[ax[0]], p[ax[1]] = val, a, b
                verts.append(p)
        n = len(g)
        for i in range(n - 1):
            for j in range(n - 1):
                q = base + i * n + j
                tri = [[q, q + 1, q + n + 1], [q, q + n + 1, q + n]]
                faces.extend(tri if val > 0 else [t[::-1] for t in tri])
    for axis in range(3):
        quad_grid(axis, 0.5)
        quad_grid(axis, -0.5)
    verts, faces = np.array(verts), np.array(faces)
    with open(tmp_path / "star.pickle", "wb") as h:
        pickle.dump({"vertices": verts.astype(np.float32), "faces": faces}, h)
    args = types.SimpleNamespace(mesh_pth=str(tmp_path / "star.pickle"), points_pth=None, simplify_method="uniform", seed=5, skip_done=False,
                                 save_dir=str(tmp_path / "mesh"), num_human_downsample_points=100)
    pth = dh.downsample_smplx(args, device="cuda:0")
    assert pth.endswith("smplx_star_downsampled_100.pickle")
    d = pickle.load(open(pth, "rb"))
    assert set(d) == {"vertices", "faces", "V", "F", "N", "N_raw", "downsample_indices", "downsampled_pcd_points_raw", "downsampled_pcd_normal_raw"}
    assert d["V"] == len(verts) and d["F"] == len(faces) and d["N_raw"] == 100 and d["N"] == len(d["downsample_indices"]) <= 100
    ref_idx = orc.nearest_vertex(d["downsampled_pcd_points_raw"], verts.astype(np.float32).astype(np.float64))
    assert d["downsample_indices"] == [int(i) for i in ref_idx]
    assert np.allclose(np.linalg.norm(d["downsampled_pcd_normal_raw"], axis=1), 1.0)
    args.num_human_downsample_points = 10 ** 6                       # >= V: every vertex, FULL file name
    assert dh.downsample_smplx(args, device="cuda:0").endswith("smplx_star_downsampled_FULL.pickle")
    # object writer from an OBJ file + supplied points (two of them with a zero normal -> dropped from the raw set only)
    with open(tmp_path / "box.obj", "w") as h:
        h.writelines(f"v {x} {y} {z}\n" for x, y, z in verts)
        h.writelines(f"f {a + 1} {b + 1} {c + 1}\n" for a, b, c in faces)
    pts = rng.uniform(-0.5, 0.5, size=(40, 3))
    nrm = rng.normal(size=(40, 3))
    nrm[[3, 17]] = 0.0
    np.savez(tmp_path / "pts.npz", points=pts, normals=nrm)
    a = types.SimpleNamespace(supercategory="BEHAVE", category="backpack", asset_id="behave_asset", obj_pth=str(tmp_path / "box.obj"),
                              asset_downsample_dir=str(tmp_path / "ads"), num_object_downsample_points_list=[40], simplify_method="poisson_disk",
                              points_pth=str(tmp_path / "pts.npz"), skip_done=False, debug=False, seed=1)
    out = do.main(a)
    assert out == [str(tmp_path / "ads" / "BEHAVE" / "backpack" / "behave_asset_40.pickle")]
    o = pickle.load(open(out[0], "rb"))
    assert set(o) == {"supercategory", "category", "asset_id", "V", "F", "N", "N_raw", "downsample_indices", "downsampled_pcd_points_raw",
                      "downsampled_pcd_normal_raw", "obj_vertices_original", "obj_faces_original", "obj_vertex_normals_original"}
    assert o["N"] == 40 and o["N_raw"] == 38 and len(o["downsampled_pcd_points_raw"]) == 38
    assert o["downsample_indices"] == [int(i) for i in orc.nearest_vertex(pts, verts)]
    # without supplied points the third-party sampler is refused, not silently replaced
    a.points_pth = None
    with pytest.raises(NotImplementedError):
        do.main(a)
