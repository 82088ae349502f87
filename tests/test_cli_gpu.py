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
