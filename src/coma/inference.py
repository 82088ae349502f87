"""Reduce a saved ComA pickle to its affordance artefact on MI355X.

CLI surface of the reference's ``src/coma/inference.py`` (:150-182): the same flags
``--supercategory --category --coma_path --visualize_type --smplx_downsample_pth --asset_downsample_pth
--hyperparams_key --output_dir --seed`` and the same outputs under ``{output_dir}/{supercategory}/{category}/``:
``human_contact.npy`` (x / x.max()), ``object_contact.ply`` (jet-coloured point cloud), ``orientational_tendency.npy``
(min-max normalised entropy score of object point 0) and ``occupancy.npy`` (dict: 0.7 * field / max + grid metadata).
As in the reference the branch taken is the preset's ``visualize_type`` (the flag of that name is parsed and ignored,
:46).  The reference file does not import as shipped (``constants.coma.coma_basic_settings`` is missing, :17); here every
preset comes from ``constants/coma/{qual,quant}.py``.  All reductions run in libcoma_hip.so.
"""
import argparse
import os
import pickle
import sys
from copy import deepcopy

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from constants.coma.qual import QUAL_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT  # noqa: E402
from constants.coma.quant import QUANT_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT  # noqa: E402
from constants.metadata import DEFAULT_SEED  # noqa: E402
from utils.coma import ComA, get_aggregated_contact  # noqa: E402
from utils.coma_occupancy import ComA_Occupancy  # noqa: E402
from utils.reproducibility import seed_everything  # noqa: E402


def jet_rgb(score):
    """matplotlib's 'jet' colormap on [0,1] (the reference colours the object point cloud with it)."""
    from matplotlib import cm
    import matplotlib as mpl
    return cm.ScalarMappable(norm=mpl.colors.Normalize(vmin=0.0, vmax=1.0), cmap="jet").to_rgba(score)[:, :3]


def write_ply_pointcloud(path, points, normals, colors):
    """ASCII PLY with per-point normals and uchar colours (what open3d's write_point_cloud stores; open3d is not
    available offline)."""
    points, normals = np.asarray(points, np.float64), np.asarray(normals, np.float64)
    rgb = np.clip(np.round(np.asarray(colors) * 255.0), 0, 255).astype(np.uint8)
    with open(path, "w") as fh:
        fh.write("ply\nformat ascii 1.0\n" + f"element vertex {len(points)}\n"
                 "property double x\nproperty double y\nproperty double z\n"
                 "property double nx\nproperty double ny\nproperty double nz\n"
                 "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
        for p, n, c in zip(points, normals, rgb):
            fh.write(f"{p[0]:.9g} {p[1]:.9g} {p[2]:.9g} {n[0]:.9g} {n[1]:.9g} {n[2]:.9g} {c[0]} {c[1]} {c[2]}\n")


def inference(supercategory, category, coma_path, smplx_downsample_pth, asset_downsample_pth, visualize_type, hyperparams_key,
              hyperparams, output_dir, device="cuda"):
    hp = hyperparams
    visualize_type = hp["visualize_type"]                 # the preset decides, as in the reference
    with open(smplx_downsample_pth, "rb") as handle:
        human_meta = pickle.load(handle)
    with open(asset_downsample_pth, "rb") as handle:
        object_meta = deepcopy(pickle.load(handle))
    H = human_meta["N_raw"] if hp["human_use_downsample_pcd_raw"] else human_meta["N"]
    O = object_meta["N_raw"] if hp["object_use_downsample_pcd_raw"] else object_meta["N"]
    common = dict(human_res=H, obj_res=O, normal_res=hp["normal_res"], spatial_res=hp["spatial_res"],
                  proximity_settings=dict(spatial_grid_size=hp["spatial_grid_size"], spatial_grid_thres=hp["spatial_grid_thres"]),
                  principle_vec=hp["principle_vec"], sub_principle_vec=hp["sub_principle_vec"],
                  rel_dist_method=hp["rel_dist_method"], normal_gaussian_sigma=hp["normal_gaussian_sigma"], eps=hp["eps"],
                  device=device)
    coma = ComA_Occupancy(scale_tolerance=3.0, **common) if visualize_type == "occupancy" else ComA(**common)
    coma.load(coma_path)
    out = f"{output_dir}/{supercategory}/{category}"
    os.makedirs(out, exist_ok=True)

    if visualize_type == "aggr-human-contact":
        agg, _ = get_aggregated_contact(coma=coma, contact_map_type="human", significant_contact_ratio=hp["significant_contact_ratio"])
        np.save(f"{out}/human_contact.npy", agg / agg.max())
    elif visualize_type == "aggr-object-contact":
        agg, _ = get_aggregated_contact(coma=coma, contact_map_type="obj", significant_contact_ratio=hp["significant_contact_ratio"])
        score = agg / agg.max()
        write_ply_pointcloud(f"{out}/object_contact.ply", object_meta["downsampled_pcd_points_raw"],
                             object_meta["downsampled_pcd_normal_raw"], jet_rgb(score))
    elif visualize_type == "orientation":
        s = coma.compute_nonphysical_response_sphere(n_bin=1e6, nonphysical_type="human", as_numpy=True)["human"][:, 0]
        np.save(f"{out}/orientational_tendency.npy", (s - s.min()) / (s.max() - s.min()))
    elif visualize_type == "occupancy":
        field = coma.return_aggregated_spatial_grids(human_indices=None).cpu().numpy()
        field /= field.max()
        np.save(f"{out}/occupancy.npy", dict(prob_field=0.7 * field, spatial_grid_metadata=coma.spatial_grid_metadata))
    return out


def _presets():
    table = dict(QUAL_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT)
    table.update(QUANT_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT)
    return table


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--supercategory", type=str)
    parser.add_argument("--category", type=str)
    parser.add_argument("--coma_path", type=str)
    parser.add_argument("--visualize_type", type=str, choices=["aggr-human-contact", "aggr-object-contact", "orientation", "occupancy"])
    parser.add_argument("--smplx_downsample_pth", type=str)
    parser.add_argument("--asset_downsample_pth", type=str)
    parser.add_argument("--hyperparams_key", type=str)
    parser.add_argument("--output_dir", type=str, default="output")
    parser.add_argument("--seed", type=int, default=DEFAULT_SEED)
    args = parser.parse_args()
    seed_everything(args.seed)
    assert args.hyperparams_key is not None, "You must Specify the 'args.hypeparams_key'"
    inference(supercategory=args.supercategory, category=args.category, coma_path=args.coma_path,
              visualize_type=args.visualize_type, smplx_downsample_pth=args.smplx_downsample_pth,
              asset_downsample_pth=args.asset_downsample_pth, hyperparams_key=args.hyperparams_key,
              hyperparams=_presets()[args.hyperparams_key], output_dir=args.output_dir)
