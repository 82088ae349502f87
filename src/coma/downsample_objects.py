"""Write {asset_downsample_dir}/{SC}/{C}/{asset_id}_{N}.pickle (reference: src/coma/downsample_objects.py).

The reference walks its dataset tables (constants/generation/assets.py: BEHAVE / 3D-FUTURE / ... paths, not shipped here);
this CLI takes one asset per call: --obj_pth plus its (supercategory, category, asset_id).  Sampled points may be supplied
(--points_pth .npz {points, normals}, e.g. exported from open3d's Poisson-disk sampler) or drawn uniformly."""
import argparse
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run_downsampling(supercategory, category, asset_id, obj_pth, number_of_points, simplify_method, debug=False, points_pth=None, seed=42,
                     device="cuda"):
    from coma_amd.downsample import downsample_object, load_obj
    vertices, faces = load_obj(obj_pth)
    pts = nrm = None
    if points_pth:
        z = np.load(points_pth)
        pts, nrm = z["points"], z["normals"]
    return downsample_object(supercategory, category, asset_id, vertices, faces, number_of_points, points=pts, point_normals=nrm,
                             simplify_method=simplify_method, seed=seed, device=device)


def main(args):
    sc_str, c_str = args.supercategory.replace("/", ":"), args.category.replace("/", ":")
    out = []
    for n in args.num_object_downsample_points_list:
        save_pth = f"{args.asset_downsample_dir}/{sc_str}/{c_str}/{args.asset_id}_{n}.pickle"
        if args.skip_done and os.path.exists(save_pth):
            continue
        to_save = run_downsampling(args.supercategory, args.category, args.asset_id, args.obj_pth, n, args.simplify_method, args.debug,
                                   args.points_pth, args.seed)
        os.makedirs(os.path.dirname(save_pth), exist_ok=True)
        with open(save_pth, "wb") as handle:
            pickle.dump(to_save, handle, protocol=pickle.HIGHEST_PROTOCOL)
        out.append(save_pth)
    return out


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--supercategory", type=str, required=True)
    p.add_argument("--category", type=str, required=True)
    p.add_argument("--asset_id", type=str, required=True)
    p.add_argument("--obj_pth", type=str, required=True)
    p.add_argument("--asset_downsample_dir", type=str, default="results/coma/asset_downsample")
    p.add_argument("--num_object_downsample_points_list", type=int, nargs="+", default=[180, 1500, 2048])
    p.add_argument("--simplify_method", choices=["poisson_disk", "uniform"], default="poisson_disk")
    p.add_argument("--points_pth", type=str, default=None)
    p.add_argument("--skip_done", action="store_true")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--seed", type=int, default=42)
    return p


if __name__ == "__main__":
    a = build_parser().parse_args()
    from utils.reproducibility import seed_everything
    seed_everything(a.seed)
    print(main(a))
