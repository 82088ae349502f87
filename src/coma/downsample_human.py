"""Write ./constants/mesh/smplx_star_downsampled_{N}.pickle (reference: src/coma/downsample_human.py).

The reference builds the SMPL-X "star" pose mesh itself (smplx + model files: third party, absent here); this CLI takes that
mesh as --mesh_pth (the reference's own ./constants/mesh/smplx_star.pickle {vertices, faces}, or an .obj) and, optionally,
the sampled points exported from open3d (--points_pth: .npz with points [N,3] and normals [N,3]).  Index map, normals and
the zero-normal filter run through coma_amd (HIP)."""
import argparse
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_mesh(pth):
    if pth.endswith(".obj"):
        from coma_amd.downsample import load_obj
        return load_obj(pth)
    with open(pth, "rb") as handle:
        d = pickle.load(handle)
    return np.asarray(d["vertices"]), np.asarray(d["faces"])


def downsample_smplx(args, device="cuda"):
    from coma_amd.downsample import downsample_human
    vertices, faces = load_mesh(args.mesh_pth)
    pts = nrm = None
    if args.points_pth:
        z = np.load(args.points_pth)
        pts, nrm = z["points"], z["normals"]
    n = args.num_human_downsample_points
    to_save = downsample_human(vertices, faces, n, points=pts, point_normals=nrm, simplify_method=args.simplify_method, seed=args.seed, device=device)
    name = f"smplx_star_downsampled_{n}.pickle" if n < len(vertices) else "smplx_star_downsampled_FULL.pickle"
    save_pth = os.path.join(args.save_dir, name)
    if not args.skip_done or not os.path.exists(save_pth):
        os.makedirs(args.save_dir, exist_ok=True)
        with open(save_pth, "wb") as handle:
            pickle.dump(to_save, handle, protocol=pickle.HIGHEST_PROTOCOL)
    return save_pth


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--simplify_method", choices=["poisson_disk", "uniform"], default="poisson_disk")
    p.add_argument("--num_human_downsample_points_list", type=int, nargs="+", default=[1000, 1500, 2000, 2048, 20000])
    p.add_argument("--use_watertight", action="store_true")
    p.add_argument("--skip_done", action="store_true")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--seed", type=int, default=42)
    # additions: where the star-pose mesh and (for poisson_disk) the open3d-sampled points come from, where to write
    p.add_argument("--mesh_pth", type=str, default="./constants/mesh/smplx_star.pickle")
    p.add_argument("--points_pth", type=str, default=None, help=".npz {points, normals}; required for poisson_disk")
    p.add_argument("--save_dir", type=str, default="./constants/mesh")
    return p


if __name__ == "__main__":
    args = build_parser().parse_args()
    from utils.reproducibility import seed_everything
    seed_everything(args.seed)
    for n in args.num_human_downsample_points_list:
        args.num_human_downsample_points = n
        print(downsample_smplx(args))
