"""Learn ComA from the 3D HOI samples of an asset on MI355X (one process per GPU; RCCL all-reduce when launched with
torch.distributed).

CLI surface and host rules of the reference's ``src/coma/extract_coma.py``:
  * flags (:506-534) and defaults; presets from constants/coma/{qual,quant}.py;
  * SCAM discovery over ``{human_sample_dir}/{SC}/{C}/{asset}/{view}/{mask}/{prompt}/{id}.pickle`` with the
    ``mainprompt`` rule ("total" if the first comma field starts a "total:" prompt, else that field) (:148-173);
  * post-filter membership test on (view_id, asset_mask_id, prompt, inpaint_id) against
    ``{postfilter_dir}/{SC}/{C}/{asset}/{mainprompt}.json`` (:29-63); string sentinels are skipped (:233-243);
  * H / O from ``N`` vs ``N_raw`` of the down-sample pickles (:293-294); metadata json; skip_done reload (:350-351);
  * export BEFORE the reducers normalise the state (:426 then :428+); the four artefacts and their normalisations.
Differences by design: samples are ingested in batches (device vertex normals, coma_amd/ingest.py) and accumulated by
the fused HIP kernel; with WORLD_SIZE > 1 the sample list of each SCAM is sharded across ranks and the partial states are
summed with one RCCL all-reduce before export (rank 0 writes).
"""
import argparse
import json
import os
import pickle
import sys
from copy import deepcopy
from glob import glob

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from constants.coma.qual import QUAL_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT  # noqa: E402
from constants.coma.quant import QUANT_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT  # noqa: E402
from constants.metadata import DEFAULT_SEED  # noqa: E402

# occupancy grids larger than this leave a multi-GPU run as one pickle per rank (row slices) instead of being gathered onto rank 0
SHARD_EXPORT_BYTES = int(os.environ.get("COMA_SHARD_EXPORT_BYTES", 8 << 30))
KNOWN_SENTINELS = ["NOT ALLOWED VIEWPOINT PROMPTS", "ERRONEOUS SAMPLE DUE TO TOO SMALL HUMAN", "TOO LITTLE INLIERS",
                   "LARGELY PENETRATED HUMAN"]


def mainprompt_of(prompt_dirname):
    head = prompt_dirname.split(",")[0]
    return "total" if "total:" in head else head


def discover_scams(human_sample_dir, supercategories=None, categories=None, prompts=None):
    """Sorted unique (supercategory, category, asset_id, mainprompt) with at least one sample file."""
    scams = set()
    for pth in sorted(set(glob(f"{human_sample_dir}/*/*/*/*/*/*/*.pickle"))):
        sc_str, c_str, asset_id, _, _, prompt, _ = pth.split("/")[-7:]
        sc, c, mp = sc_str.replace(":", "/"), c_str.replace(":", "/"), mainprompt_of(prompt)
        if supercategories is not None and sc.lower() not in supercategories:
            continue
        if categories is not None and c.lower() not in categories:
            continue
        if prompts is not None and mp.lower() not in prompts:
            continue
        scams.add((sc, c, asset_id, mp))
    return sorted(scams)


class PostFilter:
    """Cache of the per-(SCAM) lists written by src/coma/filter.py; a sample survives only if its key tuple is listed."""

    def __init__(self, enable, root):
        self.enable, self.root, self.cache = enable, root, {}

    def rejects(self, sc, c, asset_id, view_id, mask_id, prompt, mainprompt, inpaint_id):
        if not self.enable:
            return False
        key = (sc, c, asset_id, mainprompt)
        if key not in self.cache:
            pth = f"{self.root}/{sc.replace('/', ':')}/{c.replace('/', ':')}/{asset_id}/{mainprompt}.json"
            assert os.path.exists(pth), pth
            with open(pth, "r") as rf:
                self.cache[key] = [tuple(x) for x in json.load(rf)]
        return (view_id, mask_id, prompt, inpaint_id) not in self.cache[key]


def collect_inputs(scam, human_sample_dir, postfilter, enable_postfilter):
    sc, c, asset_id, mp = scam
    sc_str, c_str = sc.replace("/", ":"), c.replace("/", ":")
    kept = []
    for pth in sorted(set(glob(f"{human_sample_dir}/{sc_str}/{c_str}/{asset_id}/*/*/{mp}*/*.pickle"))):
        k1, k2, k3, view_id, mask_id, prompt, id_ext = pth.split("/")[-7:]
        assert (k1, k2, k3) == (sc_str, c_str, asset_id) and mainprompt_of(prompt) == mp, (pth, mp)
        inpaint_id, ext = id_ext.split(".")
        assert ext == "pickle", "Human Finals must have '.pickle' extension"
        if postfilter.rejects(sc, c, asset_id, view_id, mask_id, prompt, mp, inpaint_id):
            continue
        with open(pth, "rb") as handle:
            data = pickle.load(handle)
        if isinstance(data, str):
            if data in KNOWN_SENTINELS:
                assert not enable_postfilter, pth
                continue
            assert False, "What more errors could there be?"
        kept.append(pth)
    return kept


def parse_selected_object_indices(text):
    if text == "":
        return None
    out = []
    for tok in text.split(" "):
        if "-" in tok:
            a, b = tok.split("-")
            out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(tok))
    return sorted(set(out)) if out else None


def run_affordance_extraction(supercategories, categories, prompts, camera_dir, human_params_dir, asset_downsample_dir,
                              human_postfilter_dir, human_sample_dir, coma_save_dir, affordance_save_dir, smplx_downsample_dir,
                              hyperparams, hyperparams_key, scale_tolerance, skip_done, device="cuda"):
    import torch
    import torch.distributed as dist
    from coma_amd.dist import shard_slice
    from src.coma.inference import jet_rgb, write_ply_pointcloud
    from utils.coma import ComA, get_aggregated_contact, prepare_affordance_extraction_inputs
    from utils.coma_occupancy import ComA_Occupancy

    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    hp = hyperparams
    visualize_type = hp["visualize_type"]
    with open(f"{smplx_downsample_dir}/smplx_star_downsampled_{hp['human_res']}.pickle", "rb") as handle:
        human_meta = pickle.load(handle)
    postfilter = PostFilter(hp["enable_postfilter"], human_postfilter_dir)
    done = []
    for scam in discover_scams(human_sample_dir, supercategories, categories, prompts):
        sc, c, asset_id, mp = scam
        if hp["quant_mode"] and mp != "total":
            continue
        sc_str, c_str = sc.replace("/", ":"), c.replace("/", ":")
        inputs = collect_inputs(scam, human_sample_dir, postfilter, hp["enable_postfilter"])
        if not inputs:
            continue
        with open(f"{asset_downsample_dir}/{sc_str}/{c_str}/{asset_id}_{hp['object_res']}.pickle", "rb") as handle:
            object_meta = deepcopy(pickle.load(handle))
        H = human_meta["N_raw"] if hp["human_use_downsample_pcd_raw"] else human_meta["N"]
        O = object_meta["N_raw"] if hp["object_use_downsample_pcd_raw"] else object_meta["N"]
        save_dir = f"{coma_save_dir}/{sc_str}/{c_str}/{asset_id}"
        json_pth, save_pth = f"{save_dir}/{hyperparams_key}:{mp}.json", f"{save_dir}/{hyperparams_key}:{mp}.pickle"
        if rank == 0 and not os.path.exists(json_pth):
            os.makedirs(save_dir, exist_ok=True)
            with open(json_pth, "w") as wf:
                json.dump(dict(input_human_pths=inputs, asset_downsample_pth=f"{asset_downsample_dir}/{sc_str}/{c_str}/{asset_id}_{hp['object_res']}.pickle",
                               result_save_dir=save_dir, result_json_pth=json_pth, result_save_pth=save_pth, H=H, O=O,
                               **{k: v for k, v in hp.items()}), wf, indent=1)
        common = dict(human_res=H, obj_res=O, normal_res=hp["normal_res"], spatial_res=hp["spatial_res"],
                      proximity_settings=dict(spatial_grid_size=hp["spatial_grid_size"], spatial_grid_thres=hp["spatial_grid_thres"]),
                      principle_vec=hp["principle_vec"], sub_principle_vec=hp["sub_principle_vec"], rel_dist_method=hp["rel_dist_method"],
                      normal_gaussian_sigma=hp["normal_gaussian_sigma"], eps=hp["eps"], device=device)
        # multi-GPU: contact / orientation shard SAMPLES (one SUM all-reduce of the state at the end); occupancy
        # shards human-vertex ROWS -- every rank reads all samples but keeps only its rows of the [H, R^3] grid, so the only
        # collective is a MAX all-reduce of the [R,R,R] field (a SUM over sample shards would move the whole grid: 88 GB at cfg 5)
        occ_rows = visualize_type == "occupancy" and world > 1
        row_lo, row_hi = shard_slice(H, rank, world) if occ_rows else (0, H)
        field_dev = None
        if visualize_type == "occupancy":
            coma = ComA_Occupancy(scale_tolerance=scale_tolerance, **dict(common, human_res=row_hi - row_lo))
        else:
            coma = ComA(**common)
        # config 5 (H = 10475, R = 128: an 88 GB grid) never gathers the rows: every rank pickles its own row slice
        # (`..._rank{r}.pickle`, ComA_Occupancy.export(shard=...)); small grids (R = 30) keep the reference's single file
        shard_export = occ_rows and 4 * H * hp["spatial_res"] ** 3 > SHARD_EXPORT_BYTES
        # skip-or-compute is decided ONCE, on rank 0, and broadcast: the ranks of one scene must take the same branch (both contain
        # collectives), whatever a slow or non-shared filesystem shows each of them
        have = skip_done and (os.path.exists(save_pth) or (visualize_type == "occupancy" and bool(ComA_Occupancy.shard_files(save_pth))))
        if world > 1:
            flag = [bool(have)]
            dist.broadcast_object_list(flag, src=0)
            have = flag[0]
        if have:
            if occ_rows:               # every rank reloads ITS rows (its own shard file when the set was written by this world size)
                coma.load(save_pth, shard=(rank, world))
                from coma_amd.dist import occupancy_rows_reduce
                _, field_dev = occupancy_rows_reduce(coma, H, gather=False)
            else:
                coma.load(save_pth)
        else:
            lo, hi = (0, len(inputs)) if occ_rows else shard_slice(len(inputs), rank, world)
            for pth in inputs[lo:hi]:
                _, _, _, view_id, mask_id, prompt, id_ext = pth.split("/")[-7:]
                params = f"{human_params_dir}/{sc_str}/{c_str}/{asset_id}/{view_id}/{mask_id}/{prompt.replace('total:', '')}/{id_ext}"
                x = prepare_affordance_extraction_inputs(
                    human_mesh_pth=pth, human_mesh_pth_type="pickle", human_downsample_metadata=human_meta,
                    object_downsample_metadata=object_meta, human_use_downsample_pcd_raw=hp["human_use_downsample_pcd_raw"],
                    object_use_downsample_pcd_raw=hp["object_use_downsample_pcd_raw"], eps=hp["eps"],
                    standardize_human_scale=hp["standardize_human_scale"], scaler_range=hp["scaler_range"],
                    camera_pth=f"{camera_dir}/{sc_str}/{c_str}/{asset_id}/{view_id}.pickle", human_params_pth=params, device=device)
                if x is None:
                    continue
                coma.register_sample_to_cache(human_verts=x["human_verts"][row_lo:row_hi], human_normals=x["human_vertex_normals"][row_lo:row_hi],
                                              obj_verts=x["obj_verts"], obj_normals=x["obj_vertex_normals"])
            coma.aggregate_all_samples()
            if occ_rows:
                from coma_amd.dist import occupancy_rows_reduce
                if shard_export:
                    # export first: it runs the one fused pass (raw counts left in place, the field of all rows kept aside), the
                    # reduction below then only picks that field up and MAX-all-reduces it -- no row leaves its rank
                    os.makedirs(save_dir, exist_ok=True)
                    coma.export(save_pth=save_pth, shard=(rank, world, H))
                    _, field_dev = occupancy_rows_reduce(coma, H, gather=False)
                else:
                    full, field_dev = occupancy_rows_reduce(coma, H)
                    if rank == 0:      # the exported object carries the complete raw per-vertex grid, as a single process would
                        used, used_count = coma.used, coma.used_count
                        coma = ComA_Occupancy(scale_tolerance=scale_tolerance, **common)
                        coma.spatial_occupancy_grids, coma.used, coma.used_count = full, used, used_count
            elif world > 1:
                coma.all_reduce()
            if rank == 0 and not shard_export:
                os.makedirs(save_dir, exist_ok=True)
                coma.export(save_pth=save_pth)
        # K4 reducers: with more than one rank every rank holds the all-reduced (or reloaded) state; each
        # normalises / reduces its own human rows and the small vectors are combined (coma_amd/dist.py) -- after the export above,
        # which must see the raw state (export-before-normalise, src/coma/extract_coma.py:426 of the reference)
        agg = score = None
        if world > 1 and visualize_type in ("aggr-human-contact", "aggr-object-contact"):
            from coma_amd.dist import aggregated_contact_row_parallel
            agg, _ = aggregated_contact_row_parallel(coma, "human" if visualize_type == "aggr-human-contact" else "obj",
                                                     hp["significant_contact_ratio"])
        elif world > 1 and visualize_type == "orientation":
            from coma_amd.dist import nonphysical_score_row_parallel
            score = nonphysical_score_row_parallel(coma, "human")
        if rank == 0:
            out = f"{affordance_save_dir}/{sc}/{c}/{asset_id}/{hyperparams_key}:{mp}"
            os.makedirs(out, exist_ok=True)
            if visualize_type == "aggr-human-contact":
                if agg is None:
                    agg, _ = get_aggregated_contact(coma=coma, contact_map_type="human", significant_contact_ratio=hp["significant_contact_ratio"])
                np.save(f"{out}/human_contact.npy", agg / agg.max())
            elif visualize_type == "aggr-object-contact":
                if agg is None:
                    agg, _ = get_aggregated_contact(coma=coma, contact_map_type="obj", significant_contact_ratio=hp["significant_contact_ratio"])
                write_ply_pointcloud(f"{out}/object_contact.ply", object_meta["downsampled_pcd_points_raw"],
                                     object_meta["downsampled_pcd_normal_raw"], jet_rgb(agg / agg.max()))
            elif visualize_type == "orientation":
                s = (score if score is not None else
                     coma.compute_nonphysical_response_sphere(n_bin=1e6, nonphysical_type="human", as_numpy=True)["human"])[:, 0]
                np.save(f"{out}/orientational_tendency.npy", (s - s.min()) / (s.max() - s.min()))
            elif visualize_type == "occupancy":
                field = (field_dev if field_dev is not None else coma.return_aggregated_spatial_grids(human_indices=None)).cpu().numpy()
                field /= field.max()
                np.save(f"{out}/occupancy.npy", dict(prob_field=0.7 * field, spatial_grid_metadata=coma.spatial_grid_metadata))
            done.append((scam, save_pth, out))
        del coma
        if world > 1:
            dist.barrier()             # no rank runs ahead into the next scene's collectives (or its skip decision)
    return done


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--supercategories", type=str, nargs="+")
    p.add_argument("--categories", type=str, nargs="+")
    p.add_argument("--prompts", type=str, nargs="+")
    p.add_argument("--camera_dir", type=str, default="results/generation/cameras")
    p.add_argument("--human_params_dir", type=str, default="results/generation/human_preds")
    p.add_argument("--asset_downsample_dir", type=str, default="results/coma/asset_downsample")
    p.add_argument("--human_postfilter_dir", type=str, default="results/coma/human_postfilterings")
    p.add_argument("--human_sample_dir", type=str, default="results/generation/human_sample")
    p.add_argument("--coma_save_dir", type=str, default="results/coma/extracted_coma")
    p.add_argument("--affordance_save_dir", type=str, default="results/coma/affordance")
    p.add_argument("--smplx_canon_obj_pth", type=str, default="./constants/mesh/smplx_star.obj")
    p.add_argument("--hyperparams_key", choices=list(QUANT_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT.keys()) + list(QUAL_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT.keys()))
    p.add_argument("--visualize", action="store_true")
    p.add_argument("--vis_example_num", type=int)
    p.add_argument("--interactive", action="store_true")
    p.add_argument("--vis_interactive", action="store_true")
    p.add_argument("--fovy", type=float, default=27.5)
    p.add_argument("--tmp_cache_dir", type=str, default="results/coma_tmp_cache")
    p.add_argument("--selected_object_indices", type=str, help="Type as '21 22' or '21-25'", default="")
    p.add_argument("--scale_tolerance", type=float, default=3.0)
    p.add_argument("--skip_done", action="store_true")
    p.add_argument("--seed", type=int, default=DEFAULT_SEED)
    p.add_argument("--smplx_downsample_dir", type=str, default="./constants/mesh")     # addition: where the human down-sample pickles live
    return p


if __name__ == "__main__":
    args = build_parser().parse_args()
    for name in ("supercategories", "categories", "prompts"):
        if getattr(args, name) is not None:
            setattr(args, name, [x.lower() for x in getattr(args, name)])
    from utils.reproducibility import seed_everything
    seed_everything(args.seed)
    assert args.hyperparams_key is not None, "You must Specify the 'args.hypeparams_key'"
    table = dict(QUAL_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT)
    table.update(QUANT_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group("nccl")
    run_affordance_extraction(args.supercategories, args.categories, args.prompts, args.camera_dir, args.human_params_dir,
                              args.asset_downsample_dir, args.human_postfilter_dir, args.human_sample_dir, args.coma_save_dir,
                              args.affordance_save_dir, args.smplx_downsample_dir, table[args.hyperparams_key], args.hyperparams_key,
                              args.scale_tolerance, args.skip_done)
