"""Qualitative ComA hyper-parameter presets (same keys and values as the reference's constants/coma/qual.py:1-75;
every preset inherits the keys it does not set from "qual:001").  Also registers the aliases that
scripts/learn_coma.sh:50-52 passes (`qual:<category>_object|_human|_occupancy`), which the reference's argparse
`choices` rejects."""

_BASE = {
    "human_res": "FULL", "human_use_downsample_pcd_raw": False,
    "object_res": "180", "object_use_downsample_pcd_raw": True,
    "principle_vec": [0, 0, 1], "sub_principle_vec": [0, 1, 0], "rel_dist_method": "dist",
    "spatial_grid_size": 0.06, "spatial_grid_thres": 0.24, "normal_gaussian_sigma": 0.2,
    "normal_res": 250, "spatial_res": 0, "eps": 1e-10, "significant_contact_ratio": 0.3,
    "enable_postfilter": True, "standardize_human_scale": False, "scaler_range": (0.75, 1.25),
    "visualize_type": "aggr-human-contact", "vis_example_num": 0, "quant_mode": False, "quant_keys": [],
}
_COMMON = {"standardize_human_scale": False, "scaler_range": (0.75, 1.25)}
_OVERRIDES = {
    "qual:001": {},
    "qual:backpack_human_contact": {**_COMMON, "spatial_grid_size": 0.07, "spatial_grid_thres": 0.03, "normal_gaussian_sigma": 0.25,
                                    "significant_contact_ratio": 0.1, "visualize_type": "aggr-human-contact"},
    "qual:backpack_object_contact": {**_COMMON, "spatial_grid_size": 0.15, "spatial_grid_thres": 0.05, "normal_gaussian_sigma": 0.25,
                                     "significant_contact_ratio": 0.1, "human_res": "1000", "human_use_downsample_pcd_raw": False,
                                     "object_res": "1500", "object_use_downsample_pcd_raw": True,
                                     "visualize_type": "aggr-object-contact"},
    "qual:backpack_occupancy": {**_COMMON, "spatial_res": 30, "normal_res": 0, "human_res": "FULL",
                                "human_use_downsample_pcd_raw": False, "object_res": "1500",
                                "object_use_downsample_pcd_raw": False, "visualize_type": "occupancy"},
    "qual:backpack_orientation": {**_COMMON, "spatial_grid_size": 0.03, "spatial_grid_thres": 0.1, "normal_gaussian_sigma": 0.2,
                                  "significant_contact_ratio": 0.1, "visualize_type": "orientation", "vis_example_num": 1},
}
_ALIASES = {"qual:backpack_human": "qual:backpack_human_contact", "qual:backpack_object": "qual:backpack_object_contact"}

QUAL_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT = {k: {**_BASE, **v} for k, v in _OVERRIDES.items()}
for _alias, _target in _ALIASES.items():
    QUAL_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT[_alias] = dict(QUAL_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT[_target])
