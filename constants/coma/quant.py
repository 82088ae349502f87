"""Quantitative preset (values of the reference's constants/coma/quant.py:1-37)."""
QUANT_AFFORDANCE_EXTRACTION_HYPERPARAMS_DICT = {
    "quant:full": {
        "human_res": "750", "human_use_downsample_pcd_raw": False, "object_res": "2048", "object_use_downsample_pcd_raw": True,
        "principle_vec": [0, 0, 1], "sub_principle_vec": [0, 1, 0], "rel_dist_method": "dist",
        "spatial_grid_size": 0.04, "spatial_grid_thres": 0.1, "normal_gaussian_sigma": 0.2, "normal_res": 250, "spatial_res": 0,
        "eps": 1e-10, "significant_contact_ratio": 0.0, "enable_prefilter": False, "enable_postfilter": True,
        "standardize_human_scale": False, "scaler_range": (0.75, 1.25), "visualize_type": "none", "vis_example_num": 0,
        "quant_mode": True, "quant_keys": ["aggr_object_contact_metrics", "aggr_human_contact_metrics"],
    },
}
