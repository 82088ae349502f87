"""Configuration tables of the inpainting stage, restated compactly (values: reference
constants/generation/prompts.py:63-163, constants/generation/assets.py:47-69, constants/generation/inpaint_ldm.py:1-19).
Only what the work-list builder of src/generation/inpaint.py reads."""

# (supercategory, category) -> asset ids that may be processed
CATEGORY2ASSET = {
    "Chair": {"Lounge Chair / Cafe Chair / Office Chair": ["0a5a346c-cc3b-4280-b358-ccd1c4d8a865"]},
    "motorcycle,bike": {"motorcycle,bike": ["9b9794dda0a6532215a11c390f7ca182"]},
    "umbrella": {"umbrella": ["85fto9rtgcvsx2itzy9rd0gwh7758d64"]},
    "frypan": {"frypan": ["77kk57qyyj3tivpp51tpjw6xia2ds9d9"]},
    "BEHAVE": {"backpack": ["behave_asset"]},
    "INTERCAP": {"suitcase": ["intercap_asset"]},
}

# per-category overrides of the diffuser settings (missing keys fall back to the CLI defaults)
SC2DIFFUSERCONFIG = {
    "Chair": {"Lounge Chair / Cafe Chair / Office Chair": {"strength": 1.0, "controlnet_conditioning_scale": 0.0}},
    "motorcycle,bike": {"motorcycle,bike": {"strength": 0.9, "controlnet_conditioning_scale": 0.0}},
    "umbrella": {"umbrella": {}},
    "frypan": {"frypan": {}},
    "BEHAVE": {"backpack": {"strength": 0.98}},
    "INTERCAP": {"suitcase": {"strength": 0.98}},
}

ALLOWED_VIEWPOINT_AUGMENTATIONS = [", full body", "original"]


def _views(n):
    return {f"view:{i:05}": {"view_text": [", full body", "original"]} for i in range(n)}


# per-(category, view) overrides; a view that is not listed falls back to SC2DIFFUSERCONFIG
SCV2DIFFUSERCONFIG = {
    "Chair": {"Lounge Chair / Cafe Chair / Office Chair": _views(8)},
    "motorcycle,bike": {"motorcycle,bike": _views(8)},
    "umbrella": {"umbrella": _views(40)},
    "frypan": {"frypan": _views(40)},
    "cart": {"cart": _views(8)},
    "BEHAVE": {"backpack": _views(40)},
    "INTERCAP": {"suitcase": _views(40)},
}

HF_MODEL_KEYS = {
    "sd2inpaint": "stabilityai/stable-diffusion-2-inpainting",
    "dreamshaper8": "Lykon/dreamshaper-8-inpainting",
    "absolutereal": "Lykon/absolute-realism-1.6525-inpainting",
    "realisticvision": "Uminosachi/realisticVisionV51_v51VAE-inpainting",
}
