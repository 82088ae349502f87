"""Drop-in module path for the reference's ``utils/adaptive_mask_inpainting.py``; implementation in
:mod:`coma_amd.sd.pipeline` (HIP kernels behind include/sd_hip.h)."""
from coma_amd.sd.pipeline import (  # noqa: F401
    AdaptiveMaskInpaintPipeline,
    MaskDilateScheduler,
    ProvokeScheduler,
    SyntheticHumanMaskPredictor,
    default_adaptive_mask_settings,
    merge_bbox,
    prepare_mask_and_masked_image,
    seg2bbox,
)
from coma_amd.sd.scheduler import DDIMScheduler  # noqa: F401
