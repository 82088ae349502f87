"""Drop-in module path for the reference's ``utils/reproducibility.py``."""
from coma_amd.misc import seed_everything  # noqa: F401
