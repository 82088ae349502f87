"""Drop-in module path for the reference's ``utils/transformations.py`` normalisers."""
from coma_amd.misc import normalize_vectors_np, normalize_vectors_torch  # noqa: F401
