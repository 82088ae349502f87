"""Drop-in module path for the reference's ``utils/misc.py`` helpers used on the ComA path."""
from coma_amd.misc import get_3d_indexgrid_ijk, to_np_torch_recursive  # noqa: F401
