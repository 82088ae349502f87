"""Drop-in module path for the reference's ``utils/coma_occupancy.py`` (see :mod:`coma_amd.coma_occupancy`)."""
from coma_amd.coma_occupancy import ComA_Occupancy, load_voxelgrid  # noqa: F401
