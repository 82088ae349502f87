"""Drop-in module path for the reference's ``utils/coma.py``: every public name resolves to the MI355X
implementation in :mod:`coma_amd.coma` (HIP kernels behind a C ABI)."""
from coma_amd.coma import (  # noqa: F401
    ComA,
    get_aggregated_contact,
    get_nonphysical_score,
    get_uniform_points_on_sphere,
    nearest_vertex_indices,
    negative_exp,
    simplify_mesh_and_get_indices,
)
from coma_amd.ingest import prepare_affordance_extraction_inputs  # noqa: F401,E402
