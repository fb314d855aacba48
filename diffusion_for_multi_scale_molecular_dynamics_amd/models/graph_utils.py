"""src/.../models/graph_utils.py: `get_adj_matrix` under the reference's module path (implementation: utils/neighbors.py, on
the HIP radius graph)."""
from ..utils.neighbors import get_adj_matrix  # noqa: F401
