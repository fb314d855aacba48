"""Pair distances under periodic boundary conditions (src/.../utils/structure_utils.py:41-121), on the HIP radius graph.

Third consumer of kernel N1 (full mode: one edge per (source, destination, image) with its lattice shift).  The
reference returns the distances as an unordered bag (it feeds histograms / KS metrics); here they come out ordered by
(structure, source, destination, image).
"""
from typing import List

import torch

from .. import kernels


def compute_distances_in_batch(cartesian_positions: torch.Tensor, unit_cell: torch.Tensor,
                               max_distance: float) -> torch.Tensor:
    """All distances 0 < |p_i - (p_j + image)| <= max_distance over the 27 nearest images, for every structure."""
    batch_size, n_atoms, d = cartesian_positions.shape
    assert d in (1, 2, 3) and unit_cell.shape == (batch_size, d, d)
    from .neighbors import embed_in_three_dimensions
    cart, cell = embed_in_three_dimensions(cartesian_positions, unit_cell, max_distance)     # (1-D / 2-D: see there)
    out = kernels.radius_graph(cart, cell, max_distance, unique=False, status=None)
    edges, shifts = out["edges"], out["shifts"]
    structure = torch.repeat_interleave(torch.arange(batch_size, device=cart.device), out["counts"].sum(dim=1))
    flat = cart.reshape(batch_size * n_atoms, 3)
    base = structure * n_atoms
    displacement = flat.index_select(0, base + edges[:, 1]) + shifts - flat.index_select(0, base + edges[:, 0])
    return torch.linalg.norm(displacement, dim=1)


def get_orthogonal_basis_vectors(batch_size: int, cell_dimensions: List[float]) -> torch.Tensor:
    """[batch_size, d, d]: diag(cell_dimensions), once per structure (:124-139)."""
    return torch.diag(torch.tensor(cell_dimensions, dtype=torch.float32)).expand(batch_size, -1, -1).clone()


def compute_distances(cartesian_positions: torch.Tensor, basis_vectors: torch.Tensor, max_distance: float) -> torch.Tensor:
    """The lengths of the edges of the periodic radius graph (get_periodic_adjacency_information: the cutoff must be below the
    shortest cell-crossing distance, unlike compute_distances_in_batch's 27-image sweep) (:142-165)."""
    from .neighbors import get_periodic_adjacency_information
    info = get_periodic_adjacency_information(cartesian_positions, basis_vectors, radial_cutoff=max_distance)
    source, destination = info.adjacency_matrix
    displacement = cartesian_positions[info.edge_batch_indices, destination] - \
        cartesian_positions[info.edge_batch_indices, source] + info.shifts
    distances = torch.linalg.norm(displacement, dim=-1)
    return distances[distances > 0.0]
