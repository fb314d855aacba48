"""Pair distances under periodic boundary conditions (src/.../utils/structure_utils.py:41-121), on the HIP radius graph.

Third consumer of kernel N1 (full mode: one edge per (source, destination, image) with its lattice shift).  The
reference returns the distances as an unordered bag (it feeds histograms / KS metrics); here they come out ordered by
(structure, source, destination, image).
"""
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
