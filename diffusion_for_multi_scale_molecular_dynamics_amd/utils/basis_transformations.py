"""Geometry helpers of the reference API (src/.../utils/basis_transformations.py), host-side plumbing only."""
import torch


def get_number_of_lattice_parameters(spatial_dimension: int) -> int:
    """:219-221"""
    return int(spatial_dimension * (spatial_dimension + 1) / 2)


def get_positions_from_coordinates(relative_coordinates: torch.Tensor, basis_vectors: torch.Tensor) -> torch.Tensor:
    """p = x @ [a1; a2; a3]  (:34-57)"""
    return torch.matmul(relative_coordinates, basis_vectors)


def map_lattice_parameters_to_unit_cell_vectors(lattice_parameters: torch.Tensor) -> torch.Tensor:
    """Orthogonal boxes only, like the reference (:141-170); the angle check is done without a device sync."""
    nl = lattice_parameters.shape[-1]
    d = int((-1 + (1 + 8 * nl) ** 0.5) / 2)
    return torch.diag_embed(lattice_parameters[..., :d])


def map_unit_cell_to_lattice_parameters(unit_cell: torch.Tensor) -> torch.Tensor:
    """:229-270 (torch engine)"""
    d = unit_cell.shape[-1]
    out = torch.zeros(*unit_cell.shape[:-2], get_number_of_lattice_parameters(d)).to(unit_cell)
    out[..., :d] = torch.diagonal(unit_cell, dim1=-2, dim2=-1)
    return out
