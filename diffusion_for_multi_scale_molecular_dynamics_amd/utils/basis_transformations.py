"""Geometry helpers of the reference API (src/.../utils/basis_transformations.py), host-side plumbing only: every public
function of that module is here under its name and signature."""
import numpy as np
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


def get_spatial_dimension_from_number_of_lattice_parameters(number_of_lattice_parameters: int) -> int:
    """d from d (d + 1) / 2  (:178-182)."""
    return int((-1 + np.sqrt(1 + 8 * number_of_lattice_parameters)) / 2)


def get_reciprocal_basis_vectors(basis_vectors: torch.Tensor) -> torch.Tensor:
    """B with A B = I for the row-vector cell A = [a1; a2; a3]  (:9-31)."""
    return torch.inverse(basis_vectors)


def get_relative_coordinates_from_cartesian_positions(cartesian_positions: torch.Tensor,
                                                      reciprocal_basis_vectors: torch.Tensor) -> torch.Tensor:
    """x = p @ B  (:60-92)."""
    return torch.matmul(cartesian_positions, reciprocal_basis_vectors)


def map_unit_cell_to_lattice_parameters(unit_cell, engine: str = "torch"):
    """[..., d, d] cell -> [..., d (d + 1) / 2] lattice parameters: the diagonal, angles zero (:185-222).  engine "numpy": ONE
    cell [d, d] as a numpy array, like the reference's branch."""
    assert engine in ["torch", "numpy"], f"Mapping can be done for numpy or torch. Got {engine}."
    d = unit_cell.shape[-1]
    if engine == "numpy":
        out = np.zeros(get_number_of_lattice_parameters(d))
        out[..., :d] = np.diag(unit_cell)
        return out
    out = torch.zeros(*unit_cell.shape[:-2], get_number_of_lattice_parameters(d)).to(unit_cell)
    out[..., :d] = torch.diagonal(unit_cell, dim1=-2, dim2=-1)
    return out


def map_numpy_unit_cell_to_lattice_parameters(unit_cell: np.ndarray) -> np.ndarray:
    """:225-227"""
    return map_unit_cell_to_lattice_parameters(unit_cell, engine="numpy")


def map_noisy_axl_lattice_parameters_to_unit_cell_vectors(lattice_parameters: torch.Tensor, min_box_size: float = 4.0) -> torch.Tensor:
    """Cell vectors of NOISY lattice parameters [batch, d (d + 1) / 2]: lengths clipped to min_box_size, angles zeroed -- in
    place on the clipped copy, as the reference (:230-256)."""
    d = get_spatial_dimension_from_number_of_lattice_parameters(lattice_parameters.shape[-1])
    lattice_parameters = lattice_parameters.clip(min=min_box_size)
    lattice_parameters[:, d:] = 0.0
    return map_lattice_parameters_to_unit_cell_vectors(lattice_parameters)


def map_relative_coordinates_to_unit_cell(relative_coordinates: torch.Tensor) -> torch.Tensor:
    """remainder(x, 1) with the reference's fix-up 1.0 -> 0.0 (:95-119), tensor of arbitrary shape -> same shape in [0, 1).

    Runs on the device through the F1 kernel (mdx_noise_relative_coordinates: wrap(x0 + sigma z) with sigma = 0 and
    z = x0, i.e. wrap(x + 0) -- the same wrap the predictor / corrector updates apply); device tensors only."""
    from .. import kernels
    x = relative_coordinates.to(torch.float32).contiguous()
    return kernels.noise_relative_coordinates(x, x, 0.0).reshape(relative_coordinates.shape)


def map_axl_composition_to_unit_cell(composition, device: torch.device):
    """AXL with X wrapped into the unit cell, on `device` (:122-138)."""
    from ..namespace import AXL
    return AXL(A=composition.A.to(device), X=map_relative_coordinates_to_unit_cell(composition.X.to(device)),
               L=composition.L.to(device))
