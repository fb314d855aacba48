"""Generator plugin API (src/.../generators/axl_generator.py:14-95): SamplingParameters + AXLGenerator ABC."""
import warnings
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import List, Optional

import torch

from ..namespace import AXL
from ..utils.basis_transformations import map_unit_cell_to_lattice_parameters


@dataclass(kw_only=True)
class SamplingParameters:
    """The `sampling:` block of the YAML surface (axl_generator.py:14-37)."""

    algorithm: str
    spatial_dimension: int = 3
    num_atom_types: int
    number_of_atoms: int
    number_of_samples: int
    sample_batchsize: Optional[int] = None
    use_fixed_lattice_parameters: bool = False
    cell_dimensions: Optional[List[float]] = None
    record_samples: bool = False
    record_samples_corrector_steps: bool = False
    record_atom_type_update: bool = False

    def __post_init__(self):
        if self.use_fixed_lattice_parameters:
            assert self.cell_dimensions is not None, \
                "If use_fixed_lattice_parameters is True, then cell_dimensions must be provided."
            cell = torch.tensor(self.cell_dimensions)
            if cell.dim() == 1:
                cell = torch.diag(cell)
            assert cell.dim() == 2, f"Provided cell_dimensions must be a 2D tensor. Got {cell.shape}."
            assert cell.shape[0] == cell.shape[1] == self.spatial_dimension, \
                "The cell_dimensions tensor must have shape [spatial_dimension, spatial_dimension]."
            self.fixed_lattice_parameters = map_unit_cell_to_lattice_parameters(cell)
        else:
            warnings.warn("Using diffusion on lattice parameters. This is experimental and not fully tested.")
            self.fixed_lattice_parameters = None


class AXLGenerator(ABC):
    """Interface of AXL (atom types, relative coordinates, lattice) generators."""

    @abstractmethod
    def sample(self, number_of_samples: int, device: torch.device) -> AXL:
        pass

    @abstractmethod
    def initialize(self, number_of_samples: int, device: torch.device) -> AXL:
        pass
