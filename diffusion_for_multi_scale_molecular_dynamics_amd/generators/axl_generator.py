"""The two ends of the generator plugin API (reference: generators/axl_generator.py:14-95).

SamplingParameters is the common part of the YAML `sampling:` block; AXLGenerator is what `create_batch_of_samples`
drives.  Field names and defaults are the reference's (they are the configuration surface).
"""
import abc
import dataclasses
import warnings
from typing import List, Optional

import torch

from ..namespace import AXL
from ..utils.basis_transformations import map_unit_cell_to_lattice_parameters


def _fixed_lattice(cell_dimensions, spatial_dimension: int) -> torch.Tensor:
    """cell_dimensions (box lengths, or a full d x d cell) -> the d(d+1)/2 lattice parameters held fixed while sampling."""
    assert cell_dimensions is not None, "use_fixed_lattice_parameters=True needs cell_dimensions."
    cell = torch.as_tensor(cell_dimensions, dtype=torch.get_default_dtype())
    cell = torch.diag(cell) if cell.dim() == 1 else cell
    assert cell.dim() == 2 and tuple(cell.shape) == (spatial_dimension, spatial_dimension), \
        f"cell_dimensions should be d lengths or a d x d cell with d = {spatial_dimension}; got shape {tuple(cell.shape)}."
    return map_unit_cell_to_lattice_parameters(cell)


@dataclasses.dataclass(kw_only=True)
class SamplingParameters:
    algorithm: str
    spatial_dimension: int = 3
    num_atom_types: int
    number_of_atoms: int
    number_of_samples: int
    sample_batchsize: Optional[int] = None          # None: one batch
    use_fixed_lattice_parameters: bool = False
    cell_dimensions: Optional[List[float]] = None
    record_samples: bool = False                    # keep the whole trajectory (SampleTrajectory)
    record_samples_corrector_steps: bool = False
    record_atom_type_update: bool = False

    def __post_init__(self):
        self.fixed_lattice_parameters = None
        if self.use_fixed_lattice_parameters:
            self.fixed_lattice_parameters = _fixed_lattice(self.cell_dimensions, self.spatial_dimension)
        else:
            warnings.warn("The lattice parameters are diffused as well: experimental in the reference, not fully tested.")


class AXLGenerator(abc.ABC):
    """Anything that can initialise and sample batches of AXL compositions."""

    @abc.abstractmethod
    def initialize(self, number_of_samples: int, device: torch.device) -> AXL:
        """The composition a trajectory starts from."""

    @abc.abstractmethod
    def sample(self, number_of_samples: int, device: torch.device) -> AXL:
        """number_of_samples finished compositions."""
