"""What the repaint generator pins: K known atoms (reference: generators/sampling_constraint.py:10-97).

Same dataclass fields and the same pickle layout (a plain dict of the fields) as the reference, so constraint files
are interchangeable.
"""
import dataclasses
from pathlib import Path
from typing import List, Optional

import torch


def _check_tensor(name: str, value, dtype: torch.dtype, rank: int):
    assert type(value) is torch.Tensor, f"{name} should be a torch Tensor."
    assert value.dtype is dtype, f"{name} should have dtype {dtype}."
    assert value.dim() == rank, f"{name} should have {rank} dimension(s); got shape {tuple(value.shape)}."


@dataclasses.dataclass
class SamplingConstraint:
    elements: List[str]                                   # element names; atom types index into this list
    constrained_relative_coordinates: torch.Tensor        # float32 [K, d]
    constrained_atom_types: torch.Tensor                  # int64 [K]
    constrained_indices: Optional[torch.Tensor] = None    # int64 [K] rows of the structure to pin; None = the first K

    def __post_init__(self):
        _check_tensor("constrained_relative_coordinates", self.constrained_relative_coordinates, torch.float, 2)
        _check_tensor("constrained_atom_types", self.constrained_atom_types, torch.long, 1)
        count = self.constrained_relative_coordinates.shape[0]
        assert self.constrained_atom_types.shape[0] == count, "one atom type per constrained position is required."
        in_range = (self.constrained_atom_types >= 0) & (self.constrained_atom_types < len(self.elements))
        assert bool(in_range.all()), "constrained atom types must index into `elements`."
        if self.constrained_indices is not None:
            _check_tensor("constrained_indices", self.constrained_indices, torch.long, 1)
            assert self.constrained_indices.shape[0] == count, "one row index per constrained position is required."


def write_sampling_constraint(sampling_constraint: SamplingConstraint, output_path: Path):
    torch.save(dataclasses.asdict(sampling_constraint), output_path)


def read_sampling_constraint(output_path: Path) -> SamplingConstraint:
    from ..utils import reference_pickles
    fields = reference_pickles.load(output_path)
    return SamplingConstraint(**fields)
