"""Repaint constraints and their pickle format (src/.../generators/sampling_constraint.py:10-97)."""
import dataclasses
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional

import torch


@dataclass
class SamplingConstraint:
    elements: List[str]
    constrained_relative_coordinates: torch.Tensor   # [K, d] float32
    constrained_atom_types: torch.Tensor             # [K] int64, indices into `elements`
    constrained_indices: Optional[torch.Tensor] = None   # [K] int64; default arange(K)

    def __post_init__(self):
        x, a, idx = self.constrained_relative_coordinates, self.constrained_atom_types, self.constrained_indices
        assert type(x) is torch.Tensor, "the constrained_relative_coordinates should be a torch Tensor."
        assert x.dtype is torch.float, "the constrained_relative_coordinates should be composed of floats."
        assert len(x.shape) == 2, "constrained_relative_coordinates has the wrong shape."
        assert type(a) is torch.Tensor, "the constrained_atom_types should be a torch Tensor."
        assert a.dtype is torch.long, "the constrained_atom_types should be composed of long integers."
        assert len(a.shape) == 1, "constrained_atom_types has the wrong shape."
        assert x.shape[0] == a.shape[0], "The number of constrained atoms should match"
        assert torch.logical_and(a >= 0, a < len(self.elements)).all(), \
            "There is a mismatch between the specified elements and the constrained atom types."
        if idx is not None:
            assert type(idx) is torch.Tensor, "the constrained_indices should be a torch Tensor or None."
            assert len(idx.shape) == 1, "constrained_indices has the wrong shape."
            assert idx.dtype is torch.long, "the constrained_indices, if specified, should be composed of long integers."
            assert x.shape[0] == idx.shape[0], "The number of constrained atoms should match"


def write_sampling_constraint(sampling_constraint: SamplingConstraint, output_path: Path):
    torch.save(dataclasses.asdict(sampling_constraint), output_path)


def read_sampling_constraint(output_path: Path) -> SamplingConstraint:
    return SamplingConstraint(**torch.load(output_path, weights_only=False))
