"""Score-network plugin API (src/.../models/score_networks/score_network.py:27-242), kept so that any
reference-style network (or a checkpoint's axl_network) plugs into the generators.

The forward stays PyTorch.  Differences from the reference: the input checks that force a device->host sync on
every call (`.all()` x3, :110-114,129-131,161-165) run only when `check_inputs=True` (default False on the hot
path; the generators guarantee the invariants by construction).
"""
from dataclasses import dataclass
from typing import AnyStr, Dict, Optional

import torch

from ...namespace import AXL, CARTESIAN_FORCES, NOISE, NOISY_AXL_COMPOSITION, TIME
from ...utils.basis_transformations import get_number_of_lattice_parameters


@dataclass(kw_only=True)
class ScoreNetworkParameters:
    """Base hyper-parameters (:27-45)."""

    architecture: str
    spatial_dimension: int = 3
    num_atom_types: int
    conditional_prob: float = 0.0
    conditional_gamma: float = 2.0

    def __post_init__(self):
        self.num_lattice_parameters = get_number_of_lattice_parameters(self.spatial_dimension)


class ScoreNetwork(torch.nn.Module):
    """Base class: forward(batch, conditional) -> AXL(A logits [B,N,C], X sigma-normalised score, L)."""

    check_inputs = False

    def __init__(self, hyper_params: ScoreNetworkParameters):
        super().__init__()
        self._hyper_params = hyper_params
        self.spatial_dimension = hyper_params.spatial_dimension
        self.num_atom_types = hyper_params.num_atom_types
        self.conditional_prob = hyper_params.conditional_prob
        self.conditional_gamma = hyper_params.conditional_gamma

    def _check_batch(self, batch: Dict[AnyStr, torch.Tensor]):
        """Shape checks always; value checks (which synchronise) only when check_inputs is set (:68-181)."""
        assert NOISY_AXL_COMPOSITION in batch, f"'{NOISY_AXL_COMPOSITION}' missing from the batch"
        x = batch[NOISY_AXL_COMPOSITION].X
        bs = x.shape[0]
        assert x.dim() == 3 and x.shape[2] == self.spatial_dimension, \
            "relative coordinates must be [batch_size, number_of_atoms, spatial_dimension]"
        assert TIME in batch and NOISE in batch, "time and noise parameter must be in the batch"
        t = batch[TIME]
        assert t.shape == (bs, 1) and batch[NOISE].shape == t.shape, "time / noise must be [batch_size, 1]"
        lat = batch[NOISY_AXL_COMPOSITION].L
        assert lat.shape == (bs, get_number_of_lattice_parameters(self.spatial_dimension)), \
            "lattice parameters must be [batch_size, d(d+1)/2]"
        a = batch[NOISY_AXL_COMPOSITION].A
        assert a.dim() == 2 and a.shape[0] == bs, "atom types must be [batch_size, number_of_atoms]"
        if self.conditional_prob > 0:
            assert CARTESIAN_FORCES in batch and batch[CARTESIAN_FORCES].shape == x.shape
        if self.check_inputs:
            assert torch.logical_and(x >= 0.0, x < 1.0).all(), "relative coordinates must be in [0,1)"
            assert torch.logical_and(t >= 0.0, t <= 1.0).all(), "times must be in [0,1]"
            assert torch.logical_and(a >= 0, a < self.num_atom_types + 1).all(), "atom types out of range"

    def _impose_non_mask_atomic_type_prediction(self, output: AXL):
        """The MASK logit is forced to -inf (:183-185)."""
        output.A[..., self.num_atom_types] = -torch.inf

    def forward(self, batch: Dict[AnyStr, torch.Tensor], conditional: Optional[bool] = None) -> AXL:
        self._check_batch(batch)
        if conditional is None:
            conditional = bool(torch.rand(1) < self.conditional_prob)
        if not conditional:
            output = self._forward_unchecked(batch, conditional=False)
        else:
            c = self._forward_unchecked(batch, conditional=True)
            u = self._forward_unchecked(batch, conditional=False)
            g = self.conditional_gamma
            output = AXL(A=c.A * g + u.A * (1 - g), X=c.X * g + u.X * (1 - g), L=c.L * g + u.L * (1 - g))
        self._impose_non_mask_atomic_type_prediction(output)
        return output

    def _forward_unchecked(self, batch: Dict[AnyStr, torch.Tensor], conditional: bool = False) -> AXL:
        raise NotImplementedError
