"""Force-field augmented score network (src/.../models/score_networks/force_field_augmented_score_network.py:20-236).

Wraps any ScoreNetwork and adds a short-range repulsive pseudo-force to the coordinate score.  Second consumer of
the HIP radius graph (kernel N1, full mode: every (src, dst, image) edge with its lattice shift); the edges come out
sorted by source atom, so the per-atom sum is a segment reduction without atomics.
"""
from dataclasses import dataclass
from typing import AnyStr, Dict, Optional

import torch

from ...namespace import AXL, NOISY_AXL_COMPOSITION
from ...utils.neighbors import get_periodic_adjacency_information
from .score_network import ScoreNetwork


@dataclass(kw_only=True)
class ForceFieldParameters:
    """phi(r) = strength * (r - radial_cutoff)^2 for r < radial_cutoff  (:20-41)."""

    radial_cutoff: float
    strength: float

    def __post_init__(self):
        assert self.radial_cutoff > 0.0, "the radial cutoff should be greater than zero."
        assert self.strength > 0.0, "the repulsive strength should be greater than zero."


class ForceFieldAugmentedScoreNetwork(torch.nn.Module):
    def __init__(self, score_network: ScoreNetwork, force_field_parameters: ForceFieldParameters):
        super().__init__()
        self._score_network = score_network
        self._force_field_parameters = force_field_parameters

    # what the generators look for on a score network, passed through to the wrapped one (build-only attributes: the status
    # word of its HIP kernels and the arithmetic of its fused MFMA kernels)
    @property
    def graph_status(self):
        return getattr(self._score_network, "graph_status", None)

    @property
    def edge_chain_precision(self):
        return getattr(self._score_network, "edge_chain_precision", None)

    @edge_chain_precision.setter
    def edge_chain_precision(self, value):
        if hasattr(self._score_network, "edge_chain_precision"):
            self._score_network.edge_chain_precision = value

    def capture_safe(self, batch_size: int, number_of_atoms: int, device) -> bool:
        """The pseudo-force needs the FULL periodic adjacency (with shifts), whose size is read on the host per forward: an
        iteration around this wrapper is never captured into a hipGraph (LangevinGenerator then launches it eagerly)."""
        return False

    def adapt_f16_range(self):
        adapt = getattr(self._score_network, "adapt_f16_range", None)
        if adapt is not None:
            adapt()

    def begin_f16_range_fallback(self):
        begin = getattr(self._score_network, "begin_f16_range_fallback", None)
        if begin is not None:
            begin()

    def reset_f16_range(self):
        reset = getattr(self._score_network, "reset_f16_range", None)
        if reset is not None:
            reset()

    def forward(self, batch: Dict[AnyStr, torch.Tensor], conditional: Optional[bool] = None) -> AXL:
        raw = self._score_network(batch, conditional)
        return AXL(A=raw.A, X=raw.X + self.get_relative_coordinates_pseudo_force(batch), L=raw.L)

    def get_relative_coordinates_pseudo_force(self, batch: Dict[AnyStr, torch.Tensor]) -> torch.Tensor:
        """F_i = sum_j 2 s (r_ij - r0)/r_ij * (p_j + shift - p_i), converted to relative coordinates (:86-236)."""
        comp = batch[NOISY_AXL_COMPOSITION]
        x = comp.X
        bsz, n, d = x.shape
        s, r0 = self._force_field_parameters.strength, self._force_field_parameters.radial_cutoff
        lengths = comp.L[:, :d].clip(min=1.0)                       # min_box_size = 1.0 (:147-150)
        cell = torch.diag_embed(lengths)
        cart = torch.matmul(x, cell)
        info = get_periodic_adjacency_information(cart, cell, radial_cutoff=r0)      # d = 3 only, as the reference (:131-135)
        src, dst = info.adjacency_matrix
        node = info.edge_batch_indices * n + src                    # sorted: edges are grouped by source atom
        flat = cart.reshape(bsz * n, d)
        disp = flat.index_select(0, info.edge_batch_indices * n + dst) - flat.index_select(0, node) + info.shifts
        r = torch.linalg.norm(disp, dim=1)
        contrib = (2.0 * s * (r - r0) / (r + 1.0e-8)).unsqueeze(1) * disp
        degree = torch.bincount(node, minlength=bsz * n)
        forces = torch.segment_reduce(contrib, "sum", lengths=degree, axis=0, unsafe=True).reshape(bsz, n, d)
        return forces / lengths.unsqueeze(1)                        # cartesian -> relative for a diagonal cell
