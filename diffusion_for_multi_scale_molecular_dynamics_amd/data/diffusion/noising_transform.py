"""Forward noising of a batch to ONE time index, as the repaint generator uses it
(src/.../data/diffusion/noising_transform.py:30-200, method transform_given_time_index; the random-index training
transform and optimal transport are outside the sampling hot path).

Stand-alone form of what mdx_repaint_constrained_rows fuses: kernels F1 (wrapped Gaussian on X), F2 (D3PM on A),
F3 (Gaussian on L).  Draw order = the reference's: X noise, then A noise, then L noise.
"""
from typing import Dict

import torch

from ...namespace import (ATOM_TYPES, LATTICE_PARAMETERS, NOISE, NOISY_ATOM_TYPES, NOISY_LATTICE_PARAMETERS,
                          NOISY_RELATIVE_COORDINATES, Q_BAR_MATRICES, Q_BAR_TM1_MATRICES, Q_MATRICES,
                          RELATIVE_COORDINATES, TIME, TIME_INDICES)
from ...noise_schedulers.noise_parameters import NoiseParameters
from ...noise_schedulers.noise_scheduler import NoiseScheduler
from ...noisers.atom_types_noiser import AtomTypesNoiser
from ...noisers.lattice_noiser import LatticeDataParameters, LatticeNoiser
from ...noisers.relative_coordinates_noiser import RelativeCoordinatesNoiser


class NoisingTransform:
    def __init__(self, noise_parameters: NoiseParameters, num_atom_types: int, spatial_dimension: int,
                 use_fixed_lattice_parameters: bool = False, use_optimal_transport: bool = False, device="cuda"):
        assert not use_optimal_transport, "optimal transport is a training-time augmentation (out of scope)"
        self.num_atom_types = num_atom_types
        self.noise_scheduler = NoiseScheduler(noise_parameters, num_classes=num_atom_types + 1, device=device)
        self.lattice_noiser = LatticeNoiser(LatticeDataParameters(
            spatial_dimension=spatial_dimension, use_fixed_lattice_parameters=use_fixed_lattice_parameters))

    def transform_given_time_index(self, batch: Dict, index_i: int) -> Dict:
        """index_i is the one-based time index (t_1 = delta, ..., t_T = 1)  (:98-120)."""
        assert index_i > 0, "The time index should never be smaller than 1."
        idx = index_i - 1
        for key in (RELATIVE_COORDINATES, ATOM_TYPES, LATTICE_PARAMETERS):
            assert key in batch, f"The field '{key}' is missing from the input."
        x0, a0, l0 = batch[RELATIVE_COORDINATES], batch[ATOM_TYPES], batch[LATTICE_PARAMETERS]
        assert x0.dim() == 3 and a0.dim() == 2 and l0.dim() == 2
        t = self.noise_scheduler.tables
        bsz, natoms, d = x0.shape
        sigma = float(t.sigma[idx])
        out = dict(batch)
        out[TIME] = t.time[idx].expand(bsz).reshape(-1, 1)
        out[NOISE] = t.sigma[idx].expand(bsz).reshape(-1, 1)
        out[TIME_INDICES] = torch.full((bsz,), idx, dtype=torch.long, device=x0.device)
        out[Q_MATRICES] = t.q_matrix[idx].expand(bsz, natoms, -1, -1)
        out[Q_BAR_MATRICES] = t.q_bar_matrix[idx].expand(bsz, natoms, -1, -1)
        out[Q_BAR_TM1_MATRICES] = t.q_bar_tm1_matrix[idx].expand(bsz, natoms, -1, -1)
        out[NOISY_RELATIVE_COORDINATES] = RelativeCoordinatesNoiser.get_noisy_relative_coordinates_sample(x0, sigma)
        out[NOISY_ATOM_TYPES] = AtomTypesNoiser.get_noisy_atom_types_sample(a0, t.q_bar_matrix[idx])
        sigma_n = float(t.sigma[idx] / torch.tensor(float(natoms)) ** (1 / d))
        out[NOISY_LATTICE_PARAMETERS] = self.lattice_noiser.get_noisy_lattice_parameters(l0, sigma_n)
        return out
