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
                 use_fixed_lattice_parameters: bool = False, use_optimal_transport: bool = True, device="cuda"):
        """The reference's signature and defaults (:37-44) + `device`.  Optimal transport (the default there) re-assigns atoms
        while noising a TRAINING batch: outside the sampling hot path, refused loudly rather than silently skipped -- the
        sampling path passes use_optimal_transport=False (generators/constrained_langevin_generator.py:71)."""
        if use_optimal_transport:
            raise NotImplementedError("NoisingTransform(use_optimal_transport=True) is the training-time augmentation, outside "
                                      "this package's scope: pass use_optimal_transport=False (as the repaint generator does)")
        self.num_atom_types = num_atom_types
        self.noise_scheduler = NoiseScheduler(noise_parameters, num_classes=num_atom_types + 1, device=device)
        self.lattice_noiser = LatticeNoiser(LatticeDataParameters(
            spatial_dimension=spatial_dimension, use_fixed_lattice_parameters=use_fixed_lattice_parameters))

    def _check_batch(self, batch: Dict):
        """The fields a batch must hold, with their ranks (:46-60)."""
        for key in (RELATIVE_COORDINATES, ATOM_TYPES, LATTICE_PARAMETERS):
            assert key in batch, f"The field '{key}' is missing from the input."
        assert batch[RELATIVE_COORDINATES].dim() == 3 and batch[ATOM_TYPES].dim() == 2 and batch[LATTICE_PARAMETERS].dim() == 2

    def transform(self, batch: Dict) -> Dict:
        """The training transform (:62-96: a random time index per structure)."""
        raise NotImplementedError("NoisingTransform.transform noises a training batch at random time indices: outside the "
                                  "sampling hot path; the sampler's entry point is transform_given_time_index")

    def transform_given_time_index(self, batch: Dict, index_i: int) -> Dict:
        """index_i is the one-based time index (t_1 = delta, ..., t_T = 1)  (:98-120)."""
        assert index_i > 0, "The time index should never be smaller than 1."
        idx = index_i - 1
        self._check_batch(batch)
        x0, a0, l0 = batch[RELATIVE_COORDINATES], batch[ATOM_TYPES], batch[LATTICE_PARAMETERS]
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
