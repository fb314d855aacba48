"""Gaussian noising of the lattice parameters (kernel P3 with a zero score) -- src/.../noisers/lattice_noiser.py:7-81."""
from dataclasses import dataclass
from typing import Tuple

import torch

from .. import kernels


@dataclass(kw_only=True)
class LatticeDataParameters:
    spatial_dimension: int = 3
    use_fixed_lattice_parameters: bool = False


class LatticeNoiser:
    def __init__(self, lattice_parameters: LatticeDataParameters):
        self.spatial_dimension = lattice_parameters.spatial_dimension
        self.use_fixed_lattice_parameters = lattice_parameters.use_fixed_lattice_parameters

    @staticmethod
    def _get_gaussian_noise(shape: Tuple[int]) -> torch.Tensor:
        return torch.randn(shape)     # CPU generator, like the reference

    def get_noisy_lattice_parameters(self, real_lattice_parameters: torch.Tensor, sigma_n: float) -> torch.Tensor:
        """l_t = l_0 + sigma_n z; identity (and no draw) when the lattice is fixed (:69-72).  `sigma_n` is one scalar
        for the call, as in the sampler's use (a batch noised to a single time index)."""
        if self.use_fixed_lattice_parameters:
            return real_lattice_parameters
        z = self._get_gaussian_noise(real_lattice_parameters.shape).to(real_lattice_parameters).contiguous()
        zero = torch.zeros_like(real_lattice_parameters)
        # (l + (0 * 0) / 1) + sigma_n * z  ==  sigma_n * z + l  bit for bit
        return kernels.lattice_parameters_update(real_lattice_parameters.contiguous(), zero, z, 0.0, sigma_n, 1.0)
