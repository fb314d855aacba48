"""Gaussian noising of the lattice parameters (HIP kernel F3) -- src/.../noisers/lattice_noiser.py:7-81."""
from dataclasses import dataclass
from typing import Tuple, Union

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

    def get_noisy_lattice_parameters(self, real_lattice_parameters: torch.Tensor,
                                     sigmas_n: Union[torch.Tensor, float]) -> torch.Tensor:
        """l_t = sigmas_n z + l_0; identity (and no draw) when the lattice is fixed (:69-72).  `sigmas_n`: a tensor of the shape
        of real_lattice_parameters, as in the reference (:64-66 asserts the same), or one number for the whole call."""
        l0 = real_lattice_parameters
        if isinstance(sigmas_n, torch.Tensor):
            assert l0.shape == sigmas_n.shape, \
                "sigmas array is expected to be of the same shape as the real_lattice_parameters array"
        if self.use_fixed_lattice_parameters:
            return l0
        z = self._get_gaussian_noise(l0.shape).to(l0).contiguous()
        if isinstance(sigmas_n, torch.Tensor):
            return kernels.noise_lattice_parameters(l0.contiguous(), z,
                                                    sigmas_n.to(device=l0.device, dtype=torch.float32).contiguous())
        zero = torch.zeros_like(l0)
        # (l + (0 * 0) / 1) + sigma_n * z  ==  sigma_n * z + l  bit for bit
        return kernels.lattice_parameters_update(l0.contiguous(), zero, z, 0.0, sigmas_n, 1.0)
