"""Periodic-wrapped Gaussian noising (HIP kernel F1) -- src/.../noisers/relative_coordinates_noiser.py:33-67."""
from typing import Tuple

import torch

from .. import kernels


class RelativeCoordinatesNoiser:
    @staticmethod
    def _get_gaussian_noise(shape: Tuple[int]) -> torch.Tensor:
        return torch.randn(shape)     # CPU generator, like the reference

    @staticmethod
    def get_noisy_relative_coordinates_sample(real_relative_coordinates: torch.Tensor, sigma: float) -> torch.Tensor:
        """x_t = wrap(x_0 + sigma z).  `sigma` is one scalar for the whole call (the sampler noises a batch to a
        single time index); the reference's per-element sigma tensor is constant in that use."""
        z = RelativeCoordinatesNoiser._get_gaussian_noise(real_relative_coordinates.shape).to(real_relative_coordinates)
        return kernels.noise_relative_coordinates(real_relative_coordinates.contiguous(), z.contiguous(), sigma)
