"""Periodic-wrapped Gaussian noising (HIP kernel F1) -- src/.../noisers/relative_coordinates_noiser.py:33-67."""
from typing import Tuple, Union

import torch

from .. import kernels


class RelativeCoordinatesNoiser:
    @staticmethod
    def _get_gaussian_noise(shape: Tuple[int]) -> torch.Tensor:
        return torch.randn(shape)     # CPU generator, like the reference

    @staticmethod
    def get_noisy_relative_coordinates_sample(real_relative_coordinates: torch.Tensor,
                                              sigmas: Union[torch.Tensor, float]) -> torch.Tensor:
        """x_t = wrap(x_0 + sigmas z).  `sigmas`: a tensor of the shape of real_relative_coordinates, as in the reference (:52-55
        asserts the same), or one number for the whole call (the sampler noises a batch to a single time index)."""
        x0 = real_relative_coordinates
        z = RelativeCoordinatesNoiser._get_gaussian_noise(x0.shape)
        if isinstance(sigmas, torch.Tensor):
            assert x0.shape == sigmas.shape, \
                "sigmas array is expected to be of the same shape as the real_relative_coordinates array"
            sigmas = sigmas.to(device=x0.device, dtype=torch.float32).contiguous()
        return kernels.noise_relative_coordinates(x0.contiguous(), z.to(x0).contiguous(), sigmas)
