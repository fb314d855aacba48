"""The variance-exploding schedule as a function of continuous time (src/.../noise_schedulers/exploding_variance.py):
sigma(t), its derivative and g^2(t) = d sigma^2 / dt for callers that evaluate them at arbitrary times (the reference's ODE / SDE
generators, its tests).  The predictor-corrector sampler reads the DISCRETE tables of NoiseScheduler, built on the device by
schedule_kernel from the same two laws."""
import torch

from .noise_parameters import NoiseParameters
from .sigma_calculator import instantiate_sigma_calculator


class VarianceScheduler(torch.nn.Module):
    def __init__(self, noise_parameters: NoiseParameters):
        super().__init__()
        p = noise_parameters
        self.sigma_calculator = instantiate_sigma_calculator(p.sigma_min, p.sigma_max, p.schedule_type)

    def get_sigma(self, times: torch.Tensor) -> torch.Tensor:
        return self.sigma_calculator.get_sigma(times)

    def get_sigma_time_derivative(self, times: torch.Tensor) -> torch.Tensor:
        return self.sigma_calculator.get_sigma_time_derivative(times)

    def get_g_squared(self, times: torch.Tensor) -> torch.Tensor:
        """2 sigma sigma' (:53-64)."""
        return 2.0 * self.get_sigma(times) * self.get_sigma_time_derivative(times)
