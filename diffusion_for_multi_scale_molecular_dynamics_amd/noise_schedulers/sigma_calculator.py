"""sigma(t) and its time derivative as modules (src/.../noise_schedulers/sigma_calculator.py): the reference's names for callers
that evaluate the schedule at arbitrary times.  The sampler does not come through here: its tables are one launch of
schedule_kernel (csrc/mdx_hip.hip), whose sigma column follows the same two laws in the same binary32 operations."""
import torch
from torch import nn


class SigmaCalculator(nn.Module):
    """Base: holds sigma_min / sigma_max as frozen parameters; `forward(times)` = `get_sigma(times)` (:16-45)."""

    def __init__(self, sigma_min: float, sigma_max: float):
        super().__init__()
        self.sigma_min = nn.Parameter(torch.tensor(sigma_min), requires_grad=False)
        self.sigma_max = nn.Parameter(torch.tensor(sigma_max), requires_grad=False)

    def get_sigma(self, times: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("This method must be implemented in a child class.")

    def get_sigma_time_derivative(self, times: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("This method must be implemented in a child class.")

    def forward(self, times: torch.Tensor) -> torch.Tensor:
        return self.get_sigma(times)


class ExponentialSigmaCalculator(SigmaCalculator):
    """sigma_min (sigma_max / sigma_min)^t  (:48-78)."""

    def __init__(self, sigma_min: float, sigma_max: float):
        super().__init__(sigma_min, sigma_max)
        ratio = self.sigma_max / self.sigma_min
        self.ratio = nn.Parameter(ratio, requires_grad=False)
        self.log_ratio = nn.Parameter(torch.log(ratio), requires_grad=False)

    def get_sigma(self, times: torch.Tensor) -> torch.Tensor:
        return self.sigma_min * self.ratio ** times

    def get_sigma_time_derivative(self, times: torch.Tensor) -> torch.Tensor:
        return self.log_ratio * self.get_sigma(times)


class LinearSigmaCalculator(SigmaCalculator):
    """sigma_min + (sigma_max - sigma_min) t  (:81-108)."""

    def __init__(self, sigma_min: float, sigma_max: float):
        super().__init__(sigma_min, sigma_max)
        self.sigma_difference = nn.Parameter(self.sigma_max - self.sigma_min, requires_grad=False)

    def get_sigma(self, times: torch.Tensor) -> torch.Tensor:
        return self.sigma_min + self.sigma_difference * times

    def get_sigma_time_derivative(self, times: torch.Tensor) -> torch.Tensor:
        return self.sigma_difference * torch.ones_like(times)


def instantiate_sigma_calculator(sigma_min: float, sigma_max: float, schedule_type: str) -> SigmaCalculator:
    calculators = dict(exponential=ExponentialSigmaCalculator, linear=LinearSigmaCalculator)
    if schedule_type not in calculators:
        raise NotImplementedError(f"The schedule type {schedule_type} is not implemented")
    return calculators[schedule_type](sigma_min, sigma_max)
