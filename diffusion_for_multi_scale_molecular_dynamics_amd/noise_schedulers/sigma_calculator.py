"""sigma(t) and its time derivative as modules (src/.../noise_schedulers/sigma_calculator.py): the reference's names for callers
that evaluate the schedule at arbitrary times.  The sampler does not come through here: its tables are one launch of
schedule_kernel (csrc/mdx_hip.hip), whose sigma column follows the same two laws in the same binary32 operations."""
import torch
from torch import nn


def _frozen(value) -> nn.Parameter:
    return nn.Parameter(torch.as_tensor(value), requires_grad=False)


class SigmaCalculator(nn.Module):
    """Holds the two ends of the schedule as frozen parameters (they follow the module across devices and appear in its
    state_dict, as in the reference: :16-45); a law is a subclass that says sigma(t) and d sigma / dt."""

    def __init__(self, sigma_min: float, sigma_max: float):
        super().__init__()
        self.sigma_min, self.sigma_max = _frozen(sigma_min), _frozen(sigma_max)

    def forward(self, times: torch.Tensor) -> torch.Tensor:
        return self.get_sigma(times)

    def get_sigma(self, times: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("This method must be implemented in a child class.")

    def get_sigma_time_derivative(self, times: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("This method must be implemented in a child class.")


class ExponentialSigmaCalculator(SigmaCalculator):
    """sigma_min r^t with r = sigma_max / sigma_min; derivative ln(r) sigma(t)  (:48-78)."""

    def __init__(self, sigma_min: float, sigma_max: float):
        super().__init__(sigma_min, sigma_max)
        self.ratio = _frozen(self.sigma_max / self.sigma_min)
        self.log_ratio = _frozen(torch.log(self.sigma_max / self.sigma_min))

    def get_sigma(self, times):
        return self.sigma_min * self.ratio ** times

    def get_sigma_time_derivative(self, times):
        return self.log_ratio * self.get_sigma(times)


class LinearSigmaCalculator(SigmaCalculator):
    """sigma_min + (sigma_max - sigma_min) t; constant derivative  (:81-108)."""

    def __init__(self, sigma_min: float, sigma_max: float):
        super().__init__(sigma_min, sigma_max)
        self.sigma_difference = _frozen(self.sigma_max - self.sigma_min)

    def get_sigma(self, times):
        return self.sigma_min + self.sigma_difference * times

    def get_sigma_time_derivative(self, times):
        return self.sigma_difference * torch.ones_like(times)


_LAWS = {"exponential": ExponentialSigmaCalculator, "linear": LinearSigmaCalculator}


def instantiate_sigma_calculator(sigma_min: float, sigma_max: float, schedule_type: str) -> SigmaCalculator:
    if schedule_type not in _LAWS:
        raise NotImplementedError(f"The schedule type {schedule_type} is not implemented")
    return _LAWS[schedule_type](sigma_min, sigma_max)
