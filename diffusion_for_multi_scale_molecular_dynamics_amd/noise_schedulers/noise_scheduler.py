"""Variance-exploding noise scheduler with device-resident tables (HIP kernel S1).

API of src/.../noise_schedulers/noise_scheduler.py:30-378: the tables (time, sigma, sigma^2, g, g^2, beta,
alpha_bar, Q, Qbar, Qbar_{t-1}; epsilon, sqrt(2 epsilon)) are built ONCE by mdx_noise_schedule_build directly in
HBM and stay there; the per-step kernels index them on the device, so no scalar crosses PCIe during sampling.
"""
from collections import namedtuple
from typing import Tuple

import torch

from .. import kernels
from .noise_parameters import NoiseParameters

Noise = namedtuple("Noise", ["time", "sigma", "sigma_squared", "g", "g_squared", "beta", "alpha_bar", "q_matrix",
                             "q_bar_matrix", "q_bar_tm1_matrix", "indices"])
LangevinDynamics = namedtuple("LangevinDynamics", ["epsilon", "sqrt_2_epsilon"])


class NoiseScheduler:
    """Index conventions are the reference's (noise_scheduler.py:91-109): arrays are indexed by idx = i - 1 for
    time index i = 1..T, except epsilon / sqrt_2_epsilon which are indexed by i = 0..T-1."""

    def __init__(self, noise_parameters: NoiseParameters, num_classes: int, device="cuda"):
        self.noise_parameters = noise_parameters
        self.num_classes = num_classes
        p = noise_parameters
        self.tables = kernels.noise_schedule_build(p.total_time_steps, p.schedule_type, p.time_delta, p.sigma_min,
                                                   p.sigma_max, p.corrector_step_epsilon, num_classes, device)

    def get_all_sampling_parameters(self) -> Tuple[Noise, LangevinDynamics]:
        t = self.tables
        noise = Noise(time=t.time, sigma=t.sigma, sigma_squared=t.sigma_squared, g=t.g, g_squared=t.g_squared,
                      beta=t.beta, alpha_bar=t.alpha_bar, q_matrix=t.q_matrix, q_bar_matrix=t.q_bar_matrix,
                      q_bar_tm1_matrix=t.q_bar_tm1_matrix,
                      indices=torch.arange(0, self.noise_parameters.total_time_steps, device=t.device))
        return noise, LangevinDynamics(epsilon=t.epsilon, sqrt_2_epsilon=t.sqrt_2_epsilon)

    def get_random_noise_sample(self, batch_size: int) -> Noise:
        """One uniformly drawn time index (0 .. T - 1) per structure and the schedule's rows there (:289-308; the training side's
        entry point -- the sampler walks the indices in order)."""
        t = self.tables
        indices = torch.randint(0, self.noise_parameters.total_time_steps, size=(batch_size,), device=t.device)
        return self.get_noise_from_indices(indices)

    def get_noise_from_indices(self, indices: torch.Tensor) -> Noise:
        """noise_scheduler.py:310-346"""
        t = self.tables
        return Noise(time=t.time.take(indices), sigma=t.sigma.take(indices),
                     sigma_squared=t.sigma_squared.take(indices), g=t.g.take(indices),
                     g_squared=t.g_squared.take(indices), beta=t.beta.take(indices),
                     alpha_bar=t.alpha_bar.take(indices), q_matrix=t.q_matrix.index_select(0, indices),
                     q_bar_matrix=t.q_bar_matrix.index_select(0, indices),
                     q_bar_tm1_matrix=t.q_bar_tm1_matrix.index_select(0, indices), indices=indices)
