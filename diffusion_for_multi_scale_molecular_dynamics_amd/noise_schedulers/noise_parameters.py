"""NoiseParameters (src/.../noise_schedulers/noise_parameters.py:5-36): the `noise:` block of the YAML surface."""
from dataclasses import dataclass


@dataclass
class NoiseParameters:
    """Variance-exploding schedule parameters."""

    total_time_steps: int
    schedule_type: str = "exponential"   # or "linear"
    time_delta: float = 1e-5             # times cover [time_delta, 1]
    sigma_min: float = 0.005
    sigma_max: float = 0.5
    corrector_step_epsilon: float = 2e-5
    corrector_r: float = 0.17            # adaptive corrector only

    def __post_init__(self):
        assert self.schedule_type in ["exponential", "linear"], \
            f"The schedule type {self.schedule_type} is not supported."
