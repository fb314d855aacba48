"""The `diffusion_sampling:` block of a training configuration (reference: sampling/diffusion_sampling_parameters.py:15-70).

Three sub-blocks are required, as in the reference: `sampling`, `noise` and `metrics`.  The first two become the
objects the generators take; `metrics` stays a plain dict (the metrics themselves are outside the sampling hot path).
"""
import dataclasses
from typing import Any, AnyStr, Dict, Union

from ..generators.axl_generator import SamplingParameters
from ..generators.load_sampling_parameters import load_sampling_parameters
from ..noise_schedulers.noise_parameters import NoiseParameters

_REQUIRED_BLOCKS = ("sampling", "noise", "metrics")


@dataclasses.dataclass(kw_only=True)
class DiffusionSamplingParameters:
    sampling_parameters: SamplingParameters
    noise_parameters: NoiseParameters
    metrics_parameters: Dict[str, Any]


def load_diffusion_sampling_parameters(hyper_params: Dict[AnyStr, Any]) -> Union[DiffusionSamplingParameters, None]:
    """None when the configuration has no `diffusion_sampling:` block."""
    block = hyper_params.get("diffusion_sampling")
    if block is None:
        return None
    for name in _REQUIRED_BLOCKS:
        assert name in block, f"diffusion_sampling needs a '{name}' block to draw samples."
    return DiffusionSamplingParameters(sampling_parameters=load_sampling_parameters(block["sampling"]),
                                       noise_parameters=NoiseParameters(**block["noise"]),
                                       metrics_parameters=dict(block["metrics"]))
