"""`diffusion_sampling:` block of training configs (src/.../sampling/diffusion_sampling_parameters.py:15-70);
the metrics sub-block is kept as a plain dict (metrics are outside the sampling hot path)."""
from dataclasses import dataclass
from typing import Any, AnyStr, Dict, Union

from ..generators.axl_generator import SamplingParameters
from ..generators.load_sampling_parameters import load_sampling_parameters
from ..noise_schedulers.noise_parameters import NoiseParameters


@dataclass(kw_only=True)
class DiffusionSamplingParameters:
    sampling_parameters: SamplingParameters
    noise_parameters: NoiseParameters
    metrics_parameters: Dict[str, Any]


def load_diffusion_sampling_parameters(hyper_params: Dict[AnyStr, Any]) -> Union[DiffusionSamplingParameters, None]:
    if "diffusion_sampling" not in hyper_params:
        return None
    block = hyper_params["diffusion_sampling"]
    assert "sampling" in block, "The sampling parameters must be defined to draw samples."
    assert "noise" in block, "The noise parameters must be defined to draw samples."
    assert "metrics" in block, "The metrics parameters must be defined to draw samples."
    return DiffusionSamplingParameters(sampling_parameters=load_sampling_parameters(block["sampling"]),
                                       noise_parameters=NoiseParameters(**block["noise"]),
                                       metrics_parameters=dict(block["metrics"]))
