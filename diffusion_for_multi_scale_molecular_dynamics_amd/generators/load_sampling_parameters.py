"""YAML dict -> sampling parameters (src/.../generators/load_sampling_parameters.py:13-50)."""
from typing import Any, AnyStr, Dict

from .axl_generator import SamplingParameters
from .predictor_corrector_axl_generator import PredictorCorrectorSamplingParameters

SUPPORTED = ("predictor_corrector", "adaptive_corrector")
KNOWN = ("ode", "sde", "predictor_corrector", "adaptive_corrector")


def load_sampling_parameters(sampling_parameter_dictionary: Dict[AnyStr, Any]) -> SamplingParameters:
    assert "algorithm" in sampling_parameter_dictionary, "The sampling parameters must select an algorithm."
    algorithm = sampling_parameter_dictionary["algorithm"]
    assert algorithm in KNOWN, \
        "Unknown algorithm. Possible choices are 'ode', 'sde', 'predictor_corrector' and 'adaptive_corrector'"
    if algorithm not in SUPPORTED:
        raise NotImplementedError(
            f"algorithm '{algorithm}' is outside the MI355X hot path (SURVEY.md section 8: torchode/torchsde "
            "generators are out of scope)")
    return PredictorCorrectorSamplingParameters(**sampling_parameter_dictionary)
