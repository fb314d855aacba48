"""From a `sampling.algorithm` string to a generator object (reference: generators/instantiate_generator.py:25-82).

The MI355X hot path covers the predictor-corrector family; the reference's torchsde / ODE samplers are named in the
accepted list (so that a typo and an unsupported choice give different errors) but are not built here.
"""
from typing import Optional

from ..models.score_networks.score_network import ScoreNetwork
from ..noise_schedulers.noise_parameters import NoiseParameters
from .adaptive_corrector import AdaptiveCorrectorGenerator
from .axl_generator import SamplingParameters
from .constrained_langevin_generator import ConstrainedLangevinGenerator
from .langevin_generator import LangevinGenerator
from .sampling_constraint import SamplingConstraint
from .trajectory_initializer import TrajectoryInitializer

_KNOWN_ALGORITHMS = ("ode", "sde", "predictor_corrector", "adaptive_corrector")
_BUILT = {"predictor_corrector": LangevinGenerator, "adaptive_corrector": AdaptiveCorrectorGenerator}


def instantiate_generator(sampling_parameters: SamplingParameters, noise_parameters: NoiseParameters,
                          axl_network: ScoreNetwork, trajectory_initializer: TrajectoryInitializer,
                          sampling_constraints: Optional[SamplingConstraint] = None):
    algorithm = sampling_parameters.algorithm
    assert algorithm in _KNOWN_ALGORITHMS, f"Unknown algorithm '{algorithm}'; choose one of {_KNOWN_ALGORITHMS}."
    common = dict(noise_parameters=noise_parameters, sampling_parameters=sampling_parameters, axl_network=axl_network,
                  trajectory_initializer=trajectory_initializer)
    if sampling_constraints is not None:        # repaint
        assert algorithm == "predictor_corrector", "sampling constraints need the 'predictor_corrector' algorithm."
        return ConstrainedLangevinGenerator(sampling_constraints=sampling_constraints, **common)
    if algorithm not in _BUILT:
        raise NotImplementedError(f"algorithm '{algorithm}' is outside the MI355X hot path (SURVEY.md section 8)")
    return _BUILT[algorithm](**common)
