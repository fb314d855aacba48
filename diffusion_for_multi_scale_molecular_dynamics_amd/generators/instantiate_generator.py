"""Generator factory (src/.../generators/instantiate_generator.py:25-82)."""
from typing import Optional

from ..models.score_networks.score_network import ScoreNetwork
from ..noise_schedulers.noise_parameters import NoiseParameters
from .adaptive_corrector import AdaptiveCorrectorGenerator
from .axl_generator import SamplingParameters
from .constrained_langevin_generator import ConstrainedLangevinGenerator
from .langevin_generator import LangevinGenerator
from .sampling_constraint import SamplingConstraint
from .trajectory_initializer import TrajectoryInitializer


def instantiate_generator(sampling_parameters: SamplingParameters, noise_parameters: NoiseParameters,
                          axl_network: ScoreNetwork, trajectory_initializer: TrajectoryInitializer,
                          sampling_constraints: Optional[SamplingConstraint] = None):
    assert sampling_parameters.algorithm in ["ode", "sde", "predictor_corrector", "adaptive_corrector"], \
        "Unknown algorithm. Possible choices are 'ode', 'sde', 'predictor_corrector' and 'adaptive_corrector'"
    if sampling_constraints is not None:
        assert sampling_parameters.algorithm == "predictor_corrector", \
            "Only the 'predictor_corrector' scheme supports sampling constraints."
        return ConstrainedLangevinGenerator(noise_parameters=noise_parameters,
                                            sampling_parameters=sampling_parameters, axl_network=axl_network,
                                            sampling_constraints=sampling_constraints,
                                            trajectory_initializer=trajectory_initializer)
    if sampling_parameters.algorithm == "predictor_corrector":
        return LangevinGenerator(sampling_parameters=sampling_parameters, noise_parameters=noise_parameters,
                                 axl_network=axl_network, trajectory_initializer=trajectory_initializer)
    if sampling_parameters.algorithm == "adaptive_corrector":
        return AdaptiveCorrectorGenerator(sampling_parameters=sampling_parameters, noise_parameters=noise_parameters,
                                          axl_network=axl_network, trajectory_initializer=trajectory_initializer)
    raise NotImplementedError(f"algorithm '{sampling_parameters.algorithm}' is outside the MI355X hot path "
                              "(SURVEY.md section 8)")
