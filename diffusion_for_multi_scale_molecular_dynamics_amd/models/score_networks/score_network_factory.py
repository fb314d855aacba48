"""Registry / factory for score networks (src/.../models/score_networks/score_network_factory.py:25-125),
restricted to the architectures the sampling configs name (mlp, egnn)."""
import dataclasses
from typing import Any, AnyStr, Dict

from .egnn_score_network import EGNNScoreNetwork, EGNNScoreNetworkParameters
from .mlp_score_network import MLPScoreNetwork, MLPScoreNetworkParameters
from .score_network import ScoreNetwork, ScoreNetworkParameters

SCORE_NETWORKS_BY_ARCH = dict(mlp=MLPScoreNetwork, egnn=EGNNScoreNetwork)
SCORE_NETWORK_PARAMETERS_BY_ARCH = dict(mlp=MLPScoreNetworkParameters, egnn=EGNNScoreNetworkParameters)


def create_score_network_parameters(score_network_dictionary: Dict[AnyStr, Any]) -> ScoreNetworkParameters:
    assert "architecture" in score_network_dictionary, "The architecture of the score network must be specified."
    architecture = score_network_dictionary["architecture"]
    assert architecture in SCORE_NETWORK_PARAMETERS_BY_ARCH, \
        f"Architecture {architecture} is not implemented. Choices: {list(SCORE_NETWORK_PARAMETERS_BY_ARCH)}"
    return SCORE_NETWORK_PARAMETERS_BY_ARCH[architecture](**score_network_dictionary)


def create_score_network(score_network_parameters: ScoreNetworkParameters) -> ScoreNetwork:
    architecture = score_network_parameters.architecture
    assert architecture in SCORE_NETWORKS_BY_ARCH, f"Architecture {architecture} is not implemented."
    expected = SCORE_NETWORK_PARAMETERS_BY_ARCH[architecture]
    assert isinstance(score_network_parameters, expected), \
        f"{type(score_network_parameters).__name__} does not match architecture {architecture}"
    assert dataclasses.is_dataclass(score_network_parameters)
    return SCORE_NETWORKS_BY_ARCH[architecture](score_network_parameters)
