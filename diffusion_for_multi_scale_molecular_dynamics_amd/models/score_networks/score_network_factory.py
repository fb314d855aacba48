"""Registry / factory for score networks (src/.../models/score_networks/score_network_factory.py:25-125),
restricted to the architectures the sampling configs name (mlp, egnn)."""
import dataclasses
from typing import Any, AnyStr, Dict, Optional

from .egnn_score_network import EGNNScoreNetwork, EGNNScoreNetworkParameters
from .mlp_score_network import MLPScoreNetwork, MLPScoreNetworkParameters
from .score_network import ScoreNetwork, ScoreNetworkParameters

SCORE_NETWORKS_BY_ARCH = dict(mlp=MLPScoreNetwork, egnn=EGNNScoreNetwork)
SCORE_NETWORK_PARAMETERS_BY_ARCH = dict(mlp=MLPScoreNetworkParameters, egnn=EGNNScoreNetworkParameters)


def create_score_network_parameters(score_network_dictionary: Dict[AnyStr, Any],
                                    global_parameters_dictionary: Optional[Dict[AnyStr, Any]] = None) -> ScoreNetworkParameters:
    """:64-125.  global_parameters_dictionary (the reference passes dict(max_atom, spatial_dimension, elements) built from the
    top of a training configuration, models/instantiate_diffusion_model.py:34-41): `elements` must have num_atom_types entries,
    a key given in both places must agree, and a global key that is a field of the architecture's dataclass completes the
    block.  None (a sampling configuration that spells its `model: score_network:` block out): the block alone."""
    assert "architecture" in score_network_dictionary, "The architecture of the score network must be specified."
    architecture = score_network_dictionary["architecture"]
    assert architecture in SCORE_NETWORK_PARAMETERS_BY_ARCH, \
        f"Architecture {architecture} is not implemented. Choices: {list(SCORE_NETWORK_PARAMETERS_BY_ARCH)}"
    dataclass = SCORE_NETWORK_PARAMETERS_BY_ARCH[architecture]
    augmented = dict(score_network_dictionary)
    if global_parameters_dictionary is not None:
        if "elements" in global_parameters_dictionary:
            assert len(global_parameters_dictionary["elements"]) == score_network_dictionary["num_atom_types"], \
                "There should be 'num_atom_types' entries in the 'elements' list."
        for key, value in augmented.items():
            if key in global_parameters_dictionary:
                assert global_parameters_dictionary[key] == value, f"inconsistent configuration values for {key}"
        fields = [field.name for field in dataclasses.fields(dataclass)]
        for key, value in global_parameters_dictionary.items():
            if key in fields:
                augmented[key] = value
    return dataclass(**augmented)


def create_score_network(score_network_parameters: ScoreNetworkParameters) -> ScoreNetwork:
    architecture = score_network_parameters.architecture
    assert architecture in SCORE_NETWORKS_BY_ARCH, f"Architecture {architecture} is not implemented."
    expected = SCORE_NETWORK_PARAMETERS_BY_ARCH[architecture]
    assert isinstance(score_network_parameters, expected), \
        f"{type(score_network_parameters).__name__} does not match architecture {architecture}"
    assert dataclasses.is_dataclass(score_network_parameters)
    return SCORE_NETWORKS_BY_ARCH[architecture](score_network_parameters)
