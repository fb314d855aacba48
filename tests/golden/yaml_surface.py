"""The reference's configuration surface against this package's, file by file (container-only: imports /root/reference).

    PYTHONPATH=/root/reference/src python tests/golden/yaml_surface.py REFERENCE_ROOT OUT_DIR  > report.json

For every YAML file under the reference tree:
  * every block that holds `noise:` and `sampling:` with `algorithm: predictor_corrector` is loaded by the REFERENCE's own
    NoiseParameters / PredictorCorrectorSamplingParameters and by this package's NoiseParameters / load_sampling_parameters: the reference's fields must come out equal;
  * every `score_network:` block with an architecture this package implements (mlp, egnn) goes through both
    create_score_network_parameters (with the file's global parameters, models/instantiate_diffusion_model.py:34-41), both
    networks are built, and their state_dict keys and shapes compared (a checkpoint of one loads into the other).
Then two checkpoints are written into OUT_DIR the way Lightning writes them -- `state_dict` with the `axl_network.` prefix and
`hyper_parameters = {"hyper_params": AXLDiffusionParameters(score_network_parameters = <the REFERENCE's dataclass instance>, ...)}`
(the Lightning module itself cannot be imported here -- no `lightning` -- so the outer dataclass is declared under its module
path by this script; the score-network parameters, the optimizer parameters and the loss weights inside are the reference's
own objects) -- for tests/test_reference_yaml_surface.py to read back WITHOUT the reference on the path.
"""
import dataclasses
import glob
import json
import os
import sys
import types

import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden  # noqa: E402,F401  (installs the container-only stubs for pykeops / e3nn / torch_geometric)

# (generators/load_sampling_parameters.py cannot be imported here -- it pulls in the ode / sde generators, hence torchode / torchsde --
# its `predictor_corrector | adaptive_corrector` branch is PredictorCorrectorSamplingParameters(**dictionary), :38-42)
from diffusion_for_multi_scale_molecular_dynamics.generators.predictor_corrector_axl_generator import \
    PredictorCorrectorSamplingParameters as RefSampling  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics.models.optimizer import OptimizerParameters  # noqa: E402
# (models/score_networks/score_network_factory.py cannot be imported either -- it registers the MACE networks, hence `mace` -- so
# its two functions are applied to the reference's OWN dataclasses and modules below: reference_network_parameters restates
# :64-125 -- elements check, contradiction check, completion of the block by the global keys that are fields -- and
# create_score_network (:47-61) is `SCORE_NETWORKS_BY_ARCH[architecture](parameters)`)
from diffusion_for_multi_scale_molecular_dynamics.models.score_networks.egnn_score_network import (  # noqa: E402
    EGNNScoreNetwork as RefEGNN, EGNNScoreNetworkParameters as RefEGNNParameters)
from diffusion_for_multi_scale_molecular_dynamics.models.score_networks.mlp_score_network import (  # noqa: E402
    MLPScoreNetwork as RefMLP, MLPScoreNetworkParameters as RefMLPParameters)
from diffusion_for_multi_scale_molecular_dynamics.namespace import AXL  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics.noise_schedulers.noise_parameters import NoiseParameters as RefNoise  # noqa: E402

from diffusion_for_multi_scale_molecular_dynamics_amd.generators.load_sampling_parameters import load_sampling_parameters as own_sampling  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.score_network_factory import (  # noqa: E402
    SCORE_NETWORK_PARAMETERS_BY_ARCH, create_score_network as own_network, create_score_network_parameters as own_network_parameters)
from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters as OwnNoise  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd.sample_diffusion import global_parameters_of  # noqa: E402


REFERENCE_CLASSES = dict(mlp=(RefMLPParameters, RefMLP), egnn=(RefEGNNParameters, RefEGNN))


def ref_network_parameters(block, global_parameters):
    assert len(global_parameters["elements"]) == block["num_atom_types"]
    dataclass = REFERENCE_CLASSES[block["architecture"]][0]
    augmented = dict(block)
    for key, value in augmented.items():
        if key in global_parameters:
            assert global_parameters[key] == value, f"inconsistent configuration values for {key}"
    fields = [field.name for field in dataclasses.fields(dataclass)]
    for key, value in global_parameters.items():
        if key in fields:
            augmented[key] = value
    return dataclass(**augmented)


def ref_network(parameters):
    return REFERENCE_CLASSES[parameters.architecture][1](parameters)


def walk(node, path=()):
    if isinstance(node, dict):
        yield path, node
        for key, value in node.items():
            yield from walk(value, path + (key,))
    elif isinstance(node, list):
        for k, value in enumerate(node):
            yield from walk(value, path + (k,))


def plain(value):
    if dataclasses.is_dataclass(value):
        return {f.name: plain(getattr(value, f.name)) for f in dataclasses.fields(value)}
    if isinstance(value, (list, tuple)):
        return [plain(v) for v in value]
    if isinstance(value, torch.Tensor):
        return value.tolist()
    return value


def same_on_reference_fields(ref, own):
    r, o = plain(ref), plain(own)
    return [key for key in r if key not in o or r[key] != o[key]]


def main():
    root, out_dir = sys.argv[1], sys.argv[2]
    report = dict(files=0, pairs=[], networks=[], skipped_architectures=[], field_differences={}, unreadable=[])
    for path in sorted(glob.glob(os.path.join(root, "**", "*.yaml"), recursive=True)):
        rel = os.path.relpath(path, root)
        try:
            config = yaml.safe_load(open(path))
        except yaml.YAMLError:
            report["unreadable"].append(rel)      # (multi-document LAMMPS dumps: not configurations)
            continue
        report["files"] += 1
        for where, block in walk(config):
            if "noise" in block and "sampling" in block and isinstance(block["sampling"], dict) and \
                    block["sampling"].get("algorithm") == "predictor_corrector":
                entry = dict(file=rel, where="/".join(map(str, where)))
                try:
                    reference_objects = RefNoise(**block["noise"]), RefSampling(**block["sampling"])
                except TypeError as refused:
                    # a block the reference's own classes refuse (a key that is not a field): this package must refuse it too
                    entry["reference_refuses"] = str(refused)[:200]
                    try:
                        OwnNoise(**block["noise"]), own_sampling(block["sampling"])
                        entry["own_refuses"] = None
                    except TypeError as own_refused:
                        entry["own_refuses"] = str(own_refused)[:200]
                    report["pairs"].append(entry)
                    continue
                entry["differing"] = same_on_reference_fields(reference_objects[0], OwnNoise(**block["noise"])) + \
                    same_on_reference_fields(reference_objects[1], own_sampling(block["sampling"]))
                report["pairs"].append(entry)
            if where and where[-1] == "score_network" and "architecture" in block:
                if block["architecture"] not in SCORE_NETWORK_PARAMETERS_BY_ARCH:
                    report["skipped_architectures"].append(dict(file=rel, architecture=block["architecture"]))
                    continue
                entry = dict(file=rel, where="/".join(map(str, where)), architecture=block["architecture"])
                top_level = where == ("model", "score_network") and "elements" in config and "data" in config
                if top_level:                      # the training set-up's call (instantiate_diffusion_model.py:34-45)
                    reference_globals = dict(max_atom=config["data"]["max_atom"], spatial_dimension=config.get("spatial_dimension", 3),
                                             elements=config["elements"])
                    own_globals = global_parameters_of(config)
                else:
                    reference_globals = dict(elements=["X%d" % k for k in range(block["num_atom_types"])])
                    own_globals = None
                try:
                    ref_p = ref_network_parameters(block, reference_globals)
                except (AssertionError, TypeError) as refused:
                    # a file the reference refuses (configuration_templates/.../config_diffusion_mlp.yaml sets spatial_dimension 1
                    # in the block and nothing at the top: "inconsistent configuration values"): this package must refuse it too
                    entry["reference_refuses"] = f"{type(refused).__name__}: {refused}"[:200]
                    try:
                        own_network_parameters(block, own_globals)
                        entry["own_refuses"] = None
                    except (AssertionError, TypeError) as own_refused:
                        entry["own_refuses"] = f"{type(own_refused).__name__}: {own_refused}"[:200]
                    report["networks"].append(entry)
                    continue
                own_p = own_network_parameters(block, own_globals)
                entry["differing"] = same_on_reference_fields(ref_p, own_p)
                torch.manual_seed(0)
                try:
                    ref_state = ref_network(ref_p).state_dict()
                except Exception as unbuildable:          # noqa: BLE001  (an Orion template: 'orion~choices(...)' strings for numbers)
                    entry["reference_cannot_build"] = f"{type(unbuildable).__name__}: {unbuildable}"[:200]
                    report["networks"].append(entry)
                    continue
                own_state = own_network(own_p).state_dict()
                entry["state_dict_matches"] = list(ref_state) == list(own_state) and all(
                    tuple(ref_state[k].shape) == tuple(own_state[k].shape) and ref_state[k].dtype == own_state[k].dtype for k in ref_state)
                entry["tensors"] = len(ref_state)
                report["networks"].append(entry)

    # Lightning-style checkpoints around the reference's own parameter objects
    module = types.ModuleType("diffusion_for_multi_scale_molecular_dynamics.models.axl_diffusion_lightning_model")
    sys.modules[module.__name__] = module
    AXLDiffusionParameters = dataclasses.make_dataclass(
        "AXLDiffusionParameters", [("score_network_parameters", object), ("loss_parameters", object), ("optimizer_parameters", object),
                                   ("scheduler_parameters", object, None), ("kmax_target_score", int, 4),
                                   ("regularizer_parameters", object, None), ("diffusion_sampling_parameters", object, None),
                                   ("oracle_parameters", object, None)])       # (models/axl_diffusion_lightning_model.py:62-73)
    AXLDiffusionParameters.__module__ = module.__name__
    module.AXLDiffusionParameters = AXLDiffusionParameters
    blocks = dict(
        mlp=yaml.safe_load(open(os.path.join(root, "analysis_and_sanity_checks/atom_types_only_experiments/training/config.yaml"))),
        egnn=yaml.safe_load(open(os.path.join(
            root, "experiments/training_and_sampling_generative_models/inputs_and_scripts/SiGe_1x1x1/config_diffusion_egnn.yaml"))))
    report["checkpoints"] = {}
    for name, config in blocks.items():
        block = config["model"]["score_network"]
        parameters = ref_network_parameters(block, dict(max_atom=config["data"]["max_atom"],
                                                        spatial_dimension=config.get("spatial_dimension", 3), elements=config["elements"]))
        torch.manual_seed(11)
        network = ref_network(parameters)
        hyper = AXLDiffusionParameters(score_network_parameters=parameters, loss_parameters=AXL(A=1.0, X=1.0, L=1.0),
                                       optimizer_parameters=OptimizerParameters(name="adamw", learning_rate=1e-3, weight_decay=0.0))
        state = {"axl_network." + k: v for k, v in network.state_dict().items()}
        state["loss_weights"] = torch.ones(3)                      # (a Lightning module's state_dict holds more than the network)
        checkpoint = {"epoch": 3, "global_step": 120, "pytorch-lightning_version": "2.2.1", "state_dict": state,
                      "hparams_name": "hyper_params", "hyper_parameters": {"hyper_params": hyper}, "optimizer_states": [], "lr_schedulers": []}
        file = os.path.join(out_dir, f"last_model_{name}.ckpt")
        torch.save(checkpoint, file)
        report["checkpoints"][name] = dict(file=file, parameters=plain(parameters), tensors=len(network.state_dict()),
                                           checksum=float(sum(v.double().sum() for v in network.state_dict().values())))
    print(json.dumps(report))


if __name__ == "__main__":
    main()
