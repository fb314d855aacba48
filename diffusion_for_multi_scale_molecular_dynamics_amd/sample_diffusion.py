"""Entry point: draw samples with the MI355X hot path (drop-in for src/.../sample_diffusion.py:52-274).

Same flags, YAML surface (`noise:`, `sampling:`, optional `elements:`), and output files (`samples.pt` =
{"cartesian_positions", "original_axl"}, `trajectories.pt`, `config_backup.yaml`, `console.log`).
Differences, all additive:
  * the score network comes from `--checkpoint` as in the reference -- the trainer's Lightning checkpoint: rebuilt from the
    hyper-parameters pickled inside it, weights under the prefix `axl_network.` (read without Lightning and without the
    reference package: utils/lightning_checkpoint.py) -- or, additionally, from a bare state_dict together with a `model:
    score_network:` block in the config, or is randomly initialised with `--random_init_seed` (synthetic benchmarks);
  * under `torchrun` (one process per GPU) the sub-batches are sharded over the ranks and gathered once (RCCL);
    rank 0 writes the files (`trajectories.pt` holds every rank's recorded sub-batches, in sub-batch order);
  * LAMMPS energies (`oracle:`) and Orion reporting are outside the hot path and are not evaluated.
"""
import argparse
import logging
import os
import shutil
import socket
import sys
from pathlib import Path
from typing import Any, AnyStr, Dict, Optional

import torch
import yaml

from .generators.instantiate_generator import instantiate_generator
from .generators.load_sampling_parameters import load_sampling_parameters
from .generators.sampling_constraint import read_sampling_constraint
from .generators.trajectory_initializer import instantiate_trajectory_initializer
from .models.score_networks.score_network import ScoreNetwork
from .data.element_types import ElementTypes
from .models.score_networks.score_network_factory import create_score_network, create_score_network_parameters
from .noise_schedulers.noise_parameters import NoiseParameters
from .sampling.diffusion_sampling import create_batch_of_samples_sharded

logger = logging.getLogger(__name__)


def extract_and_validate_parameters(hyper_params: Dict[AnyStr, Any]):
    """:168-188"""
    assert "noise" in hyper_params, "The noise parameters must be defined to draw samples."
    assert "sampling" in hyper_params, "The sampling parameters must be defined to draw samples."
    return NoiseParameters(**hyper_params["noise"]), load_sampling_parameters(hyper_params["sampling"])


def global_parameters_of(hyper_params: Dict[AnyStr, Any]) -> Optional[Dict[AnyStr, Any]]:
    """What the reference hands to create_score_network_parameters beside the `model: score_network:` block: for a TRAINING
    configuration (top-level `elements` and `data: max_atom`) exactly its dict(max_atom, spatial_dimension -- 3 when the file
    does not say --, elements) (models/instantiate_diffusion_model.py:34-41: a block that contradicts it is refused, as there);
    otherwise whichever of `elements` / `spatial_dimension` the file has; None when it has neither (a sampling configuration
    that spells its block out)."""
    data = hyper_params.get("data")
    if "elements" in hyper_params and isinstance(data, dict) and "max_atom" in data:
        return dict(max_atom=data["max_atom"], spatial_dimension=hyper_params.get("spatial_dimension", 3),
                    elements=hyper_params["elements"])
    out = {key: hyper_params[key] for key in ("elements", "spatial_dimension") if key in hyper_params}
    return out or None


def get_axl_network(checkpoint_path, hyper_params: Optional[Dict[AnyStr, Any]] = None) -> ScoreNetwork:
    """:191-205.  The reference rebuilds the network from the hyper-parameters INSIDE the Lightning checkpoint
    (`load_from_checkpoint`), so its sampling configurations hold no `model:` block; so does this -- without Lightning and
    without the reference package (utils/lightning_checkpoint.py) -- and loads the `axl_network.*` weights.  A `model:
    score_network:` block in the configuration, when there is one, takes precedence (a bare state_dict has nothing else)."""
    from .utils.lightning_checkpoint import load_checkpoint, score_network_parameters_of, state_dict_of
    checkpoint = load_checkpoint(checkpoint_path)
    hyper_params = hyper_params or {}
    if "model" in hyper_params and "score_network" in hyper_params["model"]:
        parameters = create_score_network_parameters(hyper_params["model"]["score_network"], global_parameters_of(hyper_params))
    else:
        parameters = score_network_parameters_of(checkpoint)
        assert parameters is not None, \
            "the checkpoint holds no hyper-parameters (not a Lightning checkpoint of the reference's trainer): the config " \
            "must contain the `model: score_network:` block that describes its network"
    network = create_score_network(parameters)
    network.load_state_dict(state_dict_of(checkpoint))
    return network.eval()


def _init_distributed(device: torch.device):
    if "RANK" not in os.environ:            # (under torchrun the group is set up at any world size, 1 included)
        return 0, 1, device
    import torch.distributed as dist
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if device.type == "cuda":
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl" if device.type == "cuda" else "gloo")
    return dist.get_rank(), dist.get_world_size(), device


def main(args: Optional[Any] = None, axl_network: Optional[ScoreNetwork] = None) -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", required=True, help="config file with sampling parameters in yaml format.")
    parser.add_argument("--checkpoint", default=None, help="path to checkpoint model to be loaded.")
    parser.add_argument("--output", required=True, help="path to outputs - will store files here")
    parser.add_argument("--path_to_starting_configuration_data_pickle", default=None)
    parser.add_argument("--path_to_sampling_constraint_data_pickle", default=None)
    parser.add_argument("--device", default="cuda", help="Device to use. Defaults to cuda.")
    parser.add_argument("--reference_pickles", action="store_true",
                        help="write samples.pt / trajectories.pt naming the REFERENCE's AXL class, so that the reference's own "
                             "tools read them with a plain torch.load (default: this package's class)")
    parser.add_argument("--random_init_seed", type=int, default=None,
                        help="build the network of `model.score_network` with random weights (synthetic runs)")
    args = parser.parse_args(args)

    rank, world, device = _init_distributed(torch.device(args.device))
    os.makedirs(args.output, exist_ok=True)
    if rank == 0:
        logging.basicConfig(level=logging.INFO, force=True,
                            handlers=[logging.StreamHandler(sys.stdout),
                                      logging.FileHandler(os.path.join(args.output, "console.log"))])
        shutil.copyfile(args.config, os.path.join(args.output, "config_backup.yaml"))
    with open(args.config) as fd:
        hyper_params = yaml.safe_load(fd)
    logger.info("Sampling Experiment info:\n  Hostname : %s\n  Checkpoint : %s\n  Device   : %s\n  Ranks    : %d",
                socket.gethostname(), args.checkpoint, device, world)

    noise_parameters, sampling_parameters = extract_and_validate_parameters(hyper_params)
    if "elements" in hyper_params:
        ElementTypes.validate_elements(hyper_params["elements"])
    if getattr(sampling_parameters, "rng_mode", "reference") == "reference":
        logger.info("Sampling in the reference's mode (draws from torch's CPU generator in the reference's order, uploaded every step, "
                    "eager launches: reproduces the reference's run for a given torch.manual_seed).  `rng_mode: device` and "
                    "`use_hip_graph: true` in the `sampling:` block select the throughput modes (INTEGRATION.md).")
    if "oracle" in hyper_params:
        logger.warning("The configuration has an `oracle:` block: the energy oracle (LAMMPS) is outside this package's scope; "
                       "samples.pt is written, energies.pt is not.")
    if axl_network is None:
        if args.random_init_seed is not None:
            torch.manual_seed(args.random_init_seed)
            axl_network = create_score_network(
                create_score_network_parameters(hyper_params["model"]["score_network"], global_parameters_of(hyper_params))).eval()
        else:
            assert args.checkpoint is not None and os.path.exists(args.checkpoint), \
                f"The path {args.checkpoint} does not exist. Cannot go on."
            axl_network = get_axl_network(args.checkpoint, hyper_params)
    axl_network = axl_network.to(device)
    if "force_field" in hyper_params:                                  # src/sample_diffusion.py:132-139
        from .models.score_networks.force_field_augmented_score_network import (ForceFieldAugmentedScoreNetwork,
                                                                                 ForceFieldParameters)
        force_field_parameters = ForceFieldParameters(**hyper_params["force_field"])
        if force_field_parameters.radial_cutoff > 0.0:
            logger.info("Augmenting the AXL_network with an excluding Force Field.")
            axl_network = ForceFieldAugmentedScoreNetwork(axl_network, force_field_parameters)
        else:
            logger.info("Force field parameters are present, but the radial cutoff is zero. Using original AXL network")

    trajectory_initializer = instantiate_trajectory_initializer(
        sampling_parameters=sampling_parameters,
        path_to_starting_configuration_data_pickle=args.path_to_starting_configuration_data_pickle)
    sampling_constraints = None
    if args.path_to_sampling_constraint_data_pickle is not None:
        sampling_constraints = read_sampling_constraint(args.path_to_sampling_constraint_data_pickle)
    generator = instantiate_generator(sampling_parameters=sampling_parameters, noise_parameters=noise_parameters,
                                      axl_network=axl_network, trajectory_initializer=trajectory_initializer,
                                      sampling_constraints=sampling_constraints)
    create_samples_and_write_to_disk(generator, sampling_parameters, None, device, args.output, rank,
                                     for_reference=args.reference_pickles)


def create_samples_and_write_to_disk(generator, sampling_parameters, oracle_parameters, device, output_path, rank: int = 0,
                                     for_reference: bool = False):
    """:208-270.  oracle_parameters: the reference's third argument (an energy oracle to evaluate the samples with, LAMMPS):
    outside this package's scope -- anything but None is refused; Orion reporting likewise absent."""
    if oracle_parameters is not None:
        raise NotImplementedError("energy oracles (LAMMPS) are outside this package's scope: pass oracle_parameters=None")
    logger.info("Generating samples...")
    with torch.no_grad():
        samples_batch = create_batch_of_samples_sharded(generator=generator, sampling_parameters=sampling_parameters,
                                                        device=device)
    logger.info("Done Generating Samples.")
    fallbacks = getattr(generator, "f16_range_fallbacks", 0)
    if fallbacks:
        logger.warning("%d sampler iteration(s) were recomputed with the exact-f32 MFMA kernels: the split-f16 edge chain met "
                       "values beyond the f16 range (set edge_chain_precision='f32' on the network to avoid the retries)",
                       fallbacks)
    output_directory = Path(output_path)
    if sampling_parameters.record_samples:
        write_trajectories(generator.sample_trajectory_recorder, sampling_parameters, output_directory, for_reference)
    if rank != 0:
        return
    if for_reference:
        from .utils import reference_pickles
        reference_pickles.save_for_reference(samples_batch, output_directory / "samples.pt")
    else:
        with open(output_directory / "samples.pt", "wb") as fd:
            torch.save(samples_batch, fd)
    logger.info("Done!")


def write_trajectories(recorder, sampling_parameters, output_directory: Path, for_reference: bool = False):
    """`trajectories.pt` of the WHOLE run (src/sample_diffusion.py:253-257).  Under several ranks every rank has recorded its own
    sub-batches: each writes `trajectories.rank{r}.pt` into the (shared, single-node) output directory, and after a barrier
    rank 0 puts the entries back in sub-batch order -- the file a single process would have written -- and removes the
    per-rank files.  No collective carries the records: they are T x (1 + M) compositions per sub-batch."""
    import torch.distributed as dist
    from .sampling.diffusion_sampling import split_sizes
    from .utils.sample_trajectory import merge_sharded_entries
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        recorder.write_to_pickle(output_directory / "trajectories.pt", for_reference=for_reference)
        return
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.save(recorder.entries(), output_directory / f"trajectories.rank{rank}.pt")
    dist.barrier()
    if rank == 0:
        paths = [output_directory / f"trajectories.rank{r}.pt" for r in range(world)]
        per_rank = [torch.load(path, weights_only=False) for path in paths]
        sizes = split_sizes(sampling_parameters.number_of_samples, sampling_parameters.sample_batchsize)
        merge_sharded_entries(per_rank, sizes, world).write_to_pickle(output_directory / "trajectories.pt", for_reference=for_reference)
        for path in paths:
            os.remove(path)
    dist.barrier()


if __name__ == "__main__":
    main()
