"""Batch driver (src/.../sampling/diffusion_sampling.py:16-73) + its multi-GPU form.

create_batch_of_samples            same signature and output dict as the reference; loops generator.sample over
                                   sub-batches on one device.
create_batch_of_samples_sharded    one process per GPU: the sub-batches are dealt round-robin to the ranks (samples
                                   are independent, there is no per-step communication) and the results are
                                   gathered with ONE collective per field at the end (RCCL over xGMI under the
                                   "nccl" backend; "gloo" in the CPU tests).
"""
import logging
from typing import List, Tuple

import torch

from ..generators.axl_generator import AXLGenerator, SamplingParameters
from ..namespace import AXL, AXL_COMPOSITION, CARTESIAN_POSITIONS

logger = logging.getLogger(__name__)


def _finish(atom_types, relative_coordinates, lattice_parameters):
    """Zero the angle entries in place and add Cartesian positions (:52-66); X @ diag(L[:d]) as a broadcast."""
    d = relative_coordinates.shape[-1]
    lattice_parameters[..., d:] = 0
    cartesian_positions = relative_coordinates * lattice_parameters[:, None, :d]
    return {CARTESIAN_POSITIONS: cartesian_positions,
            AXL_COMPOSITION: AXL(A=atom_types, X=relative_coordinates, L=lattice_parameters)}


def split_sizes(number_of_samples: int, sample_batchsize) -> List[int]:
    bs = number_of_samples if sample_batchsize is None else sample_batchsize
    full, rest = divmod(number_of_samples, bs)
    return [bs] * full + ([rest] if rest else [])


def create_batch_of_samples(generator: AXLGenerator, sampling_parameters: SamplingParameters, device: torch.device):
    logger.info("Creating a batch of samples")
    parts = [generator.sample(n, device=device)
             for n in split_sizes(sampling_parameters.number_of_samples, sampling_parameters.sample_batchsize)]
    return _finish(torch.concat([p.A for p in parts]), torch.concat([p.X for p in parts]),
                   torch.concat([p.L for p in parts]))


def shard_of_rank(sizes: List[int], rank: int, world_size: int) -> List[Tuple[int, int]]:
    """(sub-batch position, size) pairs owned by `rank`: round-robin over the sub-batch list."""
    return [(k, n) for k, n in enumerate(sizes) if k % world_size == rank]


def create_batch_of_samples_sharded(generator: AXLGenerator, sampling_parameters: SamplingParameters,
                                    device: torch.device, group=None):
    """Every rank returns the full batch, ordered by sub-batch position (so the result does not depend on the
    number of ranks when the generator's draws do not)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return create_batch_of_samples(generator, sampling_parameters, device)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sizes = split_sizes(sampling_parameters.number_of_samples, sampling_parameters.sample_batchsize)
    mine = shard_of_rank(sizes, rank, world)
    parts = [generator.sample(n, device=device) for _, n in mine]
    n_atoms, d = sampling_parameters.number_of_atoms, sampling_parameters.spatial_dimension
    nl = d * (d + 1) // 2

    def cat(ts, shape, dtype):
        return torch.concat(ts) if ts else torch.empty((0,) + shape, dtype=dtype, device=device)

    local = AXL(A=cat([p.A for p in parts], (n_atoms,), torch.int64),
                X=cat([p.X for p in parts], (n_atoms, d), torch.float32),
                L=cat([p.L for p in parts], (nl,), torch.float32))
    # ranks can own different numbers of samples: pad to the largest shard, gather once per field, trim
    counts = [sum(n for _, n in shard_of_rank(sizes, r, world)) for r in range(world)]
    width = max(counts)

    def gather(t):
        padded = torch.zeros((width,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        padded[: t.shape[0]] = t
        out = torch.empty((world * width,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, padded, group=group)
        return out.view((world, width) + tuple(t.shape[1:]))

    gathered = AXL(A=gather(local.A), X=gather(local.X), L=gather(local.L))
    # restore sub-batch order
    pieces = {}
    for r in range(world):
        offset = 0
        for k, n in shard_of_rank(sizes, r, world):
            pieces[k] = (r, offset, n)
            offset += n
    order = [pieces[k] for k in range(len(sizes))]
    pick = lambda g: torch.concat([g[r, o:o + n] for r, o, n in order])   # noqa: E731
    return _finish(pick(gathered.A), pick(gathered.X), pick(gathered.L))
