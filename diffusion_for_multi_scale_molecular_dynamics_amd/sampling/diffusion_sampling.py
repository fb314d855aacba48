"""Batch driver (src/.../sampling/diffusion_sampling.py:16-73) + its multi-GPU form.

create_batch_of_samples            same signature and output dict as the reference; loops generator.sample over
                                   sub-batches on one device.
create_batch_of_samples_sharded    one process per GPU: the sub-batches are dealt round-robin to the ranks (samples
                                   are independent, there is no per-step communication) and the results are
                                   gathered with ONE collective at the end: A, X and L of a structure are packed
                                   into one byte row, so the job has a single all_gather_into_tensor (RCCL over xGMI
                                   under the "nccl" backend; "gloo" in the CPU tests).
"""
import logging
import math
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


def pack_compositions(composition: AXL) -> torch.Tensor:
    """A [B,N] int64 | X [B,N,d] f32 | L [B,nl] f32  ->  uint8 [B, 8 N + 4 N d + 4 nl]: one row of bytes per structure."""
    batch = composition.X.shape[0]
    return torch.cat([t.contiguous().view(torch.uint8).reshape(batch, t.element_size() * math.prod(t.shape[1:]))
                      for t in composition], dim=1)


def unpack_compositions(rows: torch.Tensor, number_of_atoms: int, spatial_dimension: int) -> AXL:
    """Inverse of pack_compositions for rows [..., bytes]."""
    n, d = number_of_atoms, spatial_dimension
    nl = d * (d + 1) // 2
    lead = tuple(rows.shape[:-1])
    a, x, lat = torch.split(rows, [8 * n, 4 * n * d, 4 * nl], dim=-1)
    return AXL(A=a.contiguous().view(torch.int64).reshape(lead + (n,)),
               X=x.contiguous().view(torch.float32).reshape(lead + (n, d)),
               L=lat.contiguous().view(torch.float32).reshape(lead + (nl,)))


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
    # ranks can own different numbers of samples: pad to the largest shard, ONE gather of the packed rows, trim
    counts = [sum(n for _, n in shard_of_rank(sizes, r, world)) for r in range(world)]
    width = max(counts)
    rows = pack_compositions(local)
    padded = torch.zeros((width, rows.shape[1]), dtype=torch.uint8, device=rows.device)
    padded[: rows.shape[0]] = rows
    if dist.get_backend(group) == "gloo" and padded.is_cuda:
        # (rehearsals of the multi-rank flow on one GPU, and the per-shard parity test: gloo moves host memory)
        host = torch.empty((world * width, rows.shape[1]), dtype=torch.uint8)
        dist.all_gather_into_tensor(host, padded.cpu(), group=group)
        out = host.to(rows.device)
    else:
        out = torch.empty((world * width, rows.shape[1]), dtype=torch.uint8, device=rows.device)
        dist.all_gather_into_tensor(out, padded, group=group)          # the job's single collective
    gathered = unpack_compositions(out.view(world, width, -1), n_atoms, d)
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
