"""Trajectory recorder with the reference's on-disk format (src/.../utils/sample_trajectory.py:7-44)."""
from collections import defaultdict
from typing import Any, Dict, NamedTuple, Union

import torch


def _compact(obj):
    if isinstance(obj, torch.Tensor):
        return obj.clone() if obj.untyped_storage().nbytes() > obj.numel() * obj.element_size() else obj
    if isinstance(obj, tuple) and hasattr(obj, "_fields"):
        return type(obj)(*[_compact(v) for v in obj])
    if isinstance(obj, dict):
        return {k: _compact(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_compact(v) for v in obj)
    return obj


class PinnedStaging:
    """Host side of the recorder's device-to-host copies: page-locked memory taken in large chunks, so that recording a
    step is a handful of ASYNCHRONOUS copies on the sampling stream (no host synchronisation per step, which is what
    `.cpu()` costs); the copies are complete once the stream has been synchronised -- LangevinGenerator.sample() does
    that when it reads the status word.  The recorded tensors are views into the chunks and keep them alive."""

    CHUNK_BYTES = 32 << 20

    def __init__(self):
        self._chunk = None
        self._used = 0

    def to_host(self, t: torch.Tensor) -> torch.Tensor:
        t = t.detach()
        if not t.is_cuda:
            return t.clone()
        n = t.numel() * t.element_size()
        if n == 0:
            return torch.empty(t.shape, dtype=t.dtype)
        start = (self._used + 63) & ~63
        if self._chunk is None or start + n > self._chunk.numel():
            self._chunk = torch.empty(max(self.CHUNK_BYTES, n), dtype=torch.uint8, pin_memory=True)
            start = 0
        self._used = start + n
        host = self._chunk[start:start + n].view(t.dtype).view(t.shape)
        host.copy_(t if t.is_contiguous() else t.contiguous(), non_blocking=True)
        return host


class SampleTrajectory:
    def __init__(self):
        self._internal_data = defaultdict(list)

    def reset(self):
        self._internal_data = defaultdict(list)

    def record(self, key: str, entry: Union[Dict[str, Any], NamedTuple]):
        self._internal_data[key].append(entry)

    def write_to_pickle(self, path_to_pickle: str, for_reference: bool = False):
        """for_reference: name the reference's AXL class in the file (utils/reference_pickles.save_for_reference)."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()          # the recorder's asynchronous device-to-host copies (PinnedStaging)
        data = dict(self._internal_data)
        for key, value in data.items():
            if isinstance(value, list) and len(value) == 1:
                data[key] = value[0]
        data = _compact(data)                 # views into the staging chunks -> tensors that own exactly their bytes
        self._internal_data = data
        if for_reference:
            from . import reference_pickles
            reference_pickles.save_for_reference(data, path_to_pickle)
            return
        with open(path_to_pickle, "wb") as fd:
            torch.save(data, fd)

    def entries(self) -> Dict[str, list]:
        """The recorded lists as they are (no unwrapping of single entries), owning their bytes: what one rank contributes
        to the run's trajectory file."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        return _compact({key: list(value) if isinstance(value, list) else [value]
                         for key, value in self._internal_data.items()})


# keys that grow by a fixed number of entries per generator.sample() call (one sub-batch); everything else is a description
# of the run, recorded once and equal on every rank
PER_CALL_KEYS = ("predictor_step", "corrector_step", "atom_type_update")


def merge_sharded_entries(per_rank: list, sizes: list, world_size: int) -> SampleTrajectory:
    """One recorder for a run whose sub-batches were dealt round-robin to `world_size` ranks (sampling/diffusion_sampling.py):
    the per-call entries of every rank, put back in SUB-BATCH order -- what a single process looping over the sub-batches
    (the reference: src/.../sampling/diffusion_sampling.py:44-50 with src/.../sample_diffusion.py:253-257) would have recorded."""
    merged = SampleTrajectory()
    for key, value in per_rank[0].items():
        if key not in PER_CALL_KEYS:
            merged._internal_data[key] = list(value)
    owners = [[k for k in range(len(sizes)) if k % world_size == r] for r in range(world_size)]
    for key in PER_CALL_KEYS:
        if not any(key in data for data in per_rank):
            continue
        chunks = {}
        for r, data in enumerate(per_rank):
            entries, calls = data.get(key, []), len(owners[r])
            if calls == 0:
                assert not entries
                continue
            assert len(entries) % calls == 0, f"rank {r}: {len(entries)} '{key}' entries for {calls} sample() calls"
            per_call = len(entries) // calls
            for c, k in enumerate(owners[r]):
                chunks[k] = entries[c * per_call:(c + 1) * per_call]
        for k in range(len(sizes)):
            merged._internal_data[key].extend(chunks.get(k, []))
    return merged
