"""Batch means that survive sharding (SURVEY 8e caveat).

AdaptiveCorrectorGenerator's step size uses means over the WHOLE batch of score and noise norms
(src/.../generators/adaptive_corrector.py:130-136), so a batch sharded over ranks needs one small all-reduce per
corrector step to reproduce the single-process result: the local sums and counts of both quantities travel in ONE
4-element all-reduce (RCCL on GPU tensors, gloo on CPU tensors).
"""
from typing import Tuple

import torch


def global_means(a: torch.Tensor, b: torch.Tensor, across_ranks: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """(mean(a), mean(b)) over the local tensors, or over the concatenation of every rank's tensors."""
    if not (across_ranks and torch.distributed.is_available() and torch.distributed.is_initialized()):
        return a.mean(), b.mean()
    packed = torch.stack([a.sum(dtype=torch.float64), b.sum(dtype=torch.float64),
                          torch.tensor(float(a.numel()), dtype=torch.float64, device=a.device),
                          torch.tensor(float(b.numel()), dtype=torch.float64, device=a.device)])
    torch.distributed.all_reduce(packed, op=torch.distributed.ReduceOp.SUM)
    return (packed[0] / packed[2]).to(a.dtype), (packed[1] / packed[3]).to(b.dtype)
