"""src/.../models/egnn_utils.py under the reference's module path: segment reductions over an UNSORTED index (the plain
PyTorch form E_GCL's public sub-methods use; the sampler's path reduces sorted segments inside the kernels) and the edge-list
builders (implementation: utils/neighbors.py, on the HIP radius graph)."""
from typing import List

import torch

from ..utils.neighbors import get_edges_batch, get_edges_with_radial_cutoff  # noqa: F401


def unsorted_segment_sum(data: torch.Tensor, segment_ids: torch.Tensor, num_segments: int) -> torch.Tensor:
    """out[s] = sum of the rows of `data` whose id is s  ([num_segments, width]; :11-38)."""
    out = torch.zeros(num_segments, data.size(1), dtype=data.dtype, device=data.device)
    return out.index_add_(0, segment_ids, data)


def unsorted_segment_mean(data: torch.Tensor, segment_ids: torch.Tensor, num_segments: int) -> torch.Tensor:
    """The same divided by the number of rows per segment, empty segments counted as 1 (:41-70)."""
    count = torch.zeros(num_segments, 1, dtype=data.dtype, device=data.device)
    count.index_add_(0, segment_ids, torch.ones(data.size(0), 1, dtype=data.dtype, device=data.device))
    return unsorted_segment_sum(data, segment_ids, num_segments) / count.clamp(min=1)


def get_edges(n_nodes: int) -> List[List[int]]:
    """The n (n - 1) [source, destination] pairs of the fully connected graph without self-loops, source-major (:73-82)."""
    return [[source, destination] for source in range(n_nodes) for destination in range(n_nodes) if source != destination]
