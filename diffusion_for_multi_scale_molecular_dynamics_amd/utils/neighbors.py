"""Periodic neighbour search on the GPU (HIP kernel N1 behind the reference's function names).

Replaces the KeOps-based implementation of src/.../utils/neighbors.py:36-224 and its consumers
get_adj_matrix (models/graph_utils.py:10-50) / get_edges_with_radial_cutoff (models/egnn_utils.py:107-144).
Edge ORDER differs from the reference by design -- (structure, source, destination, image) instead of
(structure, image, source, k-th nearest) -- both consumers are order-insensitive, and the sorted order is what
lets the EGNN aggregate without atomics.  The edge SET, shifts and counts are identical.
"""
from collections import namedtuple
from typing import Optional

import torch

from .. import kernels
from .._hip import (STATUS_CUTOFF_TOO_LARGE, STATUS_EGNN_F16_RANGE, STATUS_GRAPH_CAPACITY, EdgeChainRangeError,
                     MdxError)

AdjacencyInfo = namedtuple(
    "AdjacencyInfo",
    ["adjacency_matrix", "shifts", "edge_batch_indices", "node_batch_indices", "number_of_edges"],
)


def _new_status(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


def _raise_if_cutoff_too_large(status: torch.Tensor):
    word = int(status.item())
    if word & STATUS_CUTOFF_TOO_LARGE:
        # the reference asserts here (neighbors.py:107-113)
        raise AssertionError("The radial cutoff is so large that neighbors could be located "
                             "beyond the first shell of periodic unit cell images.")
    if word & STATUS_EGNN_F16_RANGE:
        raise EdgeChainRangeError("EGNN edge chain, split-f16 mode: an activation left the f16 range (|x| > 6.5e4); the "
                                  "results of this call are invalid -- run the network with edge_chain_precision='f32'")
    if word & STATUS_GRAPH_CAPACITY:
        raise MdxError("radius graph: the edge list outgrew its capacity and the retry did not run (internal error)")


def embed_in_three_dimensions(cartesian_positions: torch.Tensor, basis_vectors: torch.Tensor, radial_cutoff: float):
    """(positions [B, N, 3], cell [B, 3, 3]) of a 1- or 2-dimensional periodic problem for the three-dimensional radius-graph
    kernel: the missing coordinates are zero and the missing cell vectors are orthogonal ones of length 4 x cutoff, so every
    periodic image along them is beyond the cutoff and the shortest cell-crossing distance (the "cutoff too large" check:
    neighbors.py:107-113, 248-351) is decided by the real vectors -- the same edges, shifts (their first d components) and
    check as the reference's d-dimensional search (neighbors.py:36-224 takes spatial_dimension in {1, 2, 3}).  Three-dimensional
    inputs pass through."""
    batch_size, natom, d = cartesian_positions.shape
    if d == 3:
        return cartesian_positions.contiguous(), basis_vectors.contiguous()
    positions = torch.zeros(batch_size, natom, 3, dtype=cartesian_positions.dtype, device=cartesian_positions.device)
    positions[..., :d] = cartesian_positions
    cell = torch.zeros(batch_size, 3, 3, dtype=basis_vectors.dtype, device=basis_vectors.device)
    cell[:, :d, :d] = basis_vectors
    for k in range(d, 3):
        cell[:, k, k] = 4.0 * float(radial_cutoff)
    return positions, cell


def get_periodic_adjacency_information(cartesian_positions: torch.Tensor, basis_vectors: torch.Tensor,
                                       radial_cutoff: float, spatial_dimension: int = 3,
                                       check_cutoff: bool = True) -> AdjacencyInfo:
    """All (src, dst, image) edges with 0 < |p_src - (p_dst + image)|^2 <= rc^2  (neighbors.py:36-224)."""
    assert cartesian_positions.dim() == 3, "Wrong number of dimensions for relative_coordinates"
    assert basis_vectors.dim() == 3, "Wrong number of dimensions for basis_vectors"
    batch_size, natom, d = cartesian_positions.shape
    assert d == spatial_dimension and d in (1, 2, 3), "The spatial dimension must be 1, 2 or 3."
    assert basis_vectors.shape == (batch_size, d, d), "Wrong shape for basis vectors"
    assert radial_cutoff > 0.0, "The radial cutoff should be greater than zero"
    status = _new_status(cartesian_positions.device) if check_cutoff else None
    positions3, cell3 = embed_in_three_dimensions(cartesian_positions, basis_vectors, radial_cutoff)
    out = kernels.radius_graph(positions3, cell3, radial_cutoff, unique=False, status=status)
    if d < 3:
        out = dict(out, shifts=out["shifts"][:, :d].contiguous())
    if check_cutoff:
        _raise_if_cutoff_too_large(status)
    number_of_edges = out["counts"].sum(dim=1)
    dev = cartesian_positions.device
    batch_ids = torch.arange(batch_size, device=dev)
    return AdjacencyInfo(
        adjacency_matrix=out["edges"].t(),                  # [2, E] per-structure indices, like the reference
        shifts=out["shifts"],
        edge_batch_indices=torch.repeat_interleave(batch_ids, number_of_edges),
        node_batch_indices=torch.repeat_interleave(batch_ids, natom),
        number_of_edges=number_of_edges,
    )


def shift_adjacency_matrix_indices_for_graph_batching(adjacency_matrix: torch.Tensor, num_edges: torch.Tensor,
                                                      number_of_atoms: int) -> torch.Tensor:
    """neighbors.py:381-390"""
    shifts = torch.arange(len(num_edges), device=adjacency_matrix.device) * number_of_atoms
    return adjacency_matrix + torch.repeat_interleave(shifts, num_edges).repeat(2, 1)


def get_adj_matrix(positions: torch.Tensor, basis_vectors: torch.Tensor, radial_cutoff: float = 4.0,
                   spatial_dimension: int = 3):
    """models/graph_utils.py:10-50"""
    info = get_periodic_adjacency_information(positions, basis_vectors, radial_cutoff, spatial_dimension)
    adj = shift_adjacency_matrix_indices_for_graph_batching(info.adjacency_matrix, info.number_of_edges,
                                                            positions.shape[1])
    return adj, info.shifts, info.node_batch_indices, info.number_of_edges


def get_edges_with_radial_cutoff(relative_coordinates: torch.Tensor, unit_cell: torch.Tensor,
                                 radial_cutoff: float = 4.0, drop_duplicate_edges: bool = True,
                                 spatial_dimension: int = 3, status: Optional[torch.Tensor] = None,
                                 return_degree: bool = False):
    """[E, 2] int64 batch-global (src, dst) pairs sorted lexicographically (models/egnn_utils.py:107-144).

    With drop_duplicate_edges (the only mode the EGNN uses) one edge is emitted per pair whatever the image --
    the same set, in the same order, as torch.unique(adj, dim=1) in the reference.  `status`, when given,
    receives the cutoff-too-large bit instead of a synchronising assert."""
    cart = torch.matmul(relative_coordinates, unit_cell).contiguous()
    if drop_duplicate_edges:
        own = status is None
        st = _new_status(cart.device) if own else status
        cart3, cell3 = embed_in_three_dimensions(cart, unit_cell, radial_cutoff)        # (1-D / 2-D: see there)
        out = kernels.radius_graph(cart3, cell3, radial_cutoff, unique=True, status=st)
        if own:
            _raise_if_cutoff_too_large(st)
        if return_degree:
            return out["edges"], out["counts"].view(-1)
        return out["edges"]
    adj, _, _, _ = get_adj_matrix(cart, unit_cell, radial_cutoff, spatial_dimension)
    return adj.transpose(0, 1)


def get_edges_static(relative_coordinates: torch.Tensor, unit_cell: torch.Tensor, radial_cutoff: float, capacity: int,
                     status: Optional[torch.Tensor] = None):
    """get_edges_with_radial_cutoff without the host read of the edge count: a capacity-sized edge list
    (edges [capacity, 2], degree [B*N], offsets [B*N], n_edges int64 [1] on the device).  capacity = B N (N - 1) cannot
    overflow; a smaller one reports through `status` (STATUS_GRAPH_CAPACITY)."""
    cart = torch.matmul(relative_coordinates, unit_cell).contiguous()
    out = kernels.radius_graph_static(cart, unit_cell.contiguous(), radial_cutoff, capacity, status=status)
    return out["edges"], out["counts"], out["offsets"], out["n_edges"]


def get_edges_static_clipped(relative_coordinates: torch.Tensor, lattice_parameters: torch.Tensor, clip_min: float,
                             radial_cutoff: float, capacity: int, status: Optional[torch.Tensor] = None):
    """get_edges_static for the cell EGNNScoreNetwork searches in -- orthogonal, lengths lattice_parameters[:, :3] clipped from
    below (egnn_score_network.py:236-240) -- straight from the relative coordinates: one call, two launches (kernels.egnn_radius_graph)."""
    out = kernels.egnn_radius_graph(relative_coordinates.contiguous(), lattice_parameters.contiguous(), clip_min, radial_cutoff,
                                    capacity, status=status)
    return out["edges"], out["counts"], out["offsets"], out["n_edges"]


def get_edges_batch(n_nodes: int, batch_size: int, device=None) -> torch.Tensor:
    """Fully connected edges without self loops, [B n (n-1), 2], sorted by source (models/egnn_utils.py:73-104).  Index
    arithmetic only -- no boolean mask, whose output size is a host read: the list can be built while a stream is capturing."""
    if n_nodes < 2:
        return torch.zeros(0, 2, dtype=torch.int64, device=device)
    src = torch.arange(n_nodes, device=device).repeat_interleave(n_nodes - 1)
    k = torch.arange(n_nodes - 1, device=device).repeat(n_nodes)
    dst = k + (k >= src).to(k.dtype)                                    # the k-th node other than src, in increasing order
    pair = torch.stack([src, dst], dim=1)
    offsets = (torch.arange(batch_size, device=device) * n_nodes).view(-1, 1, 1)
    return (pair.unsqueeze(0) + offsets).reshape(-1, 2)
