"""Integer lattice vectors (src/.../utils/lattice_utils.py): the names the reference's callers import; the arithmetic that
matters on the hot path lives in the kernels (the 27 image vectors of N1 are formed in radius_graph_kernel in this order)."""
import itertools
from typing import List

import torch


def get_relative_coordinates_lattice_vectors(number_of_shells: int = 1, spatial_dimension: int = 3) -> torch.Tensor:
    """All integer vectors with components in -number_of_shells .. number_of_shells, as floats, in itertools.product order --
    (2 n + 1)^d rows: the periodic images the neighbour search sweeps (:10-29)."""
    steps = range(-number_of_shells, number_of_shells + 1)
    return torch.tensor(list(itertools.product(steps, repeat=spatial_dimension)), dtype=torch.float32)


def _sort_complete_shell(complete_shell: torch.Tensor) -> torch.Tensor:
    """The rows of a shell [members, d] in descending lexicographic order: the most positive leading components first (:32-63)."""
    rows = sorted((tuple(row) for row in complete_shell.tolist()), reverse=True)
    return torch.tensor(rows, dtype=complete_shell.dtype).reshape(complete_shell.shape)


def get_cubic_point_group_complete_lattice_shells(number_of_complete_shells: int, spatial_dimension: int = 3) -> List[torch.Tensor]:
    """The first complete shells of integer lattice vectors under the cubic point group, one int32 tensor [members, d] per
    shell (:66-126)."""
    from ..models.score_networks.egnn_score_network import complete_lattice_shells
    return [torch.tensor(shell, dtype=torch.int32) for shell in complete_lattice_shells(number_of_complete_shells, spatial_dimension)]


def get_cubic_point_group_positive_normalized_bloch_wave_vectors(number_of_complete_shells: int,
                                                                 spatial_dimension: int = 3) -> torch.Tensor:
    """One integer reciprocal-lattice vector per {K, -K} pair of the first complete shells of the cubic point group (:129-177):
    the wave vectors of the EGNN's torus uplift (models/score_networks/egnn_score_network.positive_bloch_wave_vectors)."""
    from ..models.score_networks.egnn_score_network import positive_bloch_wave_vectors
    return positive_bloch_wave_vectors(number_of_complete_shells, spatial_dimension).to(torch.int32)      # (integers, as there)
