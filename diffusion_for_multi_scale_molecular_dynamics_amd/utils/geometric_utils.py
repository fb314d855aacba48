"""src/.../utils/geometric_utils.py: the cubic point group as matrices."""
import itertools

import torch


def get_cubic_point_group_symmetries(spatial_dimension: int = 3) -> torch.Tensor:
    """The 2^d d! signed permutation matrices [.., d, d]: for every permutation of the axes (itertools order), every choice
    of signs (itertools.product(-1, 1) order), the matrix P S -- row r of P is the unit vector of axis perm[r], S is diagonal."""
    d = spatial_dimension
    out = torch.zeros(0, d, d)
    blocks = []
    for perm in itertools.permutations(range(d)):
        for signs in itertools.product((-1.0, 1.0), repeat=d):
            m = torch.zeros(d, d)
            for r in range(d):
                m[r, perm[r]] = signs[perm[r]]
            blocks.append(m)
    return torch.stack(blocks) if blocks else out
