"""src/.../utils/symmetry_utils.py: the permutations of N atoms as index tensors."""
import itertools
import math
from typing import Tuple

import torch


def factorial(n):
    return math.factorial(n)


def get_all_permutation_indices(number_of_atoms) -> Tuple[torch.Tensor, torch.Tensor]:
    """([N!, N] every permutation of range(N) in itertools order, [N!, N] its inverse): x[:, p][:, p_inverse] == x (:15-36)."""
    permutations = torch.tensor(list(itertools.permutations(range(number_of_atoms))))
    return permutations, permutations.argsort(dim=1)
