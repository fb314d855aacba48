"""D3PM forward noising of atom types (HIP kernel F2) -- src/.../noisers/atom_types_noiser.py:30-60."""
from typing import Tuple

import torch

from .. import kernels


class AtomTypesNoiser:
    @staticmethod
    def _get_uniform_noise(shape: Tuple[int]) -> torch.Tensor:
        return torch.rand(shape)      # CPU generator, like the reference; NOT clipped (reference quirk)

    @staticmethod
    def get_noisy_atom_types_sample(real_atom_types: torch.Tensor, q_bar: torch.Tensor) -> torch.Tensor:
        """a_t = argmax_c(log(Qbar[a_0][c]) - log(-log u_c)); real_atom_types are class indices [.., N],
        q_bar is the [C, C] cumulative transition matrix of the (single) time index."""
        num_classes = q_bar.shape[-1]
        u = AtomTypesNoiser._get_uniform_noise(tuple(real_atom_types.shape) + (num_classes,)).to(q_bar)
        return kernels.noise_atom_types(real_atom_types.contiguous(), q_bar.contiguous(), u.contiguous())
