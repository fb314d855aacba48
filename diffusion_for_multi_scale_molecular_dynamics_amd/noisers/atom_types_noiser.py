"""D3PM forward noising of atom types (HIP kernel F2) -- src/.../noisers/atom_types_noiser.py:30-60."""
from typing import Tuple

import torch

from .. import kernels


class AtomTypesNoiser:
    @staticmethod
    def _get_uniform_noise(shape: Tuple[int]) -> torch.Tensor:
        return torch.rand(shape)      # CPU generator, like the reference; NOT clipped (reference quirk)

    @staticmethod
    def get_noisy_atom_types_sample(real_onehot_atom_types: torch.Tensor, q_bar: torch.Tensor) -> torch.Tensor:
        """a_t = argmax_c(log(Qbar[a_0][c]) - log(-log u_c)).  Two operand forms:
          * the reference's (:30-60): real_onehot_atom_types one-hot [..., C] and q_bar [..., C, C] with the same leading dimensions
            (the assertion of :42-44); a one-hot row times q_bar is the row a_0 of q_bar, every other term an exact zero;
          * class indices [...] and ONE [C, C] matrix for the call (the sampler noises a batch to a single time index)."""
        real_atom_types = real_onehot_atom_types          # (the parameter carries the reference's name; both forms come through it)
        num_classes = q_bar.shape[-1]
        if q_bar.dim() > 2:
            assert real_atom_types.shape == q_bar.shape[:-1], "q_bar array first dimensions should match real_atom_types array"
            u = AtomTypesNoiser._get_uniform_noise(tuple(real_atom_types.shape)).to(q_bar)
            assert bool(((real_atom_types == 0) | (real_atom_types == 1)).all()) and \
                bool((real_atom_types.sum(dim=-1) == 1).all()), "real_onehot_atom_types must be one-hot vectors"
            indices = real_atom_types.argmax(dim=-1)
            return kernels.noise_atom_types(indices.contiguous(), q_bar.to(torch.float32).contiguous(), u.contiguous())
        u = AtomTypesNoiser._get_uniform_noise(tuple(real_atom_types.shape) + (num_classes,)).to(q_bar)
        return kernels.noise_atom_types(real_atom_types.contiguous(), q_bar.contiguous(), u.contiguous())
