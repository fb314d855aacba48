"""Where the sampler's random numbers come from.

ReferenceOrderNoise  parity mode: every draw is made with torch's CPU default generator, in exactly the order,
                     shapes and (for Gumbel) CPU arithmetic of the reference (SURVEY.md section 8a, row RNG;
                     src/.../generators/langevin_generator.py:92-111), then uploaded.  With the same
                     torch.manual_seed this reproduces the reference CPU generator's trajectory.
DevicePhiloxNoise    throughput mode: nothing is drawn on the host; kernels evaluate the counter-based
                     Philox4x32-10 specification of DESIGN.md in registers.
Both expose `device_rng`; generators pass NULL noise pointers to the kernels when it is True.
"""
import torch

from .. import kernels
from .._hip import TAG_INIT, TAG_INIT_LATTICE


class ReferenceOrderNoise:
    device_rng = False

    def rand(self, *shape) -> torch.Tensor:
        return torch.rand(*shape)

    def randn(self, *shape) -> torch.Tensor:
        return torch.randn(*shape)

    def initial_coordinates(self, b, n, d, device):
        return self.rand(b, n, d).to(device)

    def initial_lattice(self, b, nl, device):
        return self.randn(b, nl).to(device)


class DevicePhiloxNoise:
    device_rng = True

    def __init__(self, seed: int, call: int = 0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.call = int(call)

    def initial_coordinates(self, b, n, d, device):
        return kernels.rng_fill(kernels.RNG_UNIFORM, self.seed, self.call, 0, TAG_INIT, b * n, d, device).view(b, n, d)

    def initial_lattice(self, b, nl, device):
        return kernels.rng_fill(kernels.RNG_NORMAL, self.seed, self.call, 0, TAG_INIT_LATTICE, b, nl, device)
