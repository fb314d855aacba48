"""Network parameters from a counter-based formula (test infrastructure).

The fixtures of the production-size EGNN (4 graph layers x 256 wide x 4 hidden layers, 4.7 M parameters = 19 MB) do not
carry a state_dict: tests/golden/make_golden.py fills the REFERENCE's module with this formula before it records the
reference's outputs, and the tests fill the product's module with the same formula.  Every trainable parameter, in the
order of its state_dict key, element i (row-major):

    u(p, i) = splitmix64(0x9E3779B97F4A7C15 * (p + 1) + i) >> 40          -- 24 bits
    value   = bound_p * (u / 2^23 - 1)            in [-bound_p, bound_p), exactly representable in binary32 up to the product
    bound_p = 1 / sqrt(fan_in)     (nn.Linear's default range for weights and biases; fan_in = the weight's second dimension)

so the network has the statistics of a freshly initialised one (which is what BASELINE's configurations benchmark).
"""
import math

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def formula_values(p: int, numel: int, bound: float) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = np.uint64((0x9E3779B97F4A7C15 * (p + 1)) & 0xFFFFFFFFFFFFFFFF)
        counter = base + np.arange(numel, dtype=np.uint64)
    u = (_splitmix64(counter) >> np.uint64(40)).astype(np.float64)            # 24 bits
    return (np.float32(bound) * (u / 8388608.0 - 1.0).astype(np.float32)).astype(np.float32)


def fill_with_formula(module: torch.nn.Module, scale: float = 1.0) -> torch.nn.Module:
    """Overwrite every trainable parameter of `module` (sorted by name) with the formula; buffers and frozen parameters
    (the EGNN score network's reciprocal-lattice vectors and projection matrices) keep their constructed values.
    `scale` multiplies every bound (1.0 = nn.Linear's default range)."""
    named = sorted((name, prm) for name, prm in module.named_parameters() if prm.requires_grad)
    fan_in = {}
    for name, prm in named:
        if prm.dim() >= 2:
            fan_in[name.rsplit(".", 1)[0]] = prm.shape[1]
    with torch.no_grad():
        for p, (name, prm) in enumerate(named):
            fi = fan_in.get(name.rsplit(".", 1)[0], prm.shape[-1] if prm.dim() else 1)
            bound = scale / math.sqrt(max(fi, 1))
            values = formula_values(p, prm.numel(), bound).reshape(tuple(prm.shape))
            prm.copy_(torch.from_numpy(values))
    return module
