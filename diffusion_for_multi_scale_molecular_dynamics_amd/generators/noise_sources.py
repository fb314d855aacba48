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


class one_host_thread:
    """The parity mode's host-side tensor work -- a few draws of <= 10^5 numbers per step and the Gumbel transform -- on ONE
    thread.  torch sizes its intra-op pool by the machine's cores (128 on an MI355X host) whatever share of them the process
    may use; every small CPU op then wakes that pool, and its spinning workers starve the thread that launches kernels: measured
    at C3, 40.0 ms per iteration with the default pool, 34.5 with 8 threads, 32.8 with one (device-RNG mode: 31.3;
    profiles/r05_reference_mode_rate.json).  The numbers drawn do not depend on the thread count (checked by
    tests/test_host_cpu.py::test_host_draws_do_not_depend_on_the_thread_count).  The caller's setting is restored on exit."""

    def __enter__(self):
        self.previous = torch.get_num_threads()
        if self.previous != 1:
            torch.set_num_threads(1)

    def __exit__(self, *exc):
        if self.previous != 1:
            torch.set_num_threads(self.previous)
        return False


def upload(draws, device):
    """Host draws -> float32 on the device: a plain (blocking) copy from pageable memory.  The host then moves in step with the
    GPU -- one wait per sub-step, 1.5 ms per C3 iteration against the device-RNG modes -- which is the fast way here: both
    asynchronous forms were measured and dropped (a page-locked block per upload from torch's caching host allocator: 62.7 ms
    per iteration; a ring of page-locked buffers allocated once, copies guarded by events: 108 ms -- the host running ahead of
    eager launches does not suit this loop; `profiles/r05_round_notes.md`).  None stays None."""
    if draws is None:
        return None
    return draws.to(device=device, dtype=torch.float32).contiguous()


class ReferenceOrderNoise:
    device_rng = False

    def rand(self, *shape) -> torch.Tensor:
        with one_host_thread():
            return torch.rand(*shape)

    def randn(self, *shape) -> torch.Tensor:
        with one_host_thread():
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


class RecordingNoise(ReferenceOrderNoise):
    """Reference-order draws passed through and KEPT, so that an ITERATION which has to be recomputed (the score network's
    split-f16 kernels met a value beyond the f16 range: LangevinGenerator._guarded_iteration) sees the very same numbers
    again: a CPU generator cannot be rewound, a list can -- and it works for any source a caller plugs in (a replayed fixture,
    a generator of its own), which saving torch's global RNG state would not.  One iteration's draws are held at a time
    ((1 + M) steps), not the trajectory's.  replay() returns a source that hands the kept draws out again, in order, and
    continues with `inner` once they are used up."""

    def __init__(self, inner):
        self.inner = inner
        self.kept = []

    def rand(self, *shape):
        out = self.inner.rand(*shape)
        self.kept.append(("rand", out))
        return out

    def randn(self, *shape):
        out = self.inner.randn(*shape)
        self.kept.append(("randn", out))
        return out

    def replay(self):
        return _ReplayNoise(self.kept, self.inner)


class _ReplayNoise(ReferenceOrderNoise):
    def __init__(self, kept, inner):
        self.kept, self.inner, self.position = kept, inner, 0

    def _next(self, kind, shape):
        if self.position < len(self.kept):
            want, out = self.kept[self.position]
            self.position += 1
            assert want == kind, "the recomputed call draws in a different order"
            return out
        return getattr(self.inner, kind)(*shape)

    def rand(self, *shape):
        return self._next("rand", shape)

    def randn(self, *shape):
        return self._next("randn", shape)
