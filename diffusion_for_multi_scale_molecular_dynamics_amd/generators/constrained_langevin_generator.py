"""RePaint-style constrained sampling (src/.../generators/constrained_langevin_generator.py:24-182).

After every predictor step the constrained rows are replaced by the known atoms forward-noised to the current
time index -- kernels F1 + F2 fused with the scatter (mdx_repaint_constrained_rows).  In reference-RNG mode the
reference's full-size draws (including the throw-away random composition) are reproduced draw for draw.
"""
from typing import Optional

import torch

from .. import kernels
from .._hip import Rng
from ..models.score_networks.score_network import ScoreNetwork
from ..namespace import AXL
from ..noise_schedulers.noise_parameters import NoiseParameters
from .langevin_generator import LangevinGenerator
from .noise_sources import upload
from .predictor_corrector_axl_generator import PredictorCorrectorSamplingParameters
from .sampling_constraint import SamplingConstraint
from .trajectory_initializer import TrajectoryInitializer


class ConstrainedLangevinGenerator(LangevinGenerator):
    def __init__(self, noise_parameters: NoiseParameters, sampling_parameters: PredictorCorrectorSamplingParameters,
                 axl_network: ScoreNetwork, sampling_constraints: SamplingConstraint,
                 trajectory_initializer: Optional[TrajectoryInitializer] = None):
        super().__init__(noise_parameters=noise_parameters, sampling_parameters=sampling_parameters,
                         axl_network=axl_network, trajectory_initializer=trajectory_initializer)
        self.sampling_constraints = sampling_constraints
        number_of_constraints, spatial_dimension = sampling_constraints.constrained_relative_coordinates.shape
        assert len(sampling_constraints.elements) == sampling_parameters.num_atom_types, \
            "Inconsistent number of atom types vs. elements list"
        assert number_of_constraints <= self.number_of_atoms, "There are more constrained positions than atoms!"
        assert spatial_dimension <= self.spatial_dimension, \
            "The spatial dimension of the constrained relative coordinates is inconsistent"
        if sampling_constraints.constrained_indices is None:
            self.constraint_indices = torch.arange(number_of_constraints)
        else:
            self.constraint_indices = sampling_constraints.constrained_indices
        self._constraint_device = {}
        self.resampling_steps = int(getattr(sampling_parameters, "repaint_resampling_steps", 0) or 0)
        assert self.resampling_steps >= 0, "repaint_resampling_steps should be non-negative"

    def _constraint_on(self, device):
        if device not in self._constraint_device:
            c = self.sampling_constraints
            self._constraint_device[device] = (
                c.constrained_relative_coordinates.to(device=device, dtype=torch.float32).contiguous(),
                c.constrained_atom_types.to(device=device, dtype=torch.int64).contiguous(),
                self.constraint_indices.to(device=device, dtype=torch.int64).contiguous())
        return self._constraint_device[device]

    def _repaint(self, composition: AXL, index_i: int, d_index=None, draw_index_offset: int = 1) -> AXL:
        """In-place on composition.X / composition.A, like the reference (:159-160)."""
        x, a = composition.X, composition.A
        device = x.device
        batch = x.shape[0]
        sched = self._prepare(device)
        cx, ca, cidx = self._constraint_on(device)
        z = u = None
        if not getattr(self.noise_source, "device_rng", False):
            self.initialize(batch, device)                     # composition_0_known: drawn, only constrained rows kept
            if d_index is None and index_i > 0:                # noising_transform.py:154,179
                z = upload(self.noise_source.randn(x.shape), device)
                u = upload(self.noise_source.rand(batch, self.number_of_atoms, self.num_classes), device)
        # Philox draw id of the repaint noise: that of the predictor step it follows (index_i + 1)
        rng = self._rng(0)
        rng_index = index_i
        kernels.repaint_constrained_rows(sched, rng_index, d_index, cx, ca, cidx, z, u,
                                         Rng(rng.seed, rng.call, rng.draw_stride, rng.draw_stride + rng.draw_offset, 0, rng.call_dev),
                                         x, a)
        return AXL(A=a, X=x, L=composition.L)

    def _forward_step(self, composition: AXL, index_i: int, d_index=None) -> AXL:
        """Resampling: forward-noise the WHOLE composition from time index i back to i+1, in place
        (mdx_forward_diffusion_step; build-only, no reference counterpart).  Reference-RNG mode draws z then u."""
        x, a = composition.X, composition.A
        device = x.device
        z = u = None
        if not getattr(self.noise_source, "device_rng", False):
            z = upload(self.noise_source.randn(x.shape), device)
            u = upload(self.noise_source.rand(x.shape[0], self.number_of_atoms, self.num_classes), device)
        kernels.forward_diffusion_step(self._prepare(device), index_i, d_index, z, u, self._rng(0), x, a)
        return AXL(A=a, X=x, L=composition.L)

    def predictor_step(self, composition_i: AXL, index_i: int, cartesian_forces: torch.Tensor) -> AXL:
        raw = super().predictor_step(composition_i, index_i, cartesian_forces)
        return self._repaint(raw, index_i - 1)

    def _after_predictor(self, composition: AXL, index_i: int, d_index=None) -> AXL:
        return self._repaint(composition, index_i, d_index=d_index)

    def _apply_constraint(self, composition: AXL, device: torch.device) -> AXL:
        """Hard constraint (:74-82): the index-0 path of the repaint kernel copies the known rows unnoised."""
        x, a = composition.X, composition.A
        cx, ca, cidx = self._constraint_on(x.device)
        kernels.repaint_constrained_rows(self._prepare(x.device), 0, None, cx, ca, cidx, None, None,
                                         Rng(0, 0, 1, 0), x, a)
        return AXL(A=a, X=x, L=composition.L)

    def sample(self, number_of_samples: int, device: torch.device) -> AXL:
        composition = super().sample(number_of_samples=number_of_samples, device=device)
        return self._apply_constraint(composition, device)
