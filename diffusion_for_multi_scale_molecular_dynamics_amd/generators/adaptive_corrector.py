"""Adaptive-step Langevin corrector (src/.../generators/adaptive_corrector.py:17-148).

The predictor only updates the atom types; each corrector step uses
    eps = 2 (r * mean|z| / (mean|sigma s| / sigma))^2
with batch means, so the step needs two reductions over the batch (torch, on the device) before the update kernel
(P1 with explicit scalars).  Not one of the BASELINE configurations.  Sharding a batch over ranks changes the batch
means (SURVEY 8e caveat): with `sync_batch_statistics: true` the two means are taken over every rank's shard through one
4-scalar all-reduce per corrector step (utils/batch_statistics.py), which reproduces the un-sharded step sizes.
"""
from typing import Optional

import torch

from .. import kernels
from .._hip import MDX_CORRECTOR, MDX_PREDICTOR, TAG_COORD, TAG_LATTICE, MdxError
from ..models.score_networks.score_network import ScoreNetwork
from ..namespace import AXL
from ..noise_schedulers.noise_parameters import NoiseParameters
from ..utils.batch_statistics import global_means
from .langevin_generator import LangevinGenerator
from .predictor_corrector_axl_generator import PredictorCorrectorSamplingParameters
from .trajectory_initializer import TrajectoryInitializer


class AdaptiveCorrectorGenerator(LangevinGenerator):
    def __init__(self, noise_parameters: NoiseParameters, sampling_parameters: PredictorCorrectorSamplingParameters,
                 axl_network: ScoreNetwork, trajectory_initializer: Optional[TrajectoryInitializer] = None):
        super().__init__(noise_parameters=noise_parameters, sampling_parameters=sampling_parameters,
                         axl_network=axl_network, trajectory_initializer=trajectory_initializer)
        self.corrector_r = noise_parameters.corrector_r
        self.sync_batch_statistics = bool(getattr(sampling_parameters, "sync_batch_statistics", False))
        if self.use_hip_graph or self.fused_score_network:
            raise MdxError("the adaptive corrector needs batch reductions between the forward and the update: "
                           "use_hip_graph / fused_score_network do not apply")

    def _normal(self, batch, index_i, offset, tag, n_items, width, device):
        """Device-RNG draws materialised (their norm is needed), identical to what the fused kernel would draw."""
        src = self.noise_source
        draw = index_i * (self.number_of_corrector_steps + 1) + offset
        return kernels.rng_fill(kernels.RNG_NORMAL, src.seed, src.call, draw, tag, n_items, width, device)

    def predictor_step(self, composition_i: AXL, index_i: int, cartesian_forces: torch.Tensor) -> AXL:
        """Atom types only; X and L pass through (:41-63).  The reference still draws z and z_lattice."""
        assert 1 <= index_i <= self.number_of_discretization_steps
        device = composition_i.X.device
        sched = self._prepare(device)
        batch = composition_i.X.shape[0]
        time_t, sigma_t = self._time_sigma(batch, device)
        kernels.fill_time_sigma(sched, MDX_PREDICTOR, index_i, None, time_t, sigma_t)
        predictions = self._get_model_predictions(composition_i, time_t, sigma_t, cartesian_forces)
        idx = index_i - 1
        device_rng = getattr(self.noise_source, "device_rng", False)
        if device_rng:
            draw = index_i * (self.number_of_corrector_steps + 1)
            src = self.noise_source
            from .._hip import TAG_BINARY, TAG_GUMBEL
            gumbel = kernels.rng_fill(kernels.RNG_GUMBEL, src.seed, src.call, draw, TAG_GUMBEL,
                                      batch * self.number_of_atoms, self.num_classes, device
                                      ).view(batch, self.number_of_atoms, self.num_classes)
            u = kernels.rng_fill(kernels.RNG_UNIFORM, src.seed, src.call, draw, TAG_BINARY,
                                 batch * self.number_of_atoms, 1, device).view(batch, self.number_of_atoms)
        else:
            gumbel = self._draw_gumbel_sample(batch).to(device).contiguous()
            u = self._draw_binary_sample(batch).to(device).contiguous() if self.atom_type_greedy_sampling else None
            self._draw_coordinates_gaussian_sample(batch)          # drawn by the reference, unused here
            self._draw_lattice_gaussian_sample(batch)
        one = self.one_atom_type_transition_per_step and idx != 0
        a_im1 = kernels.atom_types_update(predictions.A.contiguous(), composition_i.A.contiguous(), sched.q_matrix[idx],
                                          sched.q_bar_matrix[idx], sched.q_bar_tm1_matrix[idx], gumbel, u,
                                          self.small_epsilon, self.atom_type_greedy_sampling, one)
        if idx == 0:
            self._status |= ((a_im1 == self.masked_atom_type_index).any().to(torch.int32) * 2)
        out = AXL(A=a_im1, X=composition_i.X, L=composition_i.L)
        if self.record:
            self._record_step("predictor_step", ["composition_i", "composition_im1", "model_predictions_i"],
                              [composition_i, out, predictions], index_i)
        return out

    def _step_size(self, sigma, sigma_normalized_score, z, coordinates: bool) -> torch.Tensor:
        """eps_i (:97-148): norms over (atoms, space) per structure for the score, over the last axis for z."""
        dims = [-2, -1] if coordinates else -1
        score_mean, z_norm = global_means(torch.linalg.norm(sigma_normalized_score, dim=dims),
                                          torch.linalg.norm(z, dim=-1), self.sync_batch_statistics)
        score_norm = score_mean / sigma
        return 2 * (self.corrector_r * z_norm / score_norm.clip(min=self.small_epsilon)) ** 2

    def corrector_step(self, composition_i: AXL, index_i: int, cartesian_forces: torch.Tensor,
                       corrector_number: int = 0) -> AXL:
        assert 0 <= index_i <= self.number_of_discretization_steps - 1
        device = composition_i.X.device
        sched = self._prepare(device)
        x = composition_i.X
        batch, n, d = x.shape
        time_t, sigma_t = self._time_sigma(batch, device)
        kernels.fill_time_sigma(sched, MDX_CORRECTOR, index_i, None, time_t, sigma_t)
        predictions = self._get_model_predictions(composition_i, time_t, sigma_t, cartesian_forces)
        sigma = sigma_t[0, 0]
        device_rng = getattr(self.noise_source, "device_rng", False)
        if device_rng:
            z = self._normal(batch, index_i, 1 + corrector_number, TAG_COORD, batch * n, d, device).view(batch, n, d)
        else:
            z = self._draw_coordinates_gaussian_sample(batch).to(device).contiguous()
        eps = self._step_size(sigma, predictions.X, z, coordinates=True)
        # the step size is a batch statistic: it stays on the device ({eps, sqrt(2 eps), sigma} read by the kernel)
        x_out = kernels.relative_coordinates_update(x.contiguous(), predictions.X.contiguous(), z,
                                                    weights=torch.stack([eps, torch.sqrt(2 * eps), sigma]).float())
        lattice = composition_i.L
        if not device_rng:
            z_lattice = self._draw_lattice_gaussian_sample(batch).to(device)
        if not self.use_fixed_lattice_parameters:
            sigma_n = sigma / (n ** (1 / d))
            if device_rng:
                z_lattice = self._normal(batch, index_i, 1 + corrector_number, TAG_LATTICE, batch,
                                         self.num_lattice_parameters, device)
                z_used = z_lattice
            else:
                z_used = self._draw_lattice_gaussian_sample(batch).to(device).contiguous()   # the reference's 2nd draw
            eps_l = self._step_size(sigma_n, predictions.L, z_lattice, coordinates=False)
            lattice = kernels.lattice_parameters_update(lattice.contiguous(), predictions.L.contiguous(), z_used,
                                                        weights=torch.stack([eps_l, torch.sqrt(2 * eps_l), sigma_n]).float())
        out = AXL(A=composition_i.A, X=x_out, L=lattice)
        if self.record_corrector:
            self._record_step("corrector_step", ["composition_i", "corrected_composition_i", "model_predictions_i"],
                              [composition_i, out, predictions], index_i)
        return out
