"""Annealed Langevin / predictor-corrector generator on the MI355X hot path.

Same class name, constructor, step methods and behaviour as the reference's LangevinGenerator
(src/.../generators/langevin_generator.py:27-831); what differs is where the work runs:

  * the schedule tables are built once on the device (kernel S1) and indexed there;
  * everything after the score-network forward -- D3PM posterior + greedy/one-transition atom-type draw (P2),
    coordinate update + periodic wrap (P1), lattice update (P3) -- is ONE fused kernel launch per step
    (mdx_pc_step_update), with per-structure reductions done by wavefront shuffles;
  * no per-step host synchronisation: the reference's asserts are collected in a device status word that is read
    once at the end of sample().  (A score network that runs split-f16 MFMA kernels also has its f16-range report
    watched per ITERATION: in the device-resident loop through asynchronous copies to page-locked memory read two
    iterations behind the queue, so the device never waits for the host -- IterationLoop._advance_watched; in the eager
    loop -- reference-order RNG, recording runs, callers without a hipGraph -- by ONE blocking 4-byte read per iteration
    (_take_range_report): that loop already uploads the iteration's draws from the host, it is the parity path, not the
    fast one);
  * rng_mode="device" removes the per-step CPU draws + PCIe uploads, and use_hip_graph=True replays one captured
    predictor+correctors iteration T times with the step index living on the device.

There is no CPU fallback: calling this generator with device="cpu" raises.
"""
import dataclasses
from typing import Optional

import torch

from .. import kernels
from .._hip import MDX_CORRECTOR, MDX_PREDICTOR, STATUS_MASK_AT_LAST_STEP, MdxError, PcFlags, Rng
from ..models.score_networks.score_network import ScoreNetwork
from ..namespace import AXL, CARTESIAN_FORCES, NOISE, NOISY_AXL_COMPOSITION, TIME
from ..noise_schedulers.noise_parameters import NoiseParameters
from ..noise_schedulers.noise_scheduler import NoiseScheduler
from ..utils.sample_trajectory import PinnedStaging, SampleTrajectory
from .noise_sources import DevicePhiloxNoise, RecordingNoise, ReferenceOrderNoise, one_host_thread, upload
from .predictor_corrector_axl_generator import PredictorCorrectorAXLGenerator, PredictorCorrectorSamplingParameters
from .trajectory_initializer import TrajectoryInitializer


class LangevinGenerator(PredictorCorrectorAXLGenerator):
    def __init__(self, noise_parameters: NoiseParameters, sampling_parameters: PredictorCorrectorSamplingParameters,
                 axl_network: ScoreNetwork, trajectory_initializer: Optional[TrajectoryInitializer] = None):
        sp = sampling_parameters
        super().__init__(number_of_discretization_steps=noise_parameters.total_time_steps,
                         number_of_corrector_steps=sp.number_of_corrector_steps,
                         spatial_dimension=sp.spatial_dimension, num_atom_types=sp.num_atom_types,
                         number_of_atoms=sp.number_of_atoms,
                         use_fixed_lattice_parameters=sp.use_fixed_lattice_parameters,
                         fixed_lattice_parameters=sp.fixed_lattice_parameters,
                         trajectory_initializer=trajectory_initializer)
        self.noise_parameters = noise_parameters
        self.sampling_parameters = sp
        self.number_of_atoms = sp.number_of_atoms
        self.masked_atom_type_index = self.num_classes - 1
        self.axl_network = axl_network
        self.small_epsilon = sp.small_epsilon
        self.one_atom_type_transition_per_step = sp.one_atom_type_transition_per_step
        self.atom_type_greedy_sampling = sp.atom_type_greedy_sampling
        self.atom_type_transition_in_corrector = sp.atom_type_transition_in_corrector
        self.use_fixed_lattice_parameters = sp.use_fixed_lattice_parameters
        self.fixed_lattice_parameters = sp.fixed_lattice_parameters

        self.record = sp.record_samples
        self.record_corrector = sp.record_samples_corrector_steps
        self.record_atom_type_update = sp.record_atom_type_update
        if self.record_corrector or self.record_atom_type_update:
            assert self.record, "Corrector steps or atom_type_update can only be recorded if record_samples is True."

        rng_mode = getattr(sp, "rng_mode", "reference")
        assert rng_mode in ("reference", "device"), f"unknown rng_mode {rng_mode}"
        self.rng_mode = rng_mode
        self.use_hip_graph = bool(getattr(sp, "use_hip_graph", False))
        self.fused_score_network = bool(getattr(sp, "fused_score_network", False))
        self._mlp_pack = None
        self._noise_workspace = kernels.NoiseWorkspace()     # pre-drawn records of the fused sampler, owned per generator
        self.fused_sampler_options = 0                       # MLP_SAMPLE_* bits of _hip.py (0 = the product path)
        self.fused_predrawn_noise = True                     # False: every wavefront draws in-kernel (same numbers)
        self._seed = getattr(sp, "seed", None)
        self._call_counter = 0
        self.noise_source = ReferenceOrderNoise() if rng_mode == "reference" else None
        # ITERATIONS recomputed with the exact-f32 MFMA kernels because the split-f16 ones met a value beyond the f16 range
        # (_guarded_iteration, IterationLoop._advance_watched); logged by sample_diffusion, printed by bench.py
        self.f16_range_fallbacks = 0
        self.resampling_steps = 0     # set by ConstrainedLangevinGenerator (repaint_resampling_steps)
        self._visit = 0               # which of the 1 + resampling_steps passes through the current time index

        self._scheduler = None       # device tables, built on first use for the sampling device
        self._status = None
        self._call_word = None       # int32 [1] on the device: the Philox call index of the current sample() call (_rng)
        self._buffers = {}
        if self.record:
            self.sample_trajectory_recorder = SampleTrajectory()
            self._staging = PinnedStaging()
            self.sample_trajectory_recorder.record(key="noise_parameters", entry=dataclasses.asdict(noise_parameters))
            self.sample_trajectory_recorder.record(key="sampling_parameters",
                                                   entry={k: v for k, v in dataclasses.asdict(sp).items()})

    # ---------------------------------------------------------------------------------------------------------
    # device state
    # ---------------------------------------------------------------------------------------------------------
    @staticmethod
    def _device(device) -> torch.device:
        """`cuda` -> `cuda:<current>`: torch.device("cuda") != torch.device("cuda:0"), and everything this generator keeps per
        device (tables, status and call words, the captured iteration) is compared by device."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        return device

    def _prepare(self, device: torch.device):
        device = self._device(device)
        if device.type != "cuda":
            raise MdxError(f"LangevinGenerator runs on the GPU hot path only; got device '{device}' "
                           "(there is no CPU fallback -- use the reference implementation on CPU)")
        if self._scheduler is None or self._scheduler.tables.device != device:
            self._scheduler = NoiseScheduler(self.noise_parameters, num_classes=self.num_classes, device=device)
            self.noise, self.langevin_dynamics = self._scheduler.get_all_sampling_parameters()
            self._status = torch.zeros(1, dtype=torch.int32, device=device)
            self._buffers = {}
            if self.record:
                self.sample_trajectory_recorder.record(
                    key="noise", entry=type(self.noise)(*[t.detach().cpu() for t in self.noise]))
        return self._scheduler.tables

    def _time_sigma(self, batch: int, device):
        key = ("ts", batch)
        if key not in self._buffers:
            self._buffers[key] = (torch.empty(batch, 1, dtype=torch.float32, device=device),
                                  torch.empty(batch, 1, dtype=torch.float32, device=device))
        return self._buffers[key]

    def _flags(self, update_atom_types: bool) -> PcFlags:
        return PcFlags(int(self.atom_type_greedy_sampling), int(self.one_atom_type_transition_per_step),
                       int(self.use_fixed_lattice_parameters), int(update_atom_types), float(self.small_epsilon))

    def _draw_stride(self) -> int:
        """Philox draw ids per time index: (1 predictor + M correctors) x (1 + resampling passes)."""
        return (self.number_of_corrector_steps + 1) * (self.resampling_steps + 1)

    def _rng(self, draw_offset: int) -> Rng:
        """The counter-based request of one launch.  The call index travels as a DEVICE word (`_call_word`, written by
        _begin_call) as well as by value: a launch captured into a hipGraph reads the word, so the same graph serves every
        later sample() call of that shape."""
        src = self.noise_source
        seed, call = (src.seed, src.call) if getattr(src, "device_rng", False) else (0, 0)
        word = self._call_word.data_ptr() if getattr(src, "device_rng", False) and self._call_word is not None else None
        return Rng(seed, call, self._draw_stride(),
                   self._visit * (self.number_of_corrector_steps + 1) + draw_offset, 0, word)

    def _begin_call(self, device):
        """One sample() call = one Philox `call` index; the trajectory initialiser shares the noise source."""
        if self.rng_mode == "device":
            seed = self._seed if self._seed is not None else torch.initial_seed()
            rank = torch.distributed.get_rank() if torch.distributed.is_available() and \
                torch.distributed.is_initialized() else 0
            self.noise_source = DevicePhiloxNoise(seed + rank, self._call_counter)
            device = self._device(device)
            if self._call_word is None or self._call_word.device != device:
                # (a captured iteration reads this word: it is replaced only together with the kept graph, whose key holds
                # its address)
                self._buffers.pop("graph_loop", None)
                self._call_word = torch.zeros(1, dtype=torch.int32, device=device)
            kernels.index_set(self._call_word, self._call_counter)
            self._call_counter += 1
        if hasattr(self.trajectory_initializer, "noise_source"):
            self.trajectory_initializer.noise_source = self.noise_source

    # ---------------------------------------------------------------------------------------------------------
    # reference-compatible draw hooks (CPU generator, reference order); tests may patch them like the reference's
    # ---------------------------------------------------------------------------------------------------------
    def _draw_coordinates_gaussian_sample(self, number_of_samples):
        return self.noise_source.randn(number_of_samples, self.number_of_atoms, self.spatial_dimension)

    def _draw_lattice_gaussian_sample(self, number_of_samples):
        return self.noise_source.randn(number_of_samples, self.num_lattice_parameters)

    def _draw_gumbel_sample(self, number_of_samples):
        u = self.noise_source.rand(number_of_samples, self.number_of_atoms, self.num_classes)
        if u.is_cuda:
            return -torch.log(-torch.log(u.clip(min=self.small_epsilon)))
        with one_host_thread():           # (host arithmetic of the parity mode: see noise_sources.one_host_thread)
            return -torch.log(-torch.log(u.clip(min=self.small_epsilon)))

    def _draw_binary_sample(self, number_of_samples):
        return self.noise_source.rand(number_of_samples, self.number_of_atoms)

    # ---------------------------------------------------------------------------------------------------------
    # the reference's private update methods, same names and operands (langevin_generator.py:155-534, 669-691): what its
    # unit tests and subclasses call.  Each is ONE stand-alone kernel (P1 / P2 / P3); the sampling loop itself goes through
    # the fused per-step kernel (_step), which computes the same expressions.
    # ---------------------------------------------------------------------------------------------------------
    @staticmethod
    def _three_scalars(score_weight, gaussian_noise_weight, sigma, like: torch.Tensor) -> torch.Tensor:
        """{score weight, noise weight, sigma} as float32 [3] on the device (no host read).  The reference broadcasts these
        against the whole batch; the sampler only ever passes one value each (g2_i, g_i, sigma_i of ONE time index)."""
        vals = []
        for name, v in (("score_weight", score_weight), ("gaussian_noise_weight", gaussian_noise_weight), ("sigma", sigma)):
            t = torch.as_tensor(v)
            if t.numel() != 1:
                raise MdxError(f"{name}: one value per call is supported (got shape {tuple(t.shape)}); the sampler's steps "
                               "are taken at a single time index")
            vals.append(t.reshape(()).to(device=like.device, dtype=torch.float32))
        return torch.stack(vals)

    def _relative_coordinates_update(self, relative_coordinates: torch.Tensor, sigma_normalized_scores: torch.Tensor,
                                     sigma_i: torch.Tensor, score_weight: torch.Tensor, gaussian_noise_weight: torch.Tensor,
                                     z: Optional[torch.Tensor]) -> torch.Tensor:
        """wrap(x + score_weight * s / sigma_i + gaussian_noise_weight * z)  (:155-201); z None: drawn here."""
        x = relative_coordinates
        if z is None:
            z = self._draw_coordinates_gaussian_sample(x.shape[0])
        weights = self._three_scalars(score_weight, gaussian_noise_weight, sigma_i, x)
        return kernels.relative_coordinates_update(x.contiguous(), sigma_normalized_scores.to(x).contiguous(),
                                                   z.to(x).contiguous(), weights=weights)

    _relative_coordinates_update_predictor_step = _relative_coordinates_update       # (:203-245: both return the generic one)
    _relative_coordinates_update_corrector_step = _relative_coordinates_update

    def _lattice_parameters_update(self, lattice_parameters: torch.Tensor, sigma_normalized_scores: torch.Tensor,
                                   sigma_n_i: torch.Tensor, score_weight: torch.Tensor, gaussian_noise_weight: torch.Tensor,
                                   z: Optional[torch.Tensor] = None) -> torch.Tensor:
        """l + score_weight * s / sigma_n_i + gaussian_noise_weight * z; the input itself when the lattice is fixed  (:441-490)."""
        if self.use_fixed_lattice_parameters:
            return lattice_parameters
        lat = lattice_parameters
        if z is None:
            z = self._draw_lattice_gaussian_sample(lat.shape[0])
        weights = self._three_scalars(score_weight, gaussian_noise_weight, sigma_n_i, lat)
        return kernels.lattice_parameters_update(lat.contiguous(), sigma_normalized_scores.to(lat).contiguous(),
                                                 z.to(lat).contiguous(), weights=weights)

    _lattice_parameters_update_predictor_step = _lattice_parameters_update           # (:492-534)
    _lattice_parameters_update_corrector_step = _lattice_parameters_update

    @staticmethod
    def _shared_matrix(m: torch.Tensor, name: str) -> torch.Tensor:
        """The one [C, C] matrix behind the reference's [number_of_samples, number_of_atoms, C, C] operand.  An expand() of one
        matrix (what the sampler builds) is recognised by its strides; a materialised copy is compared entry by entry (one host
        read, in this stand-alone method only); matrices that really differ per sample or atom are refused, not ignored."""
        from ..utils.d3pm_utils import _one_matrix
        one = _one_matrix(m)
        if one is None:
            flat = m.reshape(-1, m.shape[-2], m.shape[-1])
            one = flat[0]
            if not bool((flat == one).all()):
                raise MdxError(f"{name}: the transition matrices differ between samples / atoms; the sampler's atom-type update "
                               "takes ONE time index per call (use utils.d3pm_utils.get_probability_at_previous_time_step for "
                               "per-atom matrices)")
        return one

    def _atom_types_update(self, predicted_logits: torch.Tensor, atom_types_i: torch.LongTensor, q_matrices_i: torch.Tensor,
                           q_bar_matrices_i: torch.Tensor, q_bar_tm1_matrices_i: torch.Tensor,
                           atom_type_greedy_sampling: bool, one_atom_type_transition_per_step: bool) -> torch.LongTensor:
        """a_{i-1} ~ p(a_{i-1} | a_i, logits) with the Gumbel trick, greedy MASK handling and the one-transition rule
        (:247-337).  The three matrices: [C, C], or the reference's [number_of_samples, number_of_atoms, C, C] broadcast of
        one time index's matrix (its first entry is used)."""
        logits = predicted_logits
        batch = logits.shape[0]
        gumbel = self._draw_gumbel_sample(batch).to(device=logits.device, dtype=torch.float32).contiguous()
        u = None
        if atom_type_greedy_sampling:                           # drawn inside _adjust_..._for_greedy_sampling (:417)
            u = self._draw_binary_sample(batch).to(device=logits.device, dtype=torch.float32).contiguous()
        q, q_bar, q_bar_tm1 = [self._shared_matrix(m, name).to(logits).contiguous()
                               for m, name in ((q_matrices_i, "q_matrices_i"), (q_bar_matrices_i, "q_bar_matrices_i"),
                                               (q_bar_tm1_matrices_i, "q_bar_tm1_matrices_i"))]
        a_in = atom_types_i.to(logits.device).contiguous()
        a_out, probs = kernels.atom_types_update(logits.contiguous(), a_in, q, q_bar, q_bar_tm1, gumbel, u, self.small_epsilon,
                                                 atom_type_greedy_sampling, one_atom_type_transition_per_step,
                                                 return_probabilities=True)
        if self.record_atom_type_update:
            if atom_type_greedy_sampling:                       # the Gumbel rows of fully masked samples are kept (:419-437)
                all_masked = (a_in == self.masked_atom_type_index).all(dim=-1)
                gumbel = torch.where(all_masked.view(-1, 1, 1), gumbel, torch.zeros_like(gumbel))
            self.sample_trajectory_recorder.record(key="atom_type_update", entry=dict(
                predicted_logits=logits.detach().cpu(), one_step_transition_probabilities=probs.cpu(),
                gumbel_sample=gumbel.cpu(), a_i=a_in.cpu(), a_im1=a_out.cpu()))
        return a_out

    def _get_coordinates_corrector_step_size(self, index_i: int, sigma_i: torch.Tensor, model_predictions_i: torch.Tensor,
                                             z: torch.Tensor) -> torch.Tensor:
        """epsilon_i of the tabulated Langevin dynamics, indexed with 0..T-1  (:669-679)."""
        self._prepare(model_predictions_i.device)
        return self.langevin_dynamics.epsilon[index_i].to(model_predictions_i)

    def _get_lattice_parameters_corrector_step_size(self, index_i: int, sigma_n_i: torch.Tensor,
                                                    model_predictions_i: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
        """(:681-691)"""
        self._prepare(model_predictions_i.device)
        return self.langevin_dynamics.epsilon[index_i].to(model_predictions_i)

    # ---------------------------------------------------------------------------------------------------------
    # network
    # ---------------------------------------------------------------------------------------------------------
    def _get_model_predictions(self, composition: AXL, time, sigma_noise, cartesian_forces: torch.Tensor) -> AXL:
        """:113-153.  time / sigma_noise: the reference's floats (expanded to [B, 1] here, as there) or -- what the loop passes
        -- the [B, 1] device tensors the update kernels' schedule fill wrote, so that no scalar crosses the PCIe bus per step."""
        if not isinstance(time, torch.Tensor) or time.dim() == 0:
            time = torch.full((composition.X.shape[0], 1), float(time), dtype=torch.float32, device=composition.X.device)
        if not isinstance(sigma_noise, torch.Tensor) or sigma_noise.dim() == 0:
            sigma_noise = torch.full((composition.X.shape[0], 1), float(sigma_noise), dtype=torch.float32, device=composition.X.device)
        batch = {NOISY_AXL_COMPOSITION: composition, TIME: time, NOISE: sigma_noise, CARTESIAN_FORCES: cartesian_forces}
        return self.axl_network(batch, conditional=False)

    # ---------------------------------------------------------------------------------------------------------
    # one step = network forward + ONE fused update kernel
    # ---------------------------------------------------------------------------------------------------------
    def _step(self, mode: int, composition: AXL, index_i: int, cartesian_forces: torch.Tensor, draw_offset: int,
              d_index: Optional[torch.Tensor] = None, in_place: bool = False):
        x = composition.X
        device = x.device
        sched = self._prepare(device)
        batch = x.shape[0]
        time_t, sigma_t = self._time_sigma(batch, device)
        kernels.fill_time_sigma(sched, mode, index_i, d_index, time_t, sigma_t)
        predictions = self._get_model_predictions(composition, time_t, sigma_t, cartesian_forces)

        update_types = mode == MDX_PREDICTOR or self.atom_type_transition_in_corrector
        z = gumbel = u = z_lattice = None
        if not getattr(self.noise_source, "device_rng", False):
            if mode == MDX_PREDICTOR:                                  # langevin_generator.py:280,416,623,633
                gumbel = self._draw_gumbel_sample(batch)
                if self.atom_type_greedy_sampling:
                    u = self._draw_binary_sample(batch)
                z = self._draw_coordinates_gaussian_sample(batch)
                z_lattice = self._draw_lattice_gaussian_sample(batch)
            else:                                                      # :740,761,480-483,779-792
                z = self._draw_coordinates_gaussian_sample(batch)
                self._draw_lattice_gaussian_sample(batch)              # drawn by the reference, never used
                if not self.use_fixed_lattice_parameters:
                    z_lattice = self._draw_lattice_gaussian_sample(batch)
                if update_types:
                    gumbel = self._draw_gumbel_sample(batch)
                    if self.atom_type_greedy_sampling:
                        u = self._draw_binary_sample(batch)
            if self.use_fixed_lattice_parameters:
                z_lattice = None
            z, gumbel, u, z_lattice = [upload(t, device) for t in (z, gumbel, u, z_lattice)]
            if not self.use_fixed_lattice_parameters and z_lattice is None:
                raise MdxError("internal error: lattice noise missing")

        a_in = composition.A
        if update_types:
            a_out = a_in if in_place else torch.empty_like(a_in)
        else:
            a_out = a_in
        x_out = x if in_place else torch.empty_like(x)
        if self.use_fixed_lattice_parameters or in_place:
            l_out = composition.L
        else:
            l_out = torch.empty_like(composition.L)
        logits = predictions.A.contiguous() if update_types else None
        kernels.pc_step_update(sched, mode, index_i, d_index, self._flags(update_types),
                               a_in if update_types else None, x, composition.L, logits,
                               predictions.X.contiguous(),
                               None if self.use_fixed_lattice_parameters else predictions.L.contiguous(),
                               z, gumbel, u, z_lattice, self._rng(draw_offset),
                               a_out if update_types else None, x_out, l_out, self._status)
        if self.record_atom_type_update and update_types:
            self._record_atom_type_update(sched, mode, index_i, logits, a_in, gumbel, u, a_out, batch)
        return AXL(A=a_out, X=x_out, L=l_out), predictions

    def _record_atom_type_update(self, sched, mode, index_i, logits, a_in, gumbel, u, a_out, batch):
        """langevin_generator.py:325-335: logits, p(a_{t-1}|a_t) after the greedy adjustment, Gumbel values used."""
        idx = index_i - 1 if (mode == MDX_PREDICTOR or index_i > 0) else 0
        src = self.noise_source
        if gumbel is None:       # device RNG: materialise the same draws the fused kernel generated in registers
            from .._hip import TAG_BINARY, TAG_GUMBEL
            draw = index_i * self._draw_stride() + self._visit * (self.number_of_corrector_steps + 1) + \
                (0 if mode == MDX_PREDICTOR else 1)
            n = batch * self.number_of_atoms
            gumbel = kernels.rng_fill(kernels.RNG_GUMBEL, src.seed, src.call, draw, TAG_GUMBEL, n, self.num_classes,
                                      logits.device).view(batch, self.number_of_atoms, self.num_classes)
            u = kernels.rng_fill(kernels.RNG_UNIFORM, src.seed, src.call, draw, TAG_BINARY, n, 1,
                                 logits.device).view(batch, self.number_of_atoms)
        one = self.one_atom_type_transition_per_step and not (mode == MDX_PREDICTOR and idx == 0)
        a_check, probs = kernels.atom_types_update(logits, a_in, sched.q_matrix[idx], sched.q_bar_matrix[idx],
                                                   sched.q_bar_tm1_matrix[idx], gumbel, u, self.small_epsilon,
                                                   self.atom_type_greedy_sampling, one, return_probabilities=True)
        assert torch.equal(a_check, a_out), "stand-alone and fused atom-type updates disagree"
        if self.atom_type_greedy_sampling:
            all_masked = (a_in == self.masked_atom_type_index).all(dim=-1)
            gumbel = torch.where(all_masked.view(-1, 1, 1), gumbel, torch.zeros_like(gumbel))
        self.sample_trajectory_recorder.record(key="atom_type_update", entry=dict(
            predicted_logits=self._staging.to_host(logits), one_step_transition_probabilities=self._staging.to_host(probs),
            gumbel_sample=self._staging.to_host(gumbel), a_i=self._staging.to_host(a_in),
            a_im1=self._staging.to_host(a_out)))

    def predictor_step(self, composition_i: AXL, index_i: int, cartesian_forces: torch.Tensor) -> AXL:
        """composition at time index i -> i-1  (langevin_generator.py:536-645)."""
        assert 1 <= index_i <= self.number_of_discretization_steps, \
            "The predictor step can only be invoked for index_i between 1 and the total number of discretization steps."
        composition_im1, predictions = self._step(MDX_PREDICTOR, composition_i, index_i, cartesian_forces, 0)
        if self.record:
            self._record_step("predictor_step", ["composition_i", "composition_im1", "model_predictions_i"],
                              [composition_i, composition_im1, predictions], index_i)
        return composition_im1

    def corrector_step(self, composition_i: AXL, index_i: int, cartesian_forces: torch.Tensor,
                       corrector_number: int = 0) -> AXL:
        """Langevin corrector at time index i  (langevin_generator.py:693-805)."""
        assert 0 <= index_i <= self.number_of_discretization_steps - 1, \
            "The corrector step can only be invoked for index_i between 0 and the total number of " \
            "discretization steps minus 1."
        corrected, predictions = self._step(MDX_CORRECTOR, composition_i, index_i, cartesian_forces,
                                            1 + corrector_number)
        if self.record_corrector:
            self._record_step("corrector_step", ["composition_i", "corrected_composition_i", "model_predictions_i"],
                              [composition_i, corrected, predictions], index_i)
        return corrected

    def _record_step(self, key, names, axls, index_i):
        entry = dict(time_step_index=index_i)
        for name, axl in zip(names, axls):
            # asynchronous copies into page-locked memory, ordered on the sampling stream before any later in-place
            # update of these tensors; complete when sample() synchronises to read the status word
            host = self._staging.to_host
            entry[name] = AXL(A=host(axl.A), X=host(axl.X), L=host(axl.L))
        self.sample_trajectory_recorder.record(key=key, entry=entry)

    # ---------------------------------------------------------------------------------------------------------
    # loops
    # ---------------------------------------------------------------------------------------------------------
    def _after_predictor(self, composition: AXL, index_i: int, d_index=None) -> AXL:
        """Hook for the repaint generator."""
        return composition

    def sample_from_noisy_composition(self, starting_noisy_composition: AXL, starting_step_index: int,
                                      ending_step_index: int) -> AXL:
        """The entry point both sample() and callers with their own starting composition go through; ends with the call's one
        host read of the status words (the reference's asserts; the score network's f16-range report)."""
        self._check_index_range(starting_step_index, ending_step_index)
        self._prepare(starting_noisy_composition.X.device)
        if self.noise_source is None:
            self._begin_call(starting_noisy_composition.X.device)
        self._share_noise_source()       # (a caller's own noise_source also serves the initialiser: the repaint step draws from it)
        out = self._run_loop(starting_noisy_composition, starting_step_index, ending_step_index)
        self.check_status()
        return out

    # ---------------------------------------------------------------------------------------------------------
    # the f16 range of the score network's split-f16 MFMA kernels, handled per ITERATION
    # ---------------------------------------------------------------------------------------------------------
    SPLIT_F16_MODES = ("f16x3", "f16x3_32x32")

    def _range_guarded(self) -> bool:
        """Does the score network run split-f16 kernels that can report a value beyond the f16 range?"""
        return getattr(self.axl_network, "edge_chain_precision", None) in self.SPLIT_F16_MODES

    def _take_range_report(self) -> bool:
        """Read the network's status word (a host synchronisation) and clear the f16-range bit; other bits stay for
        check_status()."""
        from .._hip import STATUS_EGNN_F16_RANGE
        status = getattr(self.axl_network, "graph_status", None)
        if status is None:
            return False
        word = int(status.item())
        if word & STATUS_EGNN_F16_RANGE:
            status.bitwise_and_(~STATUS_EGNN_F16_RANGE)
            return True
        return False

    def _adapt_f16_range(self):
        """After the exact-f32 pass: the network derives per-layer activation exponents for its split-f16 kernels from the
        maxima that pass has collected (EGNNScoreNetwork.adapt_f16_range), so a layer that runs hot does not send every
        following iteration through the f32 kernels."""
        adapt = getattr(self.axl_network, "adapt_f16_range", None)
        if adapt is not None:
            adapt()

    def _begin_f16_fallback(self):
        """Before the exact-f32 pass of a fallback: the network forgets the activation maxima earlier f32 launches left (plain
        f32 use, other inputs), so the exponents adapt() derives afterwards describe THIS iteration."""
        begin = getattr(self.axl_network, "begin_f16_range_fallback", None)
        if begin is not None:
            begin()

    def _clear_stale_range_report(self):
        """A range bit left in the network's word by something that was not an iteration of this loop (the warm-up iterations
        before a capture, a caller stepping by hand) must not be read as the first iteration's report."""
        from .._hip import STATUS_EGNN_F16_RANGE
        status = getattr(self.axl_network, "graph_status", None)
        if status is not None and self._range_guarded():
            status.bitwise_and_(~STATUS_EGNN_F16_RANGE)

    def _count_fallback(self, index_i: int):
        import warnings
        self.f16_range_fallbacks += 1
        warnings.warn(f"EGNN edge chain: a value beyond the f16 range at time index {index_i}; this iteration is recomputed with "
                      "edge_chain_precision='f32' on the same draws (the network's setting is restored afterwards)")

    def _iteration(self, composition: AXL, i: int, forces: torch.Tensor) -> AXL:
        """Time index i + 1 -> i: predictor, M correctors, and the resampling passes of the repaint generator."""
        visits = self._visits_at(i)
        for self._visit in range(visits):
            composition = self.predictor_step(composition, i + 1, forces)
            for m in range(self.number_of_corrector_steps):
                composition = self.corrector_step(composition, i, forces, m)
            if self._visit < visits - 1:
                composition = self._forward_step(composition, i)
        self._visit = 0
        return composition

    def _guarded_iteration(self, composition: AXL, i: int, forces: torch.Tensor) -> AXL:
        """_iteration(); if the score network's split-f16 kernels report a value beyond the f16 range, THAT iteration -- and
        nothing else -- is recomputed from the same composition with the exact-f32 kernels on the same draws: the network's
        precision is put back afterwards, what the discarded attempt recorded is dropped, the event is counted in
        `f16_range_fallbacks` and warned about.  Device RNG: a draw is a function of (seed, call, index).  Reference-order RNG:
        the iteration's draws are kept while it runs and handed out again (one iteration's worth, not the trajectory's)."""
        if not self._range_guarded():
            return self._iteration(composition, i, forces)
        net, source = self.axl_network, self.noise_source
        keeps = not getattr(source, "device_rng", False)
        if keeps:
            self.noise_source = RecordingNoise(source)
            self._share_noise_source()
        marks = self._recorder_marks() if self.record else None
        try:
            out = self._iteration(composition, i, forces)
            if not self._take_range_report():
                return out
            self._count_fallback(i)
            if marks is not None:
                self._recorder_truncate(marks)
            if keeps:
                self.noise_source = self.noise_source.replay()
                self._share_noise_source()
            # what the dropped attempt raised in the generator's own word goes with it (non-finite logits at time index 0 leave
            # MASKs behind: MDX_STATUS_MASK_AT_LAST_STEP) -- the exact-f32 pass raises it again if it is real
            self._status.bitwise_and_(~STATUS_MASK_AT_LAST_STEP)
            precision = net.edge_chain_precision
            self._begin_f16_fallback()
            net.edge_chain_precision = "f32"
            try:
                return self._iteration(composition, i, forces)
            finally:
                net.edge_chain_precision = precision
                self._adapt_f16_range()
        finally:
            if keeps:
                self.noise_source = source
                self._share_noise_source()

    def _share_noise_source(self):
        if hasattr(self.trajectory_initializer, "noise_source"):
            self.trajectory_initializer.noise_source = self.noise_source

    def _recorder_marks(self):
        data = self.sample_trajectory_recorder._internal_data
        return {key: len(data[key]) for key in ("predictor_step", "corrector_step", "atom_type_update") if key in data}

    def _recorder_truncate(self, marks):
        data = self.sample_trajectory_recorder._internal_data
        for key in ("predictor_step", "corrector_step", "atom_type_update"):
            if key in data:
                del data[key][marks.get(key, 0):]

    def _run_loop(self, starting_noisy_composition: AXL, starting_step_index: int, ending_step_index: int) -> AXL:
        if self.fused_score_network:
            return self._sample_fused(starting_noisy_composition, starting_step_index, ending_step_index)
        if self.use_hip_graph and getattr(self.noise_source, "device_rng", False) and not self.record and \
                self._network_is_capture_safe(starting_noisy_composition):
            return self._sample_with_graph(starting_noisy_composition, starting_step_index, ending_step_index)
        composition = starting_noisy_composition
        forces = torch.zeros_like(composition.X)
        self._clear_stale_range_report()
        for i in range(starting_step_index - 1, max(ending_step_index, 0) - 1, -1):
            composition = self._guarded_iteration(composition, i, forces)
        return composition

    def _network_is_capture_safe(self, composition: AXL) -> bool:
        """A score network may say that its forward on this batch shape needs a host synchronisation (`capture_safe(batch,
        atoms, device)`: EGNNScoreNetwork with a radius graph whose layers do not all run the fused edge chain, or whose
        capacity-sized edge list does not fit) -- the iteration is then launched eagerly, with one warning, instead of failing
        inside the capture.  Networks without the method are taken at the caller's word (use_hip_graph=True)."""
        ask = getattr(self.axl_network, "capture_safe", None)
        if ask is None:
            return True
        safe = bool(ask(composition.X.shape[0], composition.X.shape[1], composition.X.device))
        if not safe and not getattr(self, "_warned_not_capturable", False):
            import warnings
            warnings.warn("use_hip_graph=True, but the score network's forward on this batch shape needs a host "
                          "synchronisation: the sampler iteration is launched eagerly")
            self._warned_not_capturable = True
        return safe

    def _visits_at(self, index_i: int) -> int:
        """Passes through time index i: 1 + resampling steps, none at the last index (RePaint algorithm 1, t > 1)."""
        return 1 + self.resampling_steps if index_i > 0 else 1

    def _forward_step(self, composition: AXL, index_i: int, d_index=None) -> AXL:
        raise MdxError("resampling needs the repaint generator")

    def _iteration_on_device_index(self, comp: AXL, forces: torch.Tensor, d_index: torch.Tensor,
                                   visits: Optional[int] = None):
        """One predictor + M correctors (x resampling passes) with the loop variable i read from *d_index on the
        device; in place."""
        visits = 1 + self.resampling_steps if visits is None else visits
        for self._visit in range(visits):
            comp, _ = self._step(MDX_PREDICTOR, comp, 1, forces, 0, d_index=d_index, in_place=True)
            comp = self._after_predictor(comp, 0, d_index=d_index)
            for m in range(self.number_of_corrector_steps):
                comp, _ = self._step(MDX_CORRECTOR, comp, 0, forces, 1 + m, d_index=d_index, in_place=True)
            if self._visit < visits - 1:
                comp = self._forward_step(comp, 0, d_index=d_index)
        self._visit = 0
        kernels.index_add(d_index, -1)
        return comp

    def fused_pack(self, device):
        """Device copy of the MLP's parameters for the fused kernels (loud errors when the path does not apply)."""
        from ..models.score_networks.mlp_score_network import MLPScoreNetwork
        if not isinstance(self.axl_network, MLPScoreNetwork):
            raise MdxError("fused_score_network=True needs an MLPScoreNetwork; other networks use the per-step kernels")
        if not getattr(self.noise_source, "device_rng", False) or self.record or type(self)._after_predictor is not \
                LangevinGenerator._after_predictor:
            raise MdxError("fused_score_network=True needs rng_mode='device', no recording and no repaint constraint")
        if self._mlp_pack is None or self._mlp_pack.device != torch.device(device):
            self._mlp_pack = kernels.MlpPack(self.axl_network, device)
        return self._mlp_pack

    def _sample_fused(self, start: AXL, starting_step_index: int, ending_step_index: int) -> AXL:
        device = start.X.device
        sched = self._prepare(device)
        pack = self.fused_pack(device)
        comp = AXL(A=start.A.clone(), X=start.X.clone(), L=start.L.clone())
        kernels.mlp_pc_sample(sched, pack, self._flags(True), self.number_of_corrector_steps,
                              self.atom_type_transition_in_corrector, starting_step_index,
                              starting_step_index - max(ending_step_index, 0), self._rng(0), comp.A, comp.X, comp.L,
                              self._status, workspace=self._noise_workspace if self.fused_predrawn_noise else None,
                              options=self.fused_sampler_options)
        return comp

    def _graph_key(self, start: AXL):
        """What a captured iteration depends on besides the tensors it reads: the batch's shape and device, the generator's
        settings that travel as kernel arguments, the network's parameters (their storage and version: the packed weight
        images are rebuilt when a parameter changes) and its arithmetic mode."""
        net = self.axl_network
        settings = (self.number_of_corrector_steps, self.resampling_steps, self.atom_type_greedy_sampling,
                    self.one_atom_type_transition_per_step, self.atom_type_transition_in_corrector,
                    self.use_fixed_lattice_parameters, self.small_epsilon, self.num_classes)     # kernel arguments of the capture
        words = tuple(None if w is None else w.data_ptr() for w in (self._call_word, self._status))   # read by the captured kernels
        return (tuple(start.X.shape), tuple(start.L.shape), str(start.X.device), getattr(net, "edge_chain_precision", None),
                settings, words, tuple((p.data_ptr(), p._version) for p in net.parameters()))

    def _sample_with_graph(self, start: AXL, starting_step_index: int, ending_step_index: int) -> AXL:
        """The iteration is captured ONCE per (shape, network state) and kept: later sample() calls copy their starting
        composition into the graph's buffers, rewrite the step index and the Philox call index on the device and replay it
        (a capture costs about five iterations' time).  The result is a copy: the buffers belong to the graph."""
        key = self._graph_key(start)
        kept = self._buffers.get("graph_loop")
        if kept is not None and getattr(kept, "key", None) == key:
            loop = kept
            loop.reset(start, starting_step_index)
        else:
            loop = IterationLoop(self, start, starting_step_index, use_graph=True)
            loop.key = key
            self._buffers["graph_loop"] = loop
        loop.advance(starting_step_index - max(ending_step_index, 0))
        comp = loop.composition
        return AXL(A=comp.A.clone(), X=comp.X.clone(), L=comp.L.clone())

    def check_status(self):
        """Read the device status word once (the only host synchronisation of a sample() call)."""
        if self._status is None:
            return
        word = int(self._status.item())
        self._status.zero_()
        net_status = getattr(self.axl_network, "graph_status", None)
        if net_status is not None:
            # The network's report comes FIRST: an activation beyond the f16 range gives non-finite logits, which can leave
            # atoms MASKED -- a report that reaches this point (a caller stepping by hand: the loops handle it per iteration)
            # must surface as an EdgeChainRangeError, not as the "there must be a bug" assertion below.  Both words are read and zeroed before anything is raised, so a stale
            # bit never leaks into the next call.
            from ..utils.neighbors import _raise_if_cutoff_too_large
            held = net_status.clone()
            net_status.zero_()
            _raise_if_cutoff_too_large(held)
        if word & STATUS_MASK_AT_LAST_STEP:
            # the reference asserts inside the last predictor step (langevin_generator.py:616-620)
            raise AssertionError("There remains MASKED atoms at the last time step: review code, there must be a "
                                 "bug or invalid input.")

    def sample(self, number_of_samples: int, device: torch.device) -> AXL:
        device = self._device(device)
        self._prepare(device)
        self._begin_call(device)
        composition = super().sample(number_of_samples, device)      # -> sample_from_noisy_composition (status read there)
        if self.rng_mode == "device":
            self.noise_source = None
        return composition


class IterationLoop:
    """The sampler's loop with its state resident on the device: composition in static buffers, loop variable i in
    a device word that every kernel reads.  With use_graph the (predictor + M correctors + index decrement)
    iteration is captured ONCE into a hipGraph (torch.cuda.CUDAGraph) and replayed; the score network must then be
    capture-safe (no host synchronisation).  Used by LangevinGenerator.sample and by bench.py."""

    def __init__(self, generator: LangevinGenerator, start: AXL, starting_step_index: int, use_graph: bool):
        import weakref
        gen = generator
        # (a weak reference: the generator keeps its last loop alive in _buffers; a strong reference back would make the pair
        # cyclic garbage, and the collector -- not the last reference going away -- would decide WHEN the loop's hipGraph is
        # destroyed: possibly in the middle of another generator's capture, where hipGraphExecDestroy aborts the process)
        self.generator = weakref.proxy(generator)
        device = start.X.device
        gen._prepare(device)
        assert getattr(gen.noise_source, "device_rng", False), "the device-resident loop needs rng_mode='device'"
        self.composition = AXL(A=start.A.clone(), X=start.X.clone(), L=start.L.clone())
        self.forces = torch.zeros_like(self.composition.X)
        self.d_index = torch.zeros(1, dtype=torch.int32, device=device)
        self.remaining = starting_step_index
        self.graph = None
        if use_graph:
            comp = self.composition
            saved = AXL(A=comp.A.clone(), X=comp.X.clone(), L=comp.L.clone())
            # warm-up on a side stream (allocator, BLAS workspaces), then restore the state
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                kernels.index_set(self.d_index, starting_step_index - 1)
                for _ in range(2):
                    gen._iteration_on_device_index(comp, self.forces, self.d_index)
            torch.cuda.current_stream(device).wait_stream(side)
            comp.A.copy_(saved.A)
            comp.X.copy_(saved.X)
            comp.L.copy_(saved.L)
            gen._status.zero_()
            gen._clear_stale_range_report()      # (the warm-up ran the split-f16 network from an arbitrary state)
            self.graph = torch.cuda.CUDAGraph()
            # Objects whose finaliser calls HIP must not be collected while the stream is capturing: an older loop's
            # torch.cuda.CUDAGraph (hipGraphExecDestroy).  It is released deterministically: the previous loop of this
            # generator is dropped here, loops of other generators die with their generator (no reference cycle: see
            # __init__), and the collector runs right before the capture and not during it.
            previous = gen._buffers.pop("graph_loop", None)
            if previous is not None:
                previous.graph = None
            del previous
            # torch.cuda.graph() no longer collects garbage on entry (torch >= 2.9: only with
            # torch.compiler.config.force_cudagraph_gc): collect here, before the capture, and keep the collector off while
            # the stream is capturing -- nothing may run a finaliser that calls HIP in there.
            import gc
            gc.collect()
            was_enabled = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(self.graph):
                    gen._iteration_on_device_index(comp, self.forces, self.d_index)
            finally:
                if was_enabled:
                    gc.enable()
        kernels.index_set(self.d_index, starting_step_index - 1)

    def reset(self, start: AXL, starting_step_index: int):
        """Start another trajectory in the same buffers (the captured graph is reused)."""
        self.composition.A.copy_(start.A)
        self.composition.X.copy_(start.X)
        self.composition.L.copy_(start.L)
        kernels.index_set(self.d_index, starting_step_index - 1)
        self.remaining = starting_step_index

    def _one(self, visits=None):
        gen = self.generator
        if self.remaining == 1 and gen.resampling_steps > 0:     # time index 0: never resampled
            gen._iteration_on_device_index(self.composition, self.forces, self.d_index, visits=1)
        elif self.graph is not None and visits is None:
            self.graph.replay()
        else:
            gen._iteration_on_device_index(self.composition, self.forces, self.d_index)
        self.remaining -= 1

    def advance(self, iterations: int):
        """Run `iterations` sampler iterations on the current stream.  Asynchronous -- except when the score network runs
        split-f16 kernels (see _advance_watched): then the call returns once the last iteration's range report has been read."""
        assert iterations <= self.remaining, "cannot step past time index 0"
        if self.generator._range_guarded():
            return self._advance_watched(iterations)
        for _ in range(iterations):
            self._one()

    # The f16-range report of the score network, watched per iteration WITHOUT stalling the device: behind every iteration the
    # network's status word is copied to page-locked memory (4 bytes, asynchronous) and an event is recorded; the host looks at
    # the report of iteration k - LAG after it has queued iteration k, so the device always has work queued.  Before an
    # iteration the composition is copied into a ring of LAG + 1 snapshots (three device copies of a few hundred kilobytes).
    # When a report shows the range bit, the loop goes back to that iteration's snapshot, runs THAT iteration with the
    # exact-f32 kernels (eagerly: the captured graph holds the split-f16 launches; the draws are functions of the time index,
    # so they are the same) and continues with graph replays: one f32 iteration and at most LAG repeated ones per event
    # instead of the whole call.
    LAG = 2

    def _watch_buffers(self):
        if getattr(self, "_watch", None) is None:
            comp, slots = self.composition, self.LAG + 1
            self._watch = dict(
                snapshots=[AXL(A=torch.empty_like(comp.A), X=torch.empty_like(comp.X), L=torch.empty_like(comp.L))
                           for _ in range(slots)],
                words=torch.zeros(slots, dtype=torch.int32).pin_memory(),
                events=[torch.cuda.Event() for _ in range(slots)], remaining=[0] * slots)
        return self._watch

    def _advance_watched(self, iterations: int):
        from .._hip import STATUS_EGNN_F16_RANGE
        gen, w = self.generator, self._watch_buffers()
        net, slots = gen.axl_network, self.LAG + 1
        gen._clear_stale_range_report()                 # (nothing of this call is queued yet: a set bit is someone else's)
        first = self.remaining                          # iteration k of this call starts with `first - k` indices remaining
        k = checked = 0                                 # iterations queued / iterations whose report has been read
        while checked < iterations:
            if k < iterations and k - checked <= self.LAG:
                slot = k % slots
                for dst, src in zip(w["snapshots"][slot], self.composition):
                    dst.copy_(src)
                assert self.remaining == first - k
                self._one()
                status = getattr(net, "graph_status", None)      # (the network creates its status word in its first forward)
                if status is None:
                    w["words"][slot] = 0
                else:
                    w["words"][slot:slot + 1].copy_(status, non_blocking=True)
                w["events"][slot].record()
                k += 1
                continue
            slot = checked % slots
            w["events"][slot].synchronize()
            if not int(w["words"][slot]) & STATUS_EGNN_F16_RANGE:
                checked += 1
                continue
            # iteration `checked` left the f16 range: everything queued behind it worked on its output -- drop it
            torch.cuda.synchronize(self.composition.X.device)
            for dst, src in zip(self.composition, w["snapshots"][slot]):
                dst.copy_(src)
            self.remaining = first - checked
            kernels.index_set(self.d_index, self.remaining - 1)
            status.bitwise_and_(~STATUS_EGNN_F16_RANGE)
            gen._status.zero_()                         # (bits the dropped iterations may have raised)
            gen._count_fallback(self.remaining - 1)
            precision = net.edge_chain_precision
            gen._begin_f16_fallback()
            net.edge_chain_precision = "f32"
            try:
                self._one(visits=1 + gen.resampling_steps)      # eager launches of the same iteration, exact-f32 kernels
            finally:
                net.edge_chain_precision = precision
                gen._adapt_f16_range()
            checked += 1
            k = checked
