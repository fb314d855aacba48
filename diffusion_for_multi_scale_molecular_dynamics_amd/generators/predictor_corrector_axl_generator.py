"""Predictor-corrector generator skeleton (src/.../generators/predictor_corrector_axl_generator.py:22-204)."""
from abc import abstractmethod
from dataclasses import dataclass
from typing import Optional

import torch

from ..namespace import AXL
from ..utils.basis_transformations import get_number_of_lattice_parameters
from .axl_generator import AXLGenerator, SamplingParameters
from .trajectory_initializer import (FullRandomTrajectoryInitializer, TrajectoryInitializer,
                                     TrajectoryInitializerParameters)


@dataclass(kw_only=True)
class PredictorCorrectorSamplingParameters(SamplingParameters):
    """Reference fields (:22-30) plus build-only switches (all default to the reference's behaviour)."""

    algorithm: str = "predictor_corrector"
    number_of_corrector_steps: int = 1
    small_epsilon: float = 1e-8
    one_atom_type_transition_per_step: bool = True
    atom_type_greedy_sampling: bool = True
    atom_type_transition_in_corrector: bool = False
    # --- build-only ---
    rng_mode: str = "reference"      # "reference": torch CPU generator, reference draw order (parity mode);
    #                                  "device": counter-based Philox inside the kernels (throughput mode)
    seed: Optional[int] = None       # device mode: Philox key (rank is added to it); None -> torch.initial_seed()
    use_hip_graph: bool = False      # device mode: capture one predictor+correctors iteration and replay it
    fused_score_network: bool = False  # device mode + MLPScoreNetwork: network forward and update fused in ONE
    #                                    persistent kernel that runs the whole loop (mdx_mlp_pc_sample)
    sync_batch_statistics: bool = False  # adaptive_corrector under torchrun: batch means over every rank's shard
    repaint_resampling_steps: int = 0  # ConstrainedLangevinGenerator only (RePaint, arXiv:2201.09865, algorithm 1 with
    #                                    jump length 1): at every time index i > 0 the reverse step i+1 -> i is followed
    #                                    by this many (forward step i -> i+1, reverse step i+1 -> i) pairs.  0 = the
    #                                    reference, which has no resampling loop.


class PredictorCorrectorAXLGenerator(AXLGenerator):
    """Skeleton of the sampler: walk the time index down from the initialiser's start to its end; at each index take one
    predictor step (i+1 -> i) followed by the corrector steps at i.  Subclasses provide the two steps."""

    def __init__(self, number_of_discretization_steps: int, number_of_corrector_steps: int, spatial_dimension: int,
                 num_atom_types: int, number_of_atoms: int, use_fixed_lattice_parameters: bool = False,
                 fixed_lattice_parameters: Optional[torch.Tensor] = None,
                 trajectory_initializer: Optional[TrajectoryInitializer] = None, **kwargs):
        assert number_of_discretization_steps > 1, "at least two discretization steps are needed"
        assert number_of_corrector_steps >= 0, "the number of corrector steps cannot be negative"
        self.number_of_discretization_steps = number_of_discretization_steps
        self.number_of_corrector_steps = number_of_corrector_steps
        self.spatial_dimension = spatial_dimension
        self.num_classes = num_atom_types + 1                       # the real types + MASK
        self.num_lattice_parameters = get_number_of_lattice_parameters(spatial_dimension)
        if trajectory_initializer is None:                          # default start: pure noise at index T
            trajectory_initializer = FullRandomTrajectoryInitializer(TrajectoryInitializerParameters(
                spatial_dimension=spatial_dimension, num_atom_types=num_atom_types, number_of_atoms=number_of_atoms,
                use_fixed_lattice_parameters=use_fixed_lattice_parameters,
                fixed_lattice_parameters=fixed_lattice_parameters))
        self.trajectory_initializer = trajectory_initializer

    def initialize(self, number_of_samples: int, device: torch.device) -> AXL:
        return self.trajectory_initializer.initialize(number_of_samples, device)

    def sample(self, number_of_samples: int, device: torch.device) -> AXL:
        initializer = self.trajectory_initializer
        return self.sample_from_noisy_composition(
            starting_noisy_composition=self.initialize(number_of_samples, device),
            starting_step_index=initializer.create_start_time_step_index(self.number_of_discretization_steps),
            ending_step_index=initializer.create_end_time_step_index())

    @staticmethod
    def _check_index_range(starting_step_index: int, ending_step_index: int):
        assert starting_step_index > ending_step_index, "the starting index must be above the ending index."
        assert starting_step_index > 0, "the starting index must be positive."
        assert ending_step_index >= 0, "the ending index cannot be negative."

    def sample_from_noisy_composition(self, starting_noisy_composition: AXL, starting_step_index: int,
                                      ending_step_index: int) -> AXL:
        self._check_index_range(starting_step_index, ending_step_index)
        composition = starting_noisy_composition
        no_forces = torch.zeros_like(composition.X)
        index = starting_step_index
        while index > ending_step_index:
            composition = self.predictor_step(composition, index, no_forces)
            index -= 1
            for _ in range(self.number_of_corrector_steps):
                composition = self.corrector_step(composition, index, no_forces)
        return composition

    @abstractmethod
    def predictor_step(self, composition_ip1: AXL, ip1: int, cartesian_forces: torch.Tensor) -> AXL:
        """composition at time index ip1 -> composition at ip1 - 1"""

    @abstractmethod
    def corrector_step(self, composition_i: AXL, i: int, cartesian_forces: torch.Tensor) -> AXL:
        """relax the composition at time index i"""
