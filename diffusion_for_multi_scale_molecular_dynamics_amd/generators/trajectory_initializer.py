"""Where a sampling trajectory starts (reference: generators/trajectory_initializer.py:17-214).

Two starts exist: pure noise at the last time index (A = MASK, X uniform, L fixed or Gaussian), or a stored noisy
composition part-way down the schedule.  The random start takes its draws from the generator's noise source, so that
it follows the reference's CPU draw order in parity mode and the device Philox stream in throughput mode.
"""
import abc
import dataclasses
import os
from typing import Optional, Union

import torch

from ..namespace import AXL, NOISY_AXL_COMPOSITION
from ..utils.basis_transformations import get_number_of_lattice_parameters
from .axl_generator import SamplingParameters


@dataclasses.dataclass(kw_only=True)
class TrajectoryInitializerParameters:
    spatial_dimension: int = 3
    num_atom_types: int
    use_fixed_lattice_parameters: bool = False
    fixed_lattice_parameters: Optional[torch.Tensor] = None
    number_of_atoms: int
    path_to_starting_configuration_data_pickle: Optional[str] = None

    def __post_init__(self):
        lattice = self.fixed_lattice_parameters
        if not self.use_fixed_lattice_parameters:
            assert lattice is None, "fixed_lattice_parameters is only meaningful with use_fixed_lattice_parameters=True."
            return
        assert lattice is not None, "use_fixed_lattice_parameters=True needs fixed_lattice_parameters."
        expected = get_number_of_lattice_parameters(self.spatial_dimension)
        assert lattice.shape[0] == expected, f"expected {expected} lattice parameters, got shape {tuple(lattice.shape)}."


class TrajectoryInitializer(abc.ABC):
    def __init__(self, trajectory_initializer_parameters: TrajectoryInitializerParameters) -> None:
        self.trajectory_initializer_parameters = trajectory_initializer_parameters
        for name in ("spatial_dimension", "number_of_atoms", "use_fixed_lattice_parameters", "fixed_lattice_parameters"):
            setattr(self, name, getattr(trajectory_initializer_parameters, name))
        self.masked_atom_type_index = trajectory_initializer_parameters.num_atom_types      # MASK is the last class
        self.num_lattice_parameters = get_number_of_lattice_parameters(self.spatial_dimension)

    @abc.abstractmethod
    def initialize(self, number_of_samples: int, device: torch.device) -> AXL:
        """Starting composition of `number_of_samples` trajectories."""

    @abc.abstractmethod
    def create_start_time_step_index(self, number_of_discretization_steps: int) -> int:
        """Time index of the first predictor step."""

    @abc.abstractmethod
    def create_end_time_step_index(self) -> int:
        """Time index at which the trajectory stops."""


class FullRandomTrajectoryInitializer(TrajectoryInitializer):
    """Pure-noise start at time index T."""

    noise_source = None      # set by the generator; None = torch's CPU generator, as the reference

    def initialize(self, number_of_samples: int, device: torch.device) -> AXL:
        batch, atoms, dim = number_of_samples, self.number_of_atoms, self.spatial_dimension
        source = self.noise_source
        coordinates = (torch.rand(batch, atoms, dim).to(device) if source is None
                       else source.initial_coordinates(batch, atoms, dim, device))
        if self.use_fixed_lattice_parameters:
            lattice = self.fixed_lattice_parameters.repeat(batch, 1).to(device)
        elif source is None:
            lattice = torch.randn(batch, self.num_lattice_parameters).to(device)
        else:
            lattice = source.initial_lattice(batch, self.num_lattice_parameters, device)
        masked = torch.full((batch, atoms), self.masked_atom_type_index, dtype=torch.int64, device=device)
        return AXL(A=masked, X=coordinates, L=lattice)

    def create_start_time_step_index(self, number_of_discretization_steps: int) -> int:
        return number_of_discretization_steps

    def create_end_time_step_index(self) -> int:
        return 0


class StartFromGivenConfigurationTrajectoryInitializer(TrajectoryInitializer):
    """Resume from a pickle holding {noisy_axl: AXL [B, ...], start_time_step_index: int}."""

    def __init__(self, trajectory_initializer_parameters: TrajectoryInitializerParameters) -> None:
        super().__init__(trajectory_initializer_parameters)
        pickle_path = trajectory_initializer_parameters.path_to_starting_configuration_data_pickle
        assert os.path.isfile(pickle_path), f"starting configuration file not found: {pickle_path}"
        from ..utils import reference_pickles
        stored = reference_pickles.load(pickle_path)      # (a file made with the reference's tools names ITS AXL class: read here too)
        self.noisy_starting_composition = stored[NOISY_AXL_COMPOSITION]
        self.start_time_step_index = stored["start_time_step_index"]

    def initialize(self, number_of_samples: int, device: torch.device) -> AXL:
        start = self.noisy_starting_composition
        assert start.X.shape[0] == number_of_samples, \
            f"{number_of_samples} samples requested but the starting-configuration file holds {start.X.shape[0]}."
        return AXL(*[field.to(device) for field in start])

    def create_start_time_step_index(self, number_of_discretization_steps: int) -> int:
        return self.start_time_step_index

    def create_end_time_step_index(self) -> int:
        return 0


def instantiate_trajectory_initializer(sampling_parameters: SamplingParameters,
                                       path_to_starting_configuration_data_pickle: Union[str, None] = None
                                       ) -> TrajectoryInitializer:
    shared = {name: getattr(sampling_parameters, name) for name in (
        "spatial_dimension", "num_atom_types", "number_of_atoms", "use_fixed_lattice_parameters",
        "fixed_lattice_parameters")}
    parameters = TrajectoryInitializerParameters(
        **shared, path_to_starting_configuration_data_pickle=path_to_starting_configuration_data_pickle)
    if path_to_starting_configuration_data_pickle:
        return StartFromGivenConfigurationTrajectoryInitializer(parameters)
    return FullRandomTrajectoryInitializer(parameters)
