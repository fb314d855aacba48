"""Trajectory initialisers (src/.../generators/trajectory_initializer.py:17-214)."""
import os
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Optional, Union

import torch

from ..namespace import AXL, NOISY_AXL_COMPOSITION
from ..utils.basis_transformations import get_number_of_lattice_parameters
from .axl_generator import SamplingParameters


@dataclass(kw_only=True)
class TrajectoryInitializerParameters:
    spatial_dimension: int = 3
    num_atom_types: int
    use_fixed_lattice_parameters: bool = False
    fixed_lattice_parameters: Optional[torch.Tensor] = None
    number_of_atoms: int
    path_to_starting_configuration_data_pickle: Optional[str] = None

    def __post_init__(self):
        if self.use_fixed_lattice_parameters:
            assert self.fixed_lattice_parameters is not None, \
                "If use_fixed_lattice_parameters is True, then fixed_lattice_parameters must be provided."
            assert self.fixed_lattice_parameters.shape[0] == get_number_of_lattice_parameters(self.spatial_dimension), \
                f"fixed_lattice_parameters must have d(d+1)/2 entries. Got {self.fixed_lattice_parameters.shape}."
        else:
            assert self.fixed_lattice_parameters is None, \
                "fixed_lattice_parameters must be None if use_fixed_lattice_parameters is False."


class TrajectoryInitializer(ABC):
    def __init__(self, trajectory_initializer_parameters: TrajectoryInitializerParameters) -> None:
        p = trajectory_initializer_parameters
        self.trajectory_initializer_parameters = p
        self.spatial_dimension = p.spatial_dimension
        self.number_of_atoms = p.number_of_atoms
        self.masked_atom_type_index = p.num_atom_types
        self.num_lattice_parameters = get_number_of_lattice_parameters(p.spatial_dimension)
        self.use_fixed_lattice_parameters = p.use_fixed_lattice_parameters
        self.fixed_lattice_parameters = p.fixed_lattice_parameters

    @abstractmethod
    def initialize(self, number_of_samples: int, device: torch.device) -> AXL:
        pass

    @abstractmethod
    def create_start_time_step_index(self, number_of_discretization_steps: int) -> int:
        pass

    @abstractmethod
    def create_end_time_step_index(self) -> int:
        pass


class FullRandomTrajectoryInitializer(TrajectoryInitializer):
    """A = MASK, X ~ U[0,1), L fixed or N(0,1)  (:101-123).  `noise_source` (set by the generator) supplies the
    draws: reference-order CPU draws, a replayed fixture, or the device Philox stream."""

    noise_source = None

    def initialize(self, number_of_samples: int, device: torch.device) -> AXL:
        n, d = self.number_of_atoms, self.spatial_dimension
        atom_types = torch.full((number_of_samples, n), self.masked_atom_type_index, dtype=torch.int64, device=device)
        src = self.noise_source
        if src is None:
            x = torch.rand(number_of_samples, n, d).to(device)
        else:
            x = src.initial_coordinates(number_of_samples, n, d, device)
        if self.use_fixed_lattice_parameters:
            lattice = self.fixed_lattice_parameters.repeat(number_of_samples, 1).to(device)
        elif src is None:
            lattice = torch.randn(number_of_samples, self.num_lattice_parameters).to(device)
        else:
            lattice = src.initial_lattice(number_of_samples, self.num_lattice_parameters, device)
        return AXL(A=atom_types, X=x, L=lattice)

    def create_start_time_step_index(self, number_of_discretization_steps: int) -> int:
        return number_of_discretization_steps

    def create_end_time_step_index(self) -> int:
        return 0


class StartFromGivenConfigurationTrajectoryInitializer(TrajectoryInitializer):
    """Start mid-trajectory from a pickle {noisy_axl: AXL, start_time_step_index: int}  (:134-186)."""

    def __init__(self, trajectory_initializer_parameters: TrajectoryInitializerParameters) -> None:
        super().__init__(trajectory_initializer_parameters)
        path = trajectory_initializer_parameters.path_to_starting_configuration_data_pickle
        assert os.path.isfile(path), f"The file {path} does not exist. Review input."
        data = torch.load(path, weights_only=False)
        self.noisy_starting_composition = data[NOISY_AXL_COMPOSITION]
        self.start_time_step_index = data["start_time_step_index"]

    def initialize(self, number_of_samples: int, device: torch.device) -> AXL:
        comp = self.noisy_starting_composition
        assert number_of_samples == comp.X.shape[0], \
            "The number of samples requested is inconsistent with the number of starting configurations in the " \
            "data pickle. Something is probably inconsistent: stopping here, review inputs."
        return AXL(A=comp.A.to(device), X=comp.X.to(device), L=comp.L.to(device))

    def create_start_time_step_index(self, number_of_discretization_steps: int) -> int:
        return self.start_time_step_index

    def create_end_time_step_index(self) -> int:
        return 0


def instantiate_trajectory_initializer(sampling_parameters: SamplingParameters,
                                       path_to_starting_configuration_data_pickle: Union[str, None] = None
                                       ) -> TrajectoryInitializer:
    """:189-214"""
    params = TrajectoryInitializerParameters(
        spatial_dimension=sampling_parameters.spatial_dimension,
        num_atom_types=sampling_parameters.num_atom_types,
        number_of_atoms=sampling_parameters.number_of_atoms,
        use_fixed_lattice_parameters=sampling_parameters.use_fixed_lattice_parameters,
        fixed_lattice_parameters=sampling_parameters.fixed_lattice_parameters,
        path_to_starting_configuration_data_pickle=path_to_starting_configuration_data_pickle)
    if path_to_starting_configuration_data_pickle:
        return StartFromGivenConfigurationTrajectoryInitializer(params)
    return FullRandomTrajectoryInitializer(params)
