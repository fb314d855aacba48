"""String keys and the AXL currency of the reference API (src/.../namespace.py:15-44), kept verbatim so that
reference-style score networks, batches and sample files plug in unchanged."""
from collections import namedtuple

CARTESIAN_POSITIONS = "cartesian_positions"
RELATIVE_COORDINATES = "relative_coordinates"
CARTESIAN_FORCES = "cartesian_forces"

NOISY_RELATIVE_COORDINATES = "noisy_relative_coordinates"
NOISY_CARTESIAN_POSITIONS = "noisy_cartesian_positions"
TIME = "time"
NOISE = "noise_parameter"
UNIT_CELL = "unit_cell"

ATOM_TYPES = "atom_types"
NOISY_ATOM_TYPES = "noisy_atom_types"

LATTICE_PARAMETERS = "lattice_parameters"
NOISY_LATTICE_PARAMETERS = "noisy_lattice_parameters"

AXL = namedtuple("AXL", ["A", "X", "L"])
AXL_NAME_DICT = {"A": ATOM_TYPES, "X": RELATIVE_COORDINATES, "L": LATTICE_PARAMETERS}

NOISY_AXL_COMPOSITION = "noisy_axl"
AXL_COMPOSITION = "original_axl"

TIME_INDICES = "time_indices"

Q_MATRICES = "q_matrices"
Q_BAR_MATRICES = "q_bar_matrices"
Q_BAR_TM1_MATRICES = "q_bar_tm1_matrices"
