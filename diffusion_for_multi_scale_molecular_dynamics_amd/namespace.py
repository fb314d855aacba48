"""Vocabulary of the reference's plugin API (its namespace.py:15-44).

Score networks, batches and sample files written against the reference address their tensors by these string keys and
exchange compositions as the AXL triple (A = atom types, X = relative coordinates, L = lattice parameters).  The VALUES
are part of the drop-in contract and therefore identical to the reference's; nothing else is defined here.
"""
from collections import namedtuple

# the composition triple and the keys it is stored under in a batch / a samples file
AXL = namedtuple("AXL", "A X L")
AXL_COMPOSITION, NOISY_AXL_COMPOSITION = "original_axl", "noisy_axl"

# per-field keys (clean, then noised)
ATOM_TYPES, RELATIVE_COORDINATES, LATTICE_PARAMETERS = "atom_types", "relative_coordinates", "lattice_parameters"
NOISY_ATOM_TYPES, NOISY_RELATIVE_COORDINATES, NOISY_LATTICE_PARAMETERS = (
    "noisy_" + ATOM_TYPES, "noisy_" + RELATIVE_COORDINATES, "noisy_" + LATTICE_PARAMETERS)
AXL_NAME_DICT = dict(zip(AXL._fields, (ATOM_TYPES, RELATIVE_COORDINATES, LATTICE_PARAMETERS)))

# Cartesian views and the cell
CARTESIAN_POSITIONS, NOISY_CARTESIAN_POSITIONS = "cartesian_positions", "noisy_cartesian_positions"
CARTESIAN_FORCES, UNIT_CELL = "cartesian_forces", "unit_cell"

# diffusion time, the noise level sigma(t), and the D3PM transition matrices of a noised batch
TIME, TIME_INDICES, NOISE = "time", "time_indices", "noise_parameter"
Q_MATRICES, Q_BAR_MATRICES, Q_BAR_TM1_MATRICES = "q_matrices", "q_bar_matrices", "q_bar_tm1_matrices"
