"""CIF / extended-XYZ writers for sampled structures and recorded trajectories (on-disk formats downstream of the path).

Same entry points, directory / file naming and XYZ header as the reference's writers
(src/.../analysis/ovito_utilities/trajectory_io.py:24-140, xyz_utils.py:7-66), so that OVITO session files and scripts
written for the reference's output keep working.  The reference builds pymatgen `Structure`s and lets pymatgen write the
CIF; pymatgen is not a dependency here: the CIF is a plain P1 cell (lengths and angles from the basis vectors, fractional
coordinates as sampled), which OVITO, ASE and pymatgen all read.  Host-side I/O only: nothing here touches the GPU.
"""
import math
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch

from ...namespace import AXL, AXL_COMPOSITION

UNKNOWN_ATOM_TYPE = "X"

CIF_DIRECTORY_TEMPLATE = "cif_files_trajectory_{trajectory_index}"
CIF_FILENAME_TEMPLATE = "diffusion_positions_step_{time_index}.cif"
XYZ_DIRECTORY_TEMPLATE = "xyz_files_trajectory_{trajectory_index}"
XYZ_FILENAME_TEMPLATE = "diffusion_positions_step_{time_index}.xyz"


def _atom_type_map(elements: List[str]) -> Dict[int, str]:
    """index -> symbol in the reference's ElementTypes order (sorted element names); MASK = number of elements -> 'X'."""
    symbols = dict(enumerate(sorted(elements)))
    symbols[len(elements)] = UNKNOWN_ATOM_TYPE
    return symbols


def _basis(lattice) -> np.ndarray:
    """[d,d] basis vectors from either a [d,d] matrix or the [d(d+1)/2] lattice parameters (a, b, c, 0, 0, 0) of the
    sampler (orthogonal boxes, utils/basis_transformations.py:141-170)."""
    lattice = np.asarray(lattice, dtype=np.float64)
    if lattice.ndim == 2:
        return lattice
    d = int((-1 + math.sqrt(1 + 8 * lattice.shape[0])) / 2)
    return np.diag(lattice[:d])


def cif_text(symbols: List[str], relative_coordinates: np.ndarray, basis_vectors: np.ndarray) -> str:
    a, b, c = (np.linalg.norm(v) for v in basis_vectors)

    def angle(u, v):
        return math.degrees(math.acos(max(-1.0, min(1.0, float(np.dot(u, v) / (np.linalg.norm(u) * np.linalg.norm(v)))))))
    alpha, beta, gamma = angle(basis_vectors[1], basis_vectors[2]), angle(basis_vectors[0], basis_vectors[2]), \
        angle(basis_vectors[0], basis_vectors[1])
    counts = {}
    for s in symbols:
        counts[s] = counts.get(s, 0) + 1
    formula = " ".join(f"{s}{n}" for s, n in sorted(counts.items()))
    lines = [f"data_{formula.replace(' ', '')}", "_symmetry_space_group_name_H-M   'P 1'",
             f"_cell_length_a   {a:.8f}", f"_cell_length_b   {b:.8f}", f"_cell_length_c   {c:.8f}",
             f"_cell_angle_alpha   {alpha:.8f}", f"_cell_angle_beta   {beta:.8f}", f"_cell_angle_gamma   {gamma:.8f}",
             "_symmetry_Int_Tables_number   1", f"_chemical_formula_sum   '{formula}'",
             f"_cell_volume   {abs(float(np.linalg.det(basis_vectors))):.8f}", "loop_", " _symmetry_equiv_pos_site_id",
             " _symmetry_equiv_pos_as_xyz", "  1  'x, y, z'", "loop_", " _atom_site_type_symbol", " _atom_site_label",
             " _atom_site_symmetry_multiplicity", " _atom_site_fract_x", " _atom_site_fract_y", " _atom_site_fract_z",
             " _atom_site_occupancy"]
    for k, (s, x) in enumerate(zip(symbols, relative_coordinates)):
        lines.append(f"  {s}  {s}{k}  1  {x[0]:.8f}  {x[1]:.8f}  {x[2]:.8f}  1")
    return "\n".join(lines) + "\n"


def xyz_text(relative_coordinates: np.ndarray, basis_vectors: np.ndarray,
             site_properties: Optional[Dict[str, np.ndarray]] = None) -> str:
    """Extended XYZ with the reference's header (xyz_utils.py:28-66): Lattice="..." Origin="0 0 0" pbc="T T T"
    Properties=pos:R:3[:name:R:dim ...], Cartesian positions, then the per-atom properties."""
    site_properties = site_properties or {}
    header = 'Lattice="' + " ".join(map(str, np.asarray(basis_vectors, dtype=np.float64).flatten())) + \
        '" Origin="0 0 0" pbc="T T T" Properties=pos:R:3'
    columns = []
    for name, values in site_properties.items():
        values = np.asarray(values)
        values = values.reshape(len(relative_coordinates), -1)
        header += f":{name}:R:{values.shape[1]}"
        columns.append(values)
    cartesian = np.asarray(relative_coordinates, dtype=np.float64) @ np.asarray(basis_vectors, dtype=np.float64)
    rows = []
    for i, p in enumerate(cartesian):
        row = " ".join(map(str, p))
        for values in columns:
            row += " " + " ".join(map(str, values[i]))
        rows.append(row)
    return f"{len(cartesian)}\n{header}\n" + "\n".join(rows) + "\n"


def _trajectory(trajectory_axl_compositions: AXL, trajectory_index: Optional[int]):
    """[time, ...] numpy arrays of one trajectory; fields are [samples, time, ...] (or [time, ...] with index None)."""
    pick = (lambda t: t[trajectory_index]) if trajectory_index is not None else (lambda t: t)
    return [np.asarray(pick(torch.as_tensor(f)).cpu()) for f in trajectory_axl_compositions]


def create_io_files(elements: List[str], visualization_artifacts_path: Path, trajectory_index: Optional[int],
                    trajectory_axl_compositions: AXL, atomic_properties: Optional[Dict[str, torch.Tensor]], format: str):
    """One file per time step under <path>/<format>_files_trajectory_<index>/ (trajectory_io.py:84-138)."""
    if format not in ("cif", "xyz"):
        raise NotImplementedError(f"no such format {format}")
    symbols = _atom_type_map(elements)
    atom_types, coordinates, lattices = _trajectory(trajectory_axl_compositions, trajectory_index)
    n_steps = len(atom_types)
    properties = {}
    for name, values in (atomic_properties or {}).items():
        values = np.asarray((values[trajectory_index] if trajectory_index is not None else values).cpu())
        assert len(values) == n_steps, f"The number of time steps in property {name} is inconsistent with expectation."
        properties[name] = values
    template = CIF_DIRECTORY_TEMPLATE if format == "cif" else XYZ_DIRECTORY_TEMPLATE
    directory = Path(visualization_artifacts_path) / template.format(
        trajectory_index=trajectory_index if trajectory_index is not None else 0)
    directory.mkdir(parents=True, exist_ok=True)
    for time_index in range(n_steps):
        basis = _basis(lattices[time_index])
        if format == "cif":                              # (site properties are ignored by the CIF writer, as in the reference)
            text = cif_text([symbols[int(a)] for a in atom_types[time_index]], coordinates[time_index], basis)
            name = CIF_FILENAME_TEMPLATE.format(time_index=time_index)
        else:
            text = xyz_text(coordinates[time_index], basis, {k: v[time_index] for k, v in properties.items()})
            name = XYZ_FILENAME_TEMPLATE.format(time_index=time_index)
        (directory / name).write_text(text)


def create_cif_files(elements: List[str], visualization_artifacts_path: Path, trajectory_index: int,
                     trajectory_axl_compositions: AXL):
    create_io_files(elements, visualization_artifacts_path, trajectory_index, trajectory_axl_compositions, None, "cif")


def create_xyz_files(elements: List[str], visualization_artifacts_path: Path, trajectory_index: Optional[int],
                     trajectory_axl_compositions: AXL, atomic_properties: Optional[Dict[str, torch.Tensor]]):
    create_io_files(elements, visualization_artifacts_path, trajectory_index, trajectory_axl_compositions,
                    atomic_properties, "xyz")


def write_samples(samples_path, elements: List[str], output_directory, format: str = "cif"):
    """`samples.pt` of sample_diffusion ({"cartesian_positions", "original_axl"}) -> one file per sampled structure: the
    batch is written as a single 'trajectory' whose steps are the samples."""
    from ...utils import reference_pickles
    data = reference_pickles.load(samples_path)           # (this package's samples.pt or the reference's)
    create_io_files(elements, Path(output_directory), None, data[AXL_COMPOSITION], None, format)
