"""The reference's configuration and checkpoint surface (SURVEY 8(b) "CLI / files"): its YAML files load into the same parameter
objects, its networks' state_dicts fit this package's modules, and a Lightning checkpoint is read WITHOUT Lightning and without
the reference package -- a sampling configuration of the reference (noise + sampling, no `model:` block) is then all the CLI
needs, as in src/.../sample_diffusion.py:191-205."""
import dataclasses
import json
import os
import subprocess
import sys

import pytest
import torch

import nets
from conftest import GOLDEN, ROOT

REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference is not on this machine")
def test_every_reference_yaml_file_and_checkpoints_made_of_reference_objects(tmp_path):
    """tests/golden/yaml_surface.py (a child process WITH the reference on its path) walks every YAML file of the reference tree
    and writes two Lightning-style checkpoints whose pickled hyper-parameters are the reference's own dataclass instances; this
    process (WITHOUT the reference) reads the report and the checkpoints."""
    env = dict(os.environ, PYTHONPATH=os.path.join(REFERENCE, "src"))
    run = subprocess.run([sys.executable, os.path.join(GOLDEN, "yaml_surface.py"), REFERENCE, str(tmp_path)], env=env, cwd=str(tmp_path),
                         capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    report = json.loads(run.stdout.strip().splitlines()[-1])
    assert report["files"] >= 40
    # noise + sampling blocks of `algorithm: predictor_corrector`: equal on every field of the reference's dataclasses, or refused
    # by both with the same message (four templates carry a key that is not a field)
    loaded = [p for p in report["pairs"] if "differing" in p]
    refused = [p for p in report["pairs"] if "reference_refuses" in p]
    assert len(loaded) >= 13 and all(p["differing"] == [] for p in loaded), [p for p in loaded if p["differing"]]
    assert all(p["own_refuses"] == p["reference_refuses"] for p in refused), refused
    # score-network blocks (mlp, egnn): same parameters; every network the reference can build has the same state_dict keys, shapes
    # and dtypes as this package's; the template the reference refuses is refused here with the same message
    built = [n for n in report["networks"] if "state_dict_matches" in n]
    assert len(built) >= 10 and all(n["state_dict_matches"] and n["differing"] == [] for n in built), built
    assert {n["architecture"] for n in built} == {"mlp", "egnn"}
    for n in report["networks"]:
        if "reference_refuses" in n:
            assert n["own_refuses"] == n["reference_refuses"], n
        if "reference_cannot_build" in n:
            assert n["differing"] == []
    assert {s["architecture"] for s in report["skipped_architectures"]} <= {"mace", "diffusion_mace", "analytical", "equivariant_analytical"}
    # the checkpoints: network rebuilt from the pickled hyper-parameters alone, weights loaded
    assert "diffusion_for_multi_scale_molecular_dynamics" not in sys.modules
    from diffusion_for_multi_scale_molecular_dynamics_amd.sample_diffusion import get_axl_network
    for name, record in report["checkpoints"].items():
        network = get_axl_network(record["file"])
        parameters = network._hyper_params
        assert parameters.architecture == name
        for key, value in record["parameters"].items():
            assert getattr(parameters, key) == value, (name, key)
        state = network.state_dict()
        assert len(state) == record["tensors"]
        assert abs(float(sum(v.double().sum() for v in state.values())) - record["checksum"]) <= 1e-9 * max(1.0, abs(record["checksum"]))
        assert not network.training


def test_lightning_style_checkpoint_without_lightning_or_the_reference(tmp_path):
    """A checkpoint whose pickle names classes this process cannot import (the reference package's, Lightning's): the
    score-network parameters are found under the same relative module path in this package, everything else becomes an inert
    placeholder; the network is rebuilt from them alone and loads its `axl_network.*` weights.  A `model: score_network:` block
    in the configuration takes precedence; a bare state_dict without one, or an architecture this package does not implement,
    is refused with a message that says what is missing."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.egnn_score_network import (
        EGNNScoreNetwork, EGNNScoreNetworkParameters)
    from diffusion_for_multi_scale_molecular_dynamics_amd.sample_diffusion import get_axl_network
    from diffusion_for_multi_scale_molecular_dynamics_amd.utils import lightning_checkpoint
    assert "diffusion_for_multi_scale_molecular_dynamics" not in sys.modules and "lightning" not in sys.modules
    parameters = EGNNScoreNetworkParameters(num_atom_types=2, n_layers=2, coordinate_hidden_dimensions_size=32,
                                            coordinate_n_hidden_dimensions=2, message_hidden_dimensions_size=32,
                                            message_n_hidden_dimensions=2, node_hidden_dimensions_size=32, node_n_hidden_dimensions=2,
                                            edges="radial_cutoff", radial_cutoff=4.5, tanh=True)
    torch.manual_seed(5)
    trained = EGNNScoreNetwork(parameters)
    path = tmp_path / "last_model.ckpt"
    nets.write_lightning_style_checkpoint(path, trained, parameters)
    assert "diffusion_for_multi_scale_molecular_dynamics.models.axl_diffusion_lightning_model" not in sys.modules
    with pytest.raises((ModuleNotFoundError, AttributeError)):
        torch.load(path, weights_only=False)                     # the plain loader cannot read it here
    checkpoint = lightning_checkpoint.load_checkpoint(path)
    outer = checkpoint["hyper_parameters"]["hyper_params"]
    assert isinstance(outer, lightning_checkpoint.Placeholder) and outer.kmax_target_score == 4
    assert isinstance(outer.score_network_parameters, EGNNScoreNetworkParameters)      # mapped onto this package's class
    assert isinstance(checkpoint["callbacks"]["ModelCheckpoint"], lightning_checkpoint.Placeholder)
    network = get_axl_network(path)
    assert isinstance(network, EGNNScoreNetwork) and not network.training
    assert dataclasses.asdict(network._hyper_params) == dataclasses.asdict(parameters)
    assert all(torch.equal(a, b) for a, b in zip(network.state_dict().values(), trained.state_dict().values()))
    # the configuration's block wins when there is one
    block = dict(dataclasses.asdict(parameters), radial_cutoff=5.5)
    assert get_axl_network(path, dict(model=dict(score_network=block), elements=["Si", "Ge"]))._hyper_params.radial_cutoff == 5.5
    with pytest.raises(AssertionError, match="'num_atom_types' entries"):
        get_axl_network(path, dict(model=dict(score_network=block), elements=["Si"]))
    # a bare state_dict needs the block
    torch.save({"state_dict": {"axl_network." + k: v for k, v in trained.state_dict().items()}}, tmp_path / "bare.ckpt")
    with pytest.raises(AssertionError, match="holds no hyper-parameters"):
        get_axl_network(tmp_path / "bare.ckpt")
    assert isinstance(get_axl_network(tmp_path / "bare.ckpt", dict(model=dict(score_network=dataclasses.asdict(parameters)))), EGNNScoreNetwork)
    # an architecture that is not implemented here
    other = dataclasses.replace(parameters)
    other.architecture = "diffusion_mace"
    nets.write_lightning_style_checkpoint(tmp_path / "mace.ckpt", trained, other)
    with pytest.raises(AssertionError, match="not implemented here"):
        get_axl_network(tmp_path / "mace.ckpt")


def _samples():
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import AXL
    g = torch.Generator().manual_seed(7)
    axl = AXL(A=torch.randint(0, 2, (5, 8), generator=g), X=torch.rand(5, 8, 3, generator=g), L=torch.rand(5, 6, generator=g))
    return {"cartesian_positions": torch.rand(5, 8, 3, generator=g), "original_axl": axl}, axl


def test_pickles_named_for_the_reference_and_read_back(tmp_path):
    """utils/reference_pickles: save_for_reference names the reference's AXL class in the file (a process with this package
    alone cannot torch.load it: that is the point), load() reads it -- and any nesting of dicts, lists and tuples -- back into
    this package's AXL; the trajectory recorder's write_to_pickle(for_reference=True) goes the same way."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import AXL
    from diffusion_for_multi_scale_molecular_dynamics_amd.utils import reference_pickles
    from diffusion_for_multi_scale_molecular_dynamics_amd.utils.sample_trajectory import SampleTrajectory
    samples, axl = _samples()
    nested = dict(samples, steps=[axl, (axl, 3)], note="text")
    reference_pickles.save_for_reference(nested, tmp_path / "samples.pt")
    assert "diffusion_for_multi_scale_molecular_dynamics" not in sys.modules            # the lent name is returned
    with pytest.raises(ModuleNotFoundError):
        torch.load(tmp_path / "samples.pt", weights_only=False)
    assert b"diffusion_for_multi_scale_molecular_dynamics_amd" not in (tmp_path / "samples.pt").read_bytes()
    back = reference_pickles.load(tmp_path / "samples.pt")
    assert type(back["original_axl"]) is AXL and type(back["steps"][0]) is AXL and type(back["steps"][1][0]) is AXL
    assert back["steps"][1][1] == 3 and back["note"] == "text"
    assert all(torch.equal(a, b) for a, b in zip(back["original_axl"], axl))
    assert torch.equal(back["cartesian_positions"], samples["cartesian_positions"])
    recorder = SampleTrajectory()
    recorder.record(key="predictor_step", entry=dict(composition_i=axl, time_step_index=2))
    recorder.write_to_pickle(tmp_path / "trajectories.pt", for_reference=True)
    assert b"diffusion_for_multi_scale_molecular_dynamics_amd" not in (tmp_path / "trajectories.pt").read_bytes()
    assert type(reference_pickles.load(tmp_path / "trajectories.pt")["predictor_step"]["composition_i"]) is AXL


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference is not on this machine")
def test_files_cross_between_the_reference_and_this_package(tmp_path):
    """A child process that has the REFERENCE and not this package reads a samples.pt written for it with a plain torch.load
    (it holds the reference's own AXL) and writes a starting-configuration pickle with its tools' layout; this process, which has
    this package and not the reference, starts a trajectory from that pickle."""
    from diffusion_for_multi_scale_molecular_dynamics_amd.generators.trajectory_initializer import (
        StartFromGivenConfigurationTrajectoryInitializer, TrajectoryInitializerParameters)
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import AXL
    from diffusion_for_multi_scale_molecular_dynamics_amd.utils import reference_pickles
    samples, axl = _samples()
    reference_pickles.save_for_reference(samples, tmp_path / "samples.pt")
    from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_scheduler import Noise
    from diffusion_for_multi_scale_molecular_dynamics_amd.utils.sample_trajectory import SampleTrajectory
    recorder = SampleTrajectory()
    recorder.record(key="noise", entry=Noise(*[torch.arange(3.0) + k for k in range(len(Noise._fields))]))
    recorder.record(key="predictor_step", entry=dict(composition_i=axl, time_step_index=2))
    recorder.write_to_pickle(tmp_path / "trajectories.pt", for_reference=True)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = os.path.join(REFERENCE, "src")
    run = subprocess.run([sys.executable, os.path.join(GOLDEN, "cross_pickles.py"), str(tmp_path / "samples.pt"), str(tmp_path / "start.pt"),
                          str(tmp_path / "trajectories.pt")],
                         env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-3000:]
    seen = json.loads(run.stdout.strip().splitlines()[-1])
    assert seen["samples_class"] == "diffusion_for_multi_scale_molecular_dynamics.namespace.AXL"
    assert abs(seen["x_sum"] - float(axl.X.double().sum())) < 1e-9 and seen["cartesian_shape"] == [5, 8, 3]
    with pytest.raises(ModuleNotFoundError):
        torch.load(tmp_path / "start.pt", weights_only=False)                       # the reference's file names the reference's class
    parameters = TrajectoryInitializerParameters(spatial_dimension=3, num_atom_types=1, number_of_atoms=8,
                                                 path_to_starting_configuration_data_pickle=str(tmp_path / "start.pt"))
    initializer = StartFromGivenConfigurationTrajectoryInitializer(parameters)
    start = initializer.initialize(3, torch.device("cpu"))
    assert type(start) is AXL and start.X.shape == (3, 8, 3) and initializer.create_start_time_step_index(10) == 4
    assert abs(float(start.X.double().sum()) - seen["start_x_sum"]) < 1e-9


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference is not on this machine")
def test_public_surface_of_the_modules_shared_with_the_reference():
    """tests/golden/api_surface.py: for the 41 modules of this package that exist at the same relative path in the reference
    (34 importable there, 7 read from their source text), every public class, function and method the reference defines is here, dataclass fields and
    defaults agree, parameters carry the reference's names, order and defaults, and the private methods the reference's tests and
    subclasses reach for are served; the small helpers under the reference's paths return the reference's values.  The modules
    without a counterpart are this package's own (kernels, RNG sources, pickles) or need a dependency the container lacks to
    import on the reference's side (they are covered by the YAML / checkpoint test above through their dataclasses)."""
    env = dict(os.environ, PYTHONPATH=os.path.join(REFERENCE, "src"))
    run = subprocess.run([sys.executable, os.path.join(GOLDEN, "api_surface.py")], env=env, cwd="/tmp", capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    report = json.loads(run.stdout.strip().splitlines()[-1])
    assert len(report["modules_compared"]) >= 34
    # seven more cannot be imported on the reference's side here (pymatgen, torchode, mace, orion): compared from their source text
    assert set(report["modules_compared_by_syntax"]) == {
        ".analysis.ovito_utilities.trajectory_io", ".utils.structure_utils", ".generators.instantiate_generator",
        ".generators.load_sampling_parameters", ".sampling.diffusion_sampling_parameters",
        ".models.score_networks.score_network_factory", ".sample_diffusion"}
    # what is knowingly absent: the pymatgen-based INSIDES of the reference's CIF / XYZ writers (its public entry points
    # create_cif_files / create_xyz_files / create_io_files are here, writing the text themselves) and the pymatgen Structure factory
    accepted = {".analysis.ovito_utilities.trajectory_io." + name for name in (
        "get_list_site_properties_and_atomic_properties_dim", "get_list_trajectory_AXLs", "StructureWriter", "CifStructureWriter",
        "XyzStructureWriter")} | {".utils.structure_utils.create_structure"}
    assert set(report["public_missing"]) == accepted and report["served_private_missing"] == []
    assert report["signature_differences"] == [] and report["dataclass_differences"] == []
    assert all(report["helper_values"].values()), report["helper_values"]
    own_only = {".kernels", "._hip", ".generators.noise_sources", ".utils.batch_statistics", ".utils.lightning_checkpoint", ".utils.reference_pickles"}
    assert {m["module"] for m in report["modules_without_counterpart"]} <= own_only


REFERENCE_TEST_FILES = {        # the reference's own test files that need no GPU: file -> tests it holds (all must pass)
    "tests/noise_schedulers/test_sigma_calculator.py": 2, "tests/utils/test_lattice_utils.py": 12, "tests/utils/test_noise_utils.py": 6,
    "tests/models/test_egnn_utils.py": 2, "tests/generators/test_sampling_constraint.py": 2, "tests/sampling/test_diffusion_sampling.py": 1,
    "tests/noise_schedulers/test_exploding_variance.py": 6, "tests/utils/test_tensor_utils.py": 24, "tests/utils/test_symmetry_utils.py": 4}


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference is not on this machine")
@pytest.mark.parametrize("test_file", list(REFERENCE_TEST_FILES))
def test_the_references_own_tests_of_the_host_side_helpers_pass_here(test_file, tmp_path):
    """The reference's OWN test files, unmodified, run against this package: a child pytest with an import alias
    (tests/golden/reference_import_alias.py: the reference's module names -- also in the `src.`-prefixed form some of its test
    files use -- resolve to this package's modules; the plugin fails the run if any module of the reference's source tree was
    imported) collects the file from /root/reference/tests and every test in it passes.  Only the files that
    need no GPU can run in this container (everything that reaches a kernel refuses host tensors: there is no CPU fallback --
    e.g. tests/utils/test_structure_utils.py::test_compute_distances stops at exactly that message)."""
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = os.pathsep.join([GOLDEN, ROOT])
    run = subprocess.run([sys.executable, "-m", "pytest", "-p", "reference_import_alias", os.path.join(REFERENCE, test_file), "-q",
                          "--no-header", "-p", "no:cacheprovider", "--rootdir", str(tmp_path)],
                         env=env, cwd=REFERENCE, capture_output=True, text=True, timeout=600)
    tail = run.stdout.strip().splitlines()[-1] if run.stdout.strip() else run.stderr[-500:]
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-1000:]
    assert f"{REFERENCE_TEST_FILES[test_file]} passed" in tail and "failed" not in tail and "error" not in tail, tail
