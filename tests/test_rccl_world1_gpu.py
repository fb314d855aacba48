"""The RCCL code path, executed on the one GPU there is (SURVEY 8e / VERDICT round 3, item 3).

Multi-GPU runs are the driver's; what can be done on a one-GPU box is to make sure the first 8-GPU run is not the first
execution of the distributed branch: `torchrun --nproc-per-node 1` sets RANK / WORLD_SIZE = 1, and under it bench.py, the sharded
batch driver, the adaptive corrector's cross-rank means and the CLI all take their `torch.distributed` branch with the "nccl"
(= RCCL) backend on device tensors -- group set-up with a device id, the packed uint8 all_gather_into_tensor, the MAX / SUM
all-reduces, barriers, teardown.  Each case is a CHILD process started with subprocess (the ranks must be fresh processes), and
is checked through what it prints / writes.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _torchrun(args, port, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    return subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, cwd=ROOT)


def test_bench_distributed_branch_on_one_gpu(cuda):
    """bench.py exactly as the driver launches it for N > 1, with N = 1: rank 0's JSON line reports one GPU, a gather that took
    time on the device (the job's one collective really ran), and bench.py itself asserts that the gathered block equals the
    rank's local composition."""
    run = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "C2", "--steps", "50", "--warmup", "5",
                     "--no-cpu-baseline"], port=29571)
    assert run.returncode == 0, run.stderr[-4000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["config"]["global_batch"] == line["config"]["batch_per_gpu"] == 1024
    assert line["config"]["gather_ms"] > 0 and line["config"]["collective_backend"] == "nccl"
    assert line["value"] > 0 and line["scaling"] == "weak"
    # the per-rank record travelled through its own RCCL all-gather: this rank's own timings, its card's sensor readings
    (mine,) = line["per_rank"]
    assert mine["rank"] == 0 and line["slowest_rank"] == 0 and mine["ms_per_step"] > 0 and mine["trajectory_ms"] > 0
    assert abs(mine["ms_per_step"] - line["ms_per_step"]) < 1e-3 * line["ms_per_step"] + 1e-4     # N = 1: MAX over ranks = own
    assert line["n1_equivalent"]["spread"] == 0.0
    assert abs(line["n1_equivalent"]["value_per_gpu_fastest_rank"] - line["value"]) < 0.02 * line["value"]
    for key in ("sclk_mhz_mean", "power_w_mean", "power_cap_w"):      # None only if the box hides the sensor files
        assert mine[key] is None or mine[key] > 0
    assert "also_measured" not in line                                # (auto: only the default workload at N = 1)


_WORKER = r'''
import os, sys, warnings
import torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import cases, nets
from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator
from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
    PredictorCorrectorSamplingParameters
from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters
from diffusion_for_multi_scale_molecular_dynamics_amd.sampling.diffusion_sampling import (
    create_batch_of_samples, create_batch_of_samples_sharded)
from diffusion_for_multi_scale_molecular_dynamics_amd.utils.batch_statistics import global_means

assert os.environ["WORLD_SIZE"] == "1" and os.environ["RANK"] == "0"
device = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
torch.cuda.set_device(device)
dist.init_process_group(backend="nccl", device_id=device)               # RCCL
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1


def generator(name):
    noise_kw, sampling_kw, netf = cases.TRAJECTORIES[name]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        spar = PredictorCorrectorSamplingParameters(**dict(sampling_kw, number_of_samples=5, sample_batchsize=2),
                                                    rng_mode="device", seed=606)
    torch.manual_seed(1234)
    net = nets.fake_net(spar.num_atom_types) if netf is None else netf(None)
    return LangevinGenerator(NoiseParameters(**noise_kw), spar, net.to(device)), spar


for name in ("traj_fake_c3_m2", "traj_mlp_c3"):
    gen, spar = generator(name)
    with torch.no_grad():
        sharded = create_batch_of_samples_sharded(gen, spar, device)   # ONE packed all_gather_into_tensor on device rows
    gen, spar = generator(name)
    with torch.no_grad():
        local = create_batch_of_samples(gen, spar, device)
    for got, want in zip(sharded["original_axl"], local["original_axl"]):
        assert got.is_cuda and torch.equal(got, want), name
    assert torch.equal(sharded["cartesian_positions"], local["cartesian_positions"])

# the adaptive corrector's batch statistics: one 4-scalar SUM all-reduce on device tensors
g = torch.Generator().manual_seed(7)
a, b = torch.rand(10, generator=g).to(device), torch.rand(10, 8, generator=g).to(device)
ma, mb = global_means(a, b, across_ranks=True)
assert ma.is_cuda and torch.allclose(ma, a.mean(), rtol=1e-6) and torch.allclose(mb, b.mean(), rtol=1e-6)

# MAX all-reduce + barrier, as bench.py's timing protocol uses them
t = torch.tensor([3.25], dtype=torch.float64, device=device)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t[0]) == 3.25
dist.barrier()
dist.destroy_process_group()
print("worker ok")
'''


def test_sharded_driver_and_batch_statistics_over_rccl(cuda, tmp_path):
    script = tmp_path / "rccl_worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    run = _torchrun([str(script)], port=29572)
    assert run.returncode == 0 and "worker ok" in run.stdout, (run.stdout[-2000:], run.stderr[-4000:])


def test_cli_under_torchrun_writes_the_whole_run(cuda, tmp_path):
    """sample_diffusion under torchrun (one rank): the process group is initialised (nccl), the sub-batches go through the
    sharded driver's gather, and samples.pt / trajectories.pt are those of a plain single-process run with the same seed."""
    import yaml
    noise = dict(total_time_steps=4, sigma_min=0.0001, sigma_max=0.2, schedule_type="linear", corrector_step_epsilon=2.5e-8)
    sampling = dict(algorithm="predictor_corrector", num_atom_types=1, number_of_atoms=8, sample_batchsize=2,
                    spatial_dimension=3, number_of_corrector_steps=1, number_of_samples=5, record_samples=True,
                    record_samples_corrector_steps=True, use_fixed_lattice_parameters=True, cell_dimensions=[5.43, 5.43, 5.43],
                    rng_mode="device", seed=321)
    score_network = dict(architecture="mlp", number_of_atoms=8, num_atom_types=1, n_hidden_dimensions=2,
                         hidden_dimensions_size=32, relative_coordinates_embedding_dimensions_size=16,
                         noise_embedding_dimensions_size=8, time_embedding_dimensions_size=8,
                         atom_type_embedding_dimensions_size=1, lattice_parameters_embedding_dimensions_size=1)
    (tmp_path / "config.yaml").write_text(yaml.safe_dump(dict(noise=noise, sampling=sampling, elements=["Si"],
                                                              model=dict(score_network=score_network))))
    common = ["--config", str(tmp_path / "config.yaml"), "--random_init_seed", "5", "--device", "cuda"]
    run = _torchrun(["-m", "diffusion_for_multi_scale_molecular_dynamics_amd.sample_diffusion"] + common +
                    ["--output", str(tmp_path / "dist")], port=29573)
    assert run.returncode == 0, run.stderr[-4000:]
    from diffusion_for_multi_scale_molecular_dynamics_amd import sample_diffusion
    sample_diffusion.main(common + ["--output", str(tmp_path / "single")])
    a = torch.load(tmp_path / "dist" / "samples.pt", weights_only=False)
    b = torch.load(tmp_path / "single" / "samples.pt", weights_only=False)
    assert torch.equal(a["original_axl"].X.cpu(), b["original_axl"].X.cpu())
    assert torch.equal(a["original_axl"].A.cpu(), b["original_axl"].A.cpu())
    ta = torch.load(tmp_path / "dist" / "trajectories.pt", weights_only=False)
    tb = torch.load(tmp_path / "single" / "trajectories.pt", weights_only=False)
    assert len(ta["predictor_step"]) == len(tb["predictor_step"]) == 3 * 4
    for x, y in zip(ta["predictor_step"], tb["predictor_step"]):
        assert x["time_step_index"] == y["time_step_index"] and torch.equal(x["composition_im1"].X, y["composition_im1"].X)
