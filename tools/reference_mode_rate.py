"""The C3 job in the REFERENCE's mode -- draws from torch's CPU generator in the reference's order, uploaded over PCIe every step,
eager launches (what a `sampling:` block without this package's fast-mode keys runs; the parity mode) -- beside the default mode
of bench.py (device Philox, hipGraph): milliseconds per iteration over K iterations of a 512-structure batch.
    python tools/reference_mode_rate.py [K] > gpurun_out/r05_reference_mode_rate.json"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
    PredictorCorrectorSamplingParameters  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
w = bench.WORKLOADS["C3"]
out = {}
for mode, kw in (("reference (host draws, PCIe uploads, eager launches)", dict(rng_mode="reference", use_hip_graph=False)),
                 ("device Philox, eager launches", dict(rng_mode="device", seed=1, use_hip_graph=False)),
                 ("device Philox, hipGraph (bench.py's mode)", dict(rng_mode="device", seed=1, use_hip_graph=True))):
    torch.manual_seed(bench.NET_SEED)
    net = bench.egnn_experiment(w["num_atom_types"]).eval().to(dev)
    noise = NoiseParameters(**dict(w["noise"], total_time_steps=K + 5))
    sampling = PredictorCorrectorSamplingParameters(
        number_of_atoms=w["n_atoms"], num_atom_types=w["num_atom_types"], number_of_samples=w["batch"],
        number_of_corrector_steps=w["M"], atom_type_greedy_sampling=w["greedy"], one_atom_type_transition_per_step=w["one"],
        use_fixed_lattice_parameters=True, cell_dimensions=[w["cell"]] * 3, **kw)
    gen = LangevinGenerator(noise, sampling, net)
    torch.manual_seed(7)
    with torch.no_grad():
        gen.sample(w["batch"], dev)                 # warm-up: a whole (K + 5)-iteration job
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gen.sample(w["batch"], dev)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / (K + 5)
    out[mode] = dict(ms_per_iteration=round(ms, 3), structures_per_s_at_T_1000=round(w["batch"] / ms, 3))
    del gen, net
floats = w["batch"] * w["n_atoms"] * (2 + 1 + 3 + 3 + 3) + 3 * w["batch"] * 6
out["host_to_device_bytes_per_iteration_reference_mode"] = 4 * floats
print(json.dumps(out, indent=1))
