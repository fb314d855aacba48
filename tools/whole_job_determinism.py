"""BASELINE configs[2]'s WHOLE job (512 structures, production-size EGNN, T = 1000, M = 2, device Philox) run three times from
fresh generator objects with one seed -- hipGraph loop twice, eager launches once -- and compared bit for bit: the product's
output is a function of (seed, call index) alone, not of the launch mode, the run or the f16-range watch's host timing.
    python tools/whole_job_determinism.py [T] > gpurun_out/whole_job_determinism.json"""
import hashlib
import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
import nets  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import LangevinGenerator  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
    PredictorCorrectorSamplingParameters  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")
noise_kw, sampling_kw, _ = cases.C3_SHAPE
runs = []
for mode in ("graph", "graph", "eager"):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npar = NoiseParameters(**dict(noise_kw, total_time_steps=T))
        spar = PredictorCorrectorSamplingParameters(**dict(sampling_kw), rng_mode="device", seed=2025, use_hip_graph=mode == "graph")
    gen = LangevinGenerator(npar, spar, nets.egnn_c3_net(1).to(dev))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        out = gen.sample(512, dev)
    torch.cuda.synchronize()
    x, a = out.X.cpu(), out.A.cpu()
    runs.append(dict(mode=mode, seconds=round(time.perf_counter() - t0, 2), f16_range_fallbacks=gen.f16_range_fallbacks,
                     sha256_X=hashlib.sha256(x.numpy().tobytes()).hexdigest(), sha256_A=hashlib.sha256(a.numpy().tobytes()).hexdigest(),
                     all_unmasked=bool((a == 0).all()), inside_unit_cell=bool(((x >= 0) & (x < 1)).all())))
    print(json.dumps(runs[-1]), file=sys.stderr, flush=True)
    del gen
same = all(r["sha256_X"] == runs[0]["sha256_X"] and r["sha256_A"] == runs[0]["sha256_A"] for r in runs)
print(json.dumps(dict(workload="C3: 512 structures x 64 atoms, EGNN 4 x 256 x 4, T = %d, M = 2, device Philox, split-f16 edge chain" % T,
                      runs=runs, bit_identical=same)))
assert same
