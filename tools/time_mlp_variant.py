"""Time one whole C2 trajectory (MLP template, B = 1024, T = 1000) of the persistent sampler through a chosen instantiation.
    python tools/time_mlp_variant.py [--lib PATH] [--options N]     (options: MDX_MLP_SAMPLE_* bits; 128 = padded family)"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--options", type=int, default=128)
args = ap.parse_args()
from diffusion_for_multi_scale_molecular_dynamics_amd import _hip  # noqa: E402
if args.lib:
    _hip.LIB_PATH = os.path.abspath(args.lib)
import bench  # noqa: E402

dev = torch.device("cuda:0")
w = bench.WORKLOADS["C2"]
gen, noise, sampling, net = bench.build_generator(w, dev, 0, 1024, False)
gen.fused_score_network = True
gen.fused_sampler_options = args.options
with torch.no_grad():
    gen._prepare(dev)
    gen._begin_call(dev)
    start = gen.initialize(1024, dev)
    times = []
    for _ in range(5):
        loop = bench.FusedLoop(gen, start, 1000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop.advance(1000)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
print(f"{args.lib or 'tree'} options {args.options}: trajectory ms {min(times[1:]):.3f} -> {1024 / min(times[1:]) * 1e3:.0f} structures/s")
