"""Host-side and device-side cost of one mdx_mlp_pc_sample call (C2 shape) as a function of the iterations per call."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
dev = torch.device("cuda:0")
w = bench.WORKLOADS["C2"]
gen, *_ = bench.build_generator(w, dev, 0, w["batch"], False)
gen.fused_score_network = True
with torch.no_grad():
    gen._prepare(dev); gen._begin_call(dev)
    start = gen.initialize(w["batch"], dev)
    loop = bench.FusedLoop(gen, start, 1000)
    loop.advance(20)
    for n in (1, 5, 20, 100):
        torch.cuda.synchronize()
        host, total = [], []
        for rep in range(20):
            loop.remaining = 900
            t0 = time.perf_counter()
            loop.advance(n)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append(t1 - t0); total.append(t2 - t0)
        host.sort(); total.sort()
        print(f"n={n:4d}: host call {host[len(host)//2]*1e6:7.1f} us, call + wait {total[len(total)//2]*1e6:7.1f} us")
