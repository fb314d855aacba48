"""Fixed (per-launch) versus per-iteration cost of the persistent fused sampler kernel."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import bench
dev = torch.device('cuda:0')
w = bench.WORKLOADS["C2"]
gen, noise, sampling, net = bench.build_generator(w, dev, 0, 1024, False)
gen.fused_score_network = True
with torch.no_grad():
    gen._prepare(dev); gen._begin_call(dev)
    start = gen.initialize(1024, dev)
    loop = bench.FusedLoop(gen, start, 1000)
    loop.advance(10)
    for n in (1, 2, 5, 10, 50):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        host, gpu = [], []
        for rep in range(20):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev0.record()
            loop.advance(n)
            ev1.record()
            while not ev1.query():
                pass
            host.append((time.perf_counter() - t0) * 1e6)
            gpu.append(ev0.elapsed_time(ev1) * 1e3)
        host.sort(); gpu.sort()
        print(f"n={n:3d}  host median {host[10]:7.1f} us   event median {gpu[10]:7.1f} us")
