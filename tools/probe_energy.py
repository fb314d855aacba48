"""What the LDS fragment reads and the L2 -> LDS weight stream cost in JOULES, in the synthetic kernel with the edge chain's structure
(tools/mfma_shape_probe.hip, 16x16x32 shape, random operands): the full kernel against three ablations -- no LDS fragment reads
(eight fragment pairs read once and cycled), no weight-stream requests inside the loop, neither -- each timed interleaved with the
others and then run back to back for --seconds while the card's power sensor is polled (tools/power_sampler.py).
energy per launch = mean power x time per launch.

Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/_ablate/libprobe.so tools/mfma_shape_probe.hip"""
import argparse
import ctypes as C
import json
import os
import statistics
import threading
import time

import torch

from power_sampler import PowerSampler

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=6.0)
ap.add_argument("--grid", type=int, default=256)
ap.add_argument("--chunks", type=int, default=73 * 24)
args = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "tools", "_ablate", "libprobe.so"))
lib.probe_launch.restype = C.c_int
lib.probe_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]
dev = torch.device("cuda:0")
grid, chunks, image_chunks = args.grid, args.chunks, 73
torch.manual_seed(0)
w = torch.randn(image_chunks * 32 * 256, device=dev) * 0.06 * 65536.0
hi = w.half()
lo = (w - hi.float()).half()
image = torch.stack([hi.view(image_chunks * 16, 512), lo.view(image_chunks * 16, 512)], 1).contiguous()
x0 = (torch.randn(grid * 256 * 128, device=dev) * 20.0).contiguous()
out = torch.zeros(grid * 256, device=dev)
stream = torch.cuda.current_stream().cuda_stream
neg_c, k = -2.0 ** -22, 2.0 ** 16
NAMES = {1: "full (16x16x32, four wavefronts)", 101: "no LDS fragment reads", 102: "no weight-stream requests", 103: "neither"}


def launch(shape):
    rc = lib.probe_launch(shape, image.data_ptr(), image_chunks, x0.data_ptr(), chunks, out.data_ptr(), grid, neg_c, k, stream)
    assert rc == 0, rc


for shape in NAMES:
    launch(shape)
torch.cuda.synchronize()
assert bool(torch.isfinite(out).all())
t0 = time.time()
while time.time() - t0 < 1.0:
    for shape in NAMES:
        launch(shape)
    torch.cuda.synchronize()
times = {s: [] for s in NAMES}
order = list(NAMES)
for r in range(9):
    for shape in (order if r % 2 == 0 else order[::-1]):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            launch(shape)
        b.record()
        torch.cuda.synchronize()
        times[shape].append(a.elapsed_time(b) * 1000 / 3)
res = {"grid": grid, "chunks": chunks, "executed_flop_per_launch": 2.0 * grid * 128 * 32 * 256 * chunks * 3}
for shape, name in NAMES.items():
    sampler = PowerSampler()
    thread = threading.Thread(target=sampler.run)
    launch(shape)
    torch.cuda.synchronize()
    thread.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < args.seconds:
        for _ in range(20):
            launch(shape)
        torch.cuda.synchronize()
        n += 20
    elapsed = time.perf_counter() - t0
    sampler.stop = True
    thread.join()
    summary = sampler.summary()
    us = elapsed / n * 1e6
    res[name] = {"interleaved_median_us": round(statistics.median(times[shape]), 1), "back_to_back_us": round(us, 1),
                 "power_W_mean": summary["power_W_mean"], "power_W_max": summary["power_W_max"], "sclk_MHz_mean": summary["sclk_MHz_mean"],
                 "joules_per_launch": round(summary["power_W_mean"] * us * 1e-6, 3) if summary["power_W_mean"] else None,
                 "power_cap_W": summary["power_cap_W"]}
    time.sleep(2.0)
print(json.dumps(res))
