"""Driver of tools/mfma_shape_probe.hip: the same synthetic chunk loop on v_mfma_f32_32x32x16_f16 and on v_mfma_f32_16x16x32_f16,
interleaved rounds in one process, random operands.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/_ablate/libprobe.so
tools/mfma_shape_probe.hip"""
import ctypes as C
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "tools", "_ablate", "libprobe.so"))
lib.probe_launch.restype = C.c_int
lib.probe_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]
dev = torch.device("cuda:0")
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 256
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 73 * 24
torch.manual_seed(0)
image_chunks = 73
w = torch.randn(image_chunks * 32 * 256, device=dev) * 0.06 * 65536.0            # (as the shipped kernel: 2^16 W in the image)
hi = w.half()
lo = (w - hi.float()).half()
image = torch.stack([hi.view(image_chunks * 16, 512), lo.view(image_chunks * 16, 512)], 1).contiguous()   # [frag][hi|lo][512 halfs]
x0 = (torch.randn(grid * 256 * 128, device=dev) * 20.0).contiguous()
out = torch.zeros(grid * 256, device=dev)
stream = torch.cuda.current_stream().cuda_stream
neg_c, k = -2.0 ** -22, 2.0 ** 16


def launch(shape):
    rc = lib.probe_launch(shape, image.data_ptr(), image_chunks, x0.data_ptr(), chunks, out.data_ptr(), grid, neg_c, k, stream)
    assert rc == 0, rc


SHAPES = (0, 1, 2, 3)
times = {k: [] for k in SHAPES}
for shape in SHAPES:
    launch(shape)
torch.cuda.synchronize()
print("finite:", bool(torch.isfinite(out).all()), float(out.abs().mean()))
import time
t0 = time.time()
while time.time() - t0 < 1.0:
    for shape in SHAPES:
        launch(shape)
    torch.cuda.synchronize()
for r in range(9):
    for shape in (SHAPES if r % 2 == 0 else SHAPES[::-1]):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            launch(shape)
        b.record()
        torch.cuda.synchronize()
        times[shape].append(a.elapsed_time(b) * 1000 / 3)
flops = 2.0 * grid * 128 * 32 * 256 * chunks * 3
res = {"grid": grid, "chunks": chunks}
for shape, name in ((0, "32x32x16"), (1, "16x16x32"), (2, "16x16x32, 8 wavefronts x 16 columns"),
                    (3, "16x16x32, 8 wavefronts, reduction dimension split per SIMD pair")):
    med = statistics.median(times[shape])
    res[name] = {"median_us": round(med, 1), "min_us": round(min(times[shape]), 1), "executed_tflops": round(flops / med / 1e6, 1)}
print(json.dumps(res))
