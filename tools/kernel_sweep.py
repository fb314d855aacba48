"""Achieved HBM GB/s of the hand-written kernels at sizes large enough to leave the launch-latency regime."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import bench
from diffusion_for_multi_scale_molecular_dynamics_amd import kernels
from diffusion_for_multi_scale_molecular_dynamics_amd._hip import MDX_PREDICTOR, MDX_CORRECTOR, PcFlags, Rng
dev = torch.device('cuda:0')
rows = []
# P1 flat
for n in (1 << 16, 1 << 20, 1 << 24, 1 << 27):
    x, s, z = (torch.rand(n, device=dev) for _ in range(3)); out = torch.empty_like(x)
    ms = bench.time_launches(lambda: kernels.relative_coordinates_update(x, s, z, 0.01, 0.1, 0.05, out=out), dev, 20)
    rows.append(("coords_update_kernel<4> (P1, host RNG)", n, 16 * n, ms))
# fused update, device RNG
sched = kernels.noise_schedule_build(1000, "linear", 1e-5, 1e-4, 0.2, 2.5e-8, 2, dev)
for B, N in ((1024, 8), (512, 64), (16384, 64), (262144, 64)):
    C = 2
    a = torch.full((B, N), C - 1, dtype=torch.int64, device=dev); x = torch.rand(B, N, 3, device=dev)
    lat = torch.tensor([10.86] * 3 + [0.0] * 3, device=dev).repeat(B, 1)
    logits = torch.randn(B, N, C, device=dev); logits[..., -1] = -torch.inf
    score = torch.randn(B, N, 3, device=dev); a_out, x_out = torch.empty_like(a), torch.empty_like(x)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    for mode, name, bpa in ((MDX_PREDICTOR, "pc_step_kernel predictor (P2+P1, device RNG)", 52 + 4 * C), (MDX_CORRECTOR, "pc_step_kernel corrector (P1, device RNG)", 36)):
        fl = PcFlags(0, 0, 1, 1 if mode == MDX_PREDICTOR else 0, 1e-8)
        def launch():
            kernels.pc_step_update(sched, mode, 500, None, fl, a if mode == MDX_PREDICTOR else None, x, lat, logits if mode == MDX_PREDICTOR else None, score, None, None, None, None, None, Rng(1, 0, 3, 0), a_out if mode == MDX_PREDICTOR else None, x_out, lat, st)
        ms = bench.time_launches(launch, dev, 20)
        rows.append((name, B * N, bpa * B * N, ms))
# N1
for B, N in ((512, 64), (8192, 64), (256, 216), (4096, 216)):
    w = dict(n_atoms=N, cell=10.86)
    m = bench.time_radius_graph(B, w, dev, launches=10)
    rows.append((f"radius_graph_kernel<fill> N={N} (deg {m['edges_per_atom']:.1f})", B * N, m["bytes"], m["ms"]))
print("| kernel | atoms (or floats) | algorithmic MB | us / launch | GB/s | % of 8 TB/s |")
print("|---|---|---|---|---|---|")
for name, n, b, ms in rows:
    gbs = b / (ms * 1e-3) / 1e9
    print(f"| {name} | {n} | {b/1e6:.2f} | {ms*1e3:.1f} | {gbs:.0f} | {100*gbs/8000:.1f} |")
