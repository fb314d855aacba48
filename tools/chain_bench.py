"""Time the fused MFMA edge-chain kernel at a workload's shape (default C3: E ~ 819 k edges, H = 256, 4 message + 5
coordinate layers) in both arithmetic modes, next to the same chain as per-layer PyTorch linear + SiLU calls."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd import kernels  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=512 * 64)
ap.add_argument("--degree", type=int, default=25)
ap.add_argument("--hidden", type=int, default=256)
ap.add_argument("--launches", type=int, default=5)
ap.add_argument("--modes", default="f32,f16x3,library")
ap.add_argument("--n-msg", type=int, default=4)
ap.add_argument("--n-crd", type=int, default=5)
ap.add_argument("--eager", action="store_true", help="time eager launches with HIP events (no hipGraph: safe under rocprofv3 --pmc)")
ap.add_argument("--piece-sums", action="store_true", help="aggregate the messages inside the kernel (piece sums)")
ap.add_argument("--stamps", action="store_true", help="with a -DMDX_CHAIN_STAMPS build: print the stamped intervals")
ap.add_argument("--clocks", action="store_true", help="with a -DMDX_CHAIN_STAMPS=2 build: the shader clock during a launch")
ap.add_argument("--rows", action="store_true", help="also time the row chain (mdx_mlp_chain_rows): --nodes rows, --n-crd layers + residual")
ap.add_argument("--zero-activations", action="store_true", help="zero projections, coordinates and biases: every activation of "
                "every layer is an exact zero while the weights stay random (what the same instruction stream costs without "
                "operand switching activity)")
ap.add_argument("--power", type=float, default=0.0, help="seconds of back-to-back launches per mode while a thread samples the "
                "card's power sensor, power cap and shader clock (hwmon / pp_dpm_sclk in sysfs; rocm-smi as the fallback)")
ap.add_argument("--lib", default=None, help="alternative libmdx_hip.so (ablation builds)")
args = ap.parse_args()
if args.lib:
    from diffusion_for_multi_scale_molecular_dynamics_amd import _hip
    _hip.LIB_PATH = os.path.abspath(args.lib)
dev = torch.device("cuda:0")
torch.manual_seed(0)


from power_sampler import PowerSampler  # noqa: E402

H, n_in, n_msg, n_crd = args.hidden, args.hidden, args.n_msg, args.n_crd
n_nodes, E = args.nodes, args.nodes * args.degree
lin0 = torch.nn.Linear(2 * n_in + 1, H).to(dev)
msg = [torch.nn.Linear(H, H).to(dev) for _ in range(n_msg)]
crd = [torch.nn.Linear(H, H).to(dev) for _ in range(n_crd)]
out = torch.nn.Linear(H, 1, bias=False).to(dev)
src = torch.arange(n_nodes, device=dev).repeat_interleave(args.degree)
dst = (src // 64) * 64 + torch.randint(0, 64, (E,), device=dev)
edges = torch.stack([src, dst], 1).contiguous()
proj = torch.randn(n_nodes, 2 * H, device=dev)
coord = torch.rand(n_nodes, 6, device=dev)
if args.zero_activations:
    with torch.no_grad():
        proj.zero_(); coord.zero_(); lin0.bias.zero_()
        for layer in msg + crd:
            layer.bias.zero_()
flops = 2.0 * E * H * H * (n_msg + n_crd)
res = {"edges": E, "hidden": H, "layers": n_msg + n_crd, "algorithmic_gflop": flops / 1e9}
with torch.no_grad():
    for mode in args.modes.split(","):
        if mode == "library":
            radial = torch.rand(E, device=dev)

            def launch():
                x = kernels.egnn_message_input(proj, edges, radial, lin0.bias, lin0.weight[:, 2 * n_in].contiguous())
                for layer in msg:
                    x = torch.nn.functional.silu(torch.nn.functional.linear(x, layer.weight, layer.bias))
                for layer in crd:
                    x = torch.nn.functional.silu(torch.nn.functional.linear(x, layer.weight, layer.bias))
                return x
        else:
            pack = kernels.EdgeChainPack(lin0, msg, crd, out, input_size=n_in, precision=mode)

            stamps = torch.zeros(8192, dtype=torch.int32, device=dev) if args.stamps or args.clocks else None

            def launch(pack=pack):
                return kernels.egnn_edge_chain(pack, proj, coord, edges, status=stamps, piece_sums=args.piece_sums)
            if args.clocks:
                for _ in range(8):
                    launch()                        # (each launch restarts the stamp list: the last one is read)
                torch.cuda.synchronize()
                raw = stamps.view(torch.int64).cpu().numpy()
                val = {int(r >> 48): int(r & ((1 << 48) - 1)) for r in raw if r >> 48}
                # phases of workgroup 0's edge tiles: 9 tile start, 10 first layer built, 30 / 31 around the aggregation
                seq = [(int(r >> 48), int(r & ((1 << 48) - 1))) for r in raw if (r >> 48) in (9, 10, 30, 31)]
                phase = {}
                for (a, ta), (b, tb) in zip(seq, seq[1:]):
                    phase.setdefault(f"{a}->{b}", []).append(tb - ta)
                phases = {k: int(sum(v) / len(v)) for k, v in phase.items()}
                cyc, ref = val[22] - val[20], val[23] - val[21]
                res[mode] = {"shader_cycles": cyc, "refclk_ticks_100MHz": ref, "us": ref / 100.0,
                             "shader_clock_GHz": round(cyc / (ref * 10.0), 3), "phase_cycles": phases}
                if 24 in val:       # -DMDX_CHAIN_STAMPS=3: issue cost of the weight-stream requests of wavefront 0
                    res[mode].update(requests=val[25], cycles_per_request=round(val[24] / max(val[25], 1), 1))
                continue
            if args.stamps:
                launch(); launch()
                torch.cuda.synchronize()
                raw = stamps.view(torch.int64).cpu().numpy()
                ids, t = raw >> 48, raw & ((1 << 48) - 1)
                keep = ids > 0
                ids, t = ids[keep], t[keep]
                import collections
                print("first 60 stamps (id, delta to previous in cycles of s_memtime):")
                print([(int(i), int(t[k] - t[k - 1]) if k else 0) for k, i in enumerate(ids[:60])])
                seq = [(int(ids[k - 1]), int(ids[k]), int(t[k] - t[k - 1])) for k in range(1, len(ids))]
                print("(3->4) second-half-of-tile series:", [d_ for a, b, d_ in seq if (a, b) == (3, 4)][:160])
                print("(4->1) first-half series:", [d_ for a, b, d_ in seq if (a, b) == (4, 1)][:160])
                # steady state: per stamp-id transition statistics
                d = collections.defaultdict(list)
                for k in range(1, len(ids)):
                    d[(int(ids[k - 1]), int(ids[k]))].append(int(t[k] - t[k - 1]))
                for key, v in sorted(d.items()):
                    v = sorted(v)
                    print(key, "n", len(v), "median", v[len(v) // 2], "mean", sum(v) / len(v), "max", v[-1])
                continue
        if args.power > 0:
            import threading
            import time
            launch()
            torch.cuda.synchronize()
            sampler = PowerSampler()
            thread = threading.Thread(target=sampler.run)
            thread.start()
            t0, n = time.perf_counter(), 0
            while time.perf_counter() - t0 < args.power:
                for _ in range(20):
                    launch()
                torch.cuda.synchronize()
                n += 20
            elapsed = time.perf_counter() - t0
            sampler.stop = True
            thread.join()
            res[mode] = dict(ms=round(elapsed / n * 1e3, 4), launches=n, **sampler.summary())
            time.sleep(2.0)                             # let the sensor's averaging window drain before the next mode
            continue
        if args.eager:
            launch()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.launches):
                launch()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / args.launches
        else:
            ms = bench.time_launches(launch, dev, args.launches)
        res[mode] = {"ms": round(ms, 4), "algorithmic_tflops": round(flops / ms / 1e9, 2)}
if args.rows:
    with torch.no_grad():
        x = torch.randn(n_nodes, H, device=dev)
        for mode in args.modes.split(","):
            if mode == "library":
                continue
            rpack = kernels.RowChainPack(crd, mode)
            ms = bench.time_launches(lambda rpack=rpack: kernels.mlp_chain_rows(rpack, x, residual=x), dev, args.launches)
            res.setdefault("rows", {})[mode] = {"ms": round(ms, 4), "rows": n_nodes, "layers": len(crd)}
print(json.dumps(res))
