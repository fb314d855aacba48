"""Per-iteration time of the persistent sampler with parts switched off (options MDX_MLP_SAMPLE_DIAG_NO_FORWARD /
_NO_UPDATE), with pre-drawn or in-kernel noise.  Needs a diagnostics build of the library
(`make -C .../csrc -B DIAG=1`); the release build answers MDX_ERR_UNSUPPORTED to these options."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from diffusion_for_multi_scale_molecular_dynamics_amd import _hip, kernels
if len(sys.argv) > 1:                      # a diagnostics build under another name (tools/_ablate/libmdx_diag.so)
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda:0")
w = bench.WORKLOADS["C2"]
gen, *_ = bench.build_generator(w, dev, 0, w["batch"], False)
gen.fused_score_network = True
with torch.no_grad():
    gen._prepare(dev); gen._begin_call(dev)
    start = gen.initialize(w["batch"], dev)
    sched, pack = gen._prepare(dev), gen.fused_pack(dev)
    for predrawn in (True, False):
        for skip in (0, 1, 2, 3):
            options = (256 if skip & 1 else 0) | (512 if skip & 2 else 0)
            comp = type(start)(*[t.clone() for t in start])
            def launch(n):
                kernels.mlp_pc_sample(sched, pack, gen._flags(True), 1, False, 900, n, gen._rng(0), comp.A, comp.X, comp.L,
                                      gen._status, workspace=gen._noise_workspace if predrawn else None, options=options)
            launch(100)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); a.record(); launch(400); b.record(); torch.cuda.synchronize()
            t400 = a.elapsed_time(b)
            torch.cuda.synchronize(); a.record(); launch(100); b.record(); torch.cuda.synchronize()
            t100 = a.elapsed_time(b)
            print(f"predrawn={predrawn} skip={skip}: {(t400 - t100) / 300 * 1e3:.2f} us / iteration")
