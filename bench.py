#!/usr/bin/env python3
"""Benchmark of the MI355X sampling hot path: sampled structures / second for a T-step predictor-corrector run.

    python bench.py --gpus N --steps K --warmup W [--workload C2|C3|C4|C5]
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

A "step" is one sampler iteration -- score-network forward + fused predictor update, then M x (forward + fused
corrector update) -- over one batch of synthetic structures resident in HBM.  K steps are timed between
barrier + synchronize pairs, the MAX over ranks is taken, and the whole-job throughput is derived for the
workload's full trajectory:  value = structures / (T * ms_per_step + gather_ms).
Workloads: C3 = BASELINE configs[2] (Si 2x2x2, EGNN 4x256 with the radius graph, T=1000, M=2,
B=512 per GPU) is the default -- the largest single-GPU configuration and the one north_star's target is quoted on;
C2 = configs[1] (Si 1x1x1, MLP); C4/C5 are the other EGNN configurations.  Weak scaling: every rank samples its own
batch with Philox seed base+rank; the only collective is ONE all-gather of the final compositions (A, X, L packed
into one byte buffer per rank).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own ranks: the parent -- before anything has
touched the GPU -- runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process, relays
its output and exits with its code.  For the MLP workloads the job is a single 4 ms launch, so `value` comes from a
timed region that is one whole T-iteration trajectory whatever --steps says (`ms_per_step` is still the K-step figure).

The JSON line also carries
  roofline      the dominant hand-written kernel of the workload: algorithmic bytes per launch / average launch
                duration measured here with HIP events on the stream the kernel runs on, against 8 TB/s;
  cpu_baseline  the CPU oracle (oracle/, a restatement of the reference pinned to its golden vectors; kind "port")
                timed on this host's cores over a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from diffusion_for_multi_scale_molecular_dynamics_amd import kernels  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd._hip import MDX_PREDICTOR, Rng  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd.generators.constrained_langevin_generator import \
    ConstrainedLangevinGenerator  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd.generators.langevin_generator import (  # noqa: E402
    IterationLoop, LangevinGenerator)
from diffusion_for_multi_scale_molecular_dynamics_amd.generators.predictor_corrector_axl_generator import \
    PredictorCorrectorSamplingParameters  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics_amd.generators.sampling_constraint import SamplingConstraint  # noqa
from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.egnn_score_network import (  # noqa: E402
    EGNNScoreNetwork, EGNNScoreNetworkParameters)
from diffusion_for_multi_scale_molecular_dynamics_amd.models.score_networks.mlp_score_network import (  # noqa: E402
    MLPScoreNetwork, MLPScoreNetworkParameters)
from diffusion_for_multi_scale_molecular_dynamics_amd.noise_schedulers.noise_parameters import NoiseParameters  # noqa
from diffusion_for_multi_scale_molecular_dynamics_amd.sampling.diffusion_sampling import (  # noqa: E402
    pack_compositions, unpack_compositions)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BASE_SEED = 20250815
NET_SEED = 1234


def mlp_template(num_atom_types, n_atoms):
    # configuration_templates/diffusion_config_files/config_diffusion_mlp.yaml:41-53 with N=8, d=3
    return MLPScoreNetwork(MLPScoreNetworkParameters(
        number_of_atoms=n_atoms, num_atom_types=num_atom_types, n_hidden_dimensions=3, hidden_dimensions_size=64,
        relative_coordinates_embedding_dimensions_size=32, noise_embedding_dimensions_size=16,
        time_embedding_dimensions_size=16, atom_type_embedding_dimensions_size=1,
        lattice_parameters_embedding_dimensions_size=1, condition_embedding_size=64))


def egnn_experiment(num_atom_types, edge_builder=None):
    # experiments/.../Si_2x2x2/config_diffusion_egnn.yaml:44-60
    return EGNNScoreNetwork(EGNNScoreNetworkParameters(
        num_atom_types=num_atom_types, n_layers=4, coordinate_hidden_dimensions_size=256,
        coordinate_n_hidden_dimensions=4, message_hidden_dimensions_size=256, message_n_hidden_dimensions=4,
        node_hidden_dimensions_size=256, node_n_hidden_dimensions=4, coords_agg="mean", message_agg="mean",
        attention=False, normalize=False, residual=True, tanh=False, edges="radial_cutoff", radial_cutoff=7.5),
        edge_builder=edge_builder)


LINEAR = dict(sigma_min=1e-4, sigma_max=0.2, schedule_type="linear", corrector_step_epsilon=2.5e-8)
WORKLOADS = {
    "C2": dict(desc="Si_diffusion_1x1x1, MLP score net, 1000-step predictor-corrector, batch=1024 per GPU",
               n_atoms=8, num_atom_types=1, cell=5.43, net="mlp", batch=1024, M=1, greedy=True, one=True,
               noise=dict(total_time_steps=1000, sigma_min=1e-4, sigma_max=0.25, schedule_type="exponential"),
               graph=True, dominant="pc_step_kernel"),
    "C3": dict(desc="Si_diffusion_2x2x2, EGNN (4x256, rc 7.5) score net, 1000 steps, batch=512 per GPU",
               n_atoms=64, num_atom_types=1, cell=10.86, net="egnn", batch=512, M=2, greedy=False, one=False,
               noise=dict(total_time_steps=1000, **LINEAR), graph=True, dominant="radius_graph_kernel"),
    "C4": dict(desc="SiGe_diffusion_2x2x2 (two atom types), EGNN, 1000 steps, batch=512 per GPU",
               n_atoms=64, num_atom_types=2, cell=11.084, net="egnn", batch=512, M=2, greedy=True, one=True,
               noise=dict(total_time_steps=1000, **LINEAR), graph=True, dominant="radius_graph_kernel"),
    "C5": dict(desc="Si_diffusion_3x3x3 repaint (108 of 216 atoms pinned), EGNN, 2000 steps, batch=256 per GPU",
               n_atoms=216, num_atom_types=1, cell=16.29, net="egnn", batch=256, M=2, greedy=False, one=False,
               noise=dict(total_time_steps=2000, **LINEAR), graph=True, dominant="radius_graph_kernel",
               repaint=108, resampling=1),
}


def diamond_sites(n_cells):
    """Ideal diamond-cubic fractional coordinates of an n x n x n supercell (8 atoms per cell)."""
    base = torch.tensor([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                         [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]])
    cells = torch.cartesian_prod(*[torch.arange(n_cells)] * 3).float()
    return ((cells[:, None, :] + base[None]) / n_cells).reshape(-1, 3)


def build_generator(w, device, rank, batch, use_graph, edge_builder=None, resampling=0):
    torch.manual_seed(NET_SEED)
    net = (mlp_template(w["num_atom_types"], w["n_atoms"]) if w["net"] == "mlp"
           else egnn_experiment(w["num_atom_types"], edge_builder)).eval().to(device)
    noise = NoiseParameters(**w["noise"])
    sampling = PredictorCorrectorSamplingParameters(
        number_of_atoms=w["n_atoms"], num_atom_types=w["num_atom_types"], number_of_samples=batch,
        number_of_corrector_steps=w["M"], atom_type_greedy_sampling=w["greedy"],
        one_atom_type_transition_per_step=w["one"], use_fixed_lattice_parameters=True,
        cell_dimensions=[w["cell"]] * 3, rng_mode="device", seed=BASE_SEED, use_hip_graph=use_graph,
        repaint_resampling_steps=resampling if "repaint" in w else 0)
    if "repaint" in w:
        k = w["repaint"]
        constraint = SamplingConstraint(elements=["Si"], constrained_relative_coordinates=diamond_sites(3)[:k].clone(),
                                        constrained_atom_types=torch.zeros(k, dtype=torch.long))
        gen = ConstrainedLangevinGenerator(noise, sampling, net, constraint)
    else:
        gen = LangevinGenerator(noise, sampling, net)
    return gen, noise, sampling, net


class FusedLoop:
    """The sampler loop as launches of the persistent fused kernel (mdx_mlp_pc_sample): advance(n) = one launch."""

    def __init__(self, gen, start, starting_step_index):
        self.gen = gen
        self.composition = type(start)(A=start.A.clone(), X=start.X.clone(), L=start.L.clone())
        self.remaining = self.total = starting_step_index
        self.sched = gen._prepare(start.X.device)
        self.pack = gen.fused_pack(start.X.device)

    def advance(self, iterations):
        g, c = self.gen, self.composition
        while iterations > 0:        # more steps than the trajectory has: start another trajectory on the same buffers
            n = min(iterations, self.remaining)
            kernels.mlp_pc_sample(self.sched, self.pack, g._flags(True), g.number_of_corrector_steps,
                                  g.atom_type_transition_in_corrector, self.remaining, n, g._rng(0), c.A, c.X, c.L,
                                  g._status, workspace=g._noise_workspace, options=g.fused_sampler_options)
            iterations -= n
            self.remaining -= n
            if self.remaining == 0:
                self.remaining = self.total


def advance(loop, iterations, total):
    """loop.advance with wrap-around: a run longer than the trajectory continues with a fresh time index."""
    if isinstance(loop, FusedLoop):
        return loop.advance(iterations)
    while iterations > 0:
        n = min(iterations, loop.remaining)
        loop.advance(n)
        iterations -= n
        if loop.remaining == 0:
            kernels.index_set(loop.d_index, total - 1)
            loop.remaining = total


def time_fused_kernel(gen, loop, batch, w, device, iterations=None):
    """Duration of one launch of the persistent sampler over the whole trajectory (noise pre-pass + persistent kernel,
    HIP events on the launch stream) -- the launch shape of the timed run, so that the rocprofv3 per-launch average of the
    same command is directly comparable."""
    n, c, m = w["n_atoms"], w["num_atom_types"] + 1, w["M"]
    comp = type(loop.composition)(*[t.clone() for t in loop.composition])
    T = w["noise"]["total_time_steps"]
    iterations = T if iterations is None else iterations

    def launch():
        kernels.mlp_pc_sample(loop.sched, loop.pack, gen._flags(True), m, False, T, iterations,
                              gen._rng(0), comp.A, comp.X, comp.L, gen._status, workspace=gen._noise_workspace)
    launch()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    start.record()
    launch()
    stop.record()
    torch.cuda.synchronize(device)
    ms = start.elapsed_time(stop)
    # the bytes the un-fused algorithm moves per iteration: predictor 52+4C, each corrector 36 B/atom;
    # the persistent kernel keeps them in LDS and touches HBM only at the ends of the launch
    bytes_per_launch = batch * n * ((52 + 4 * c) + 36 * m) * iterations
    # The byte figure is the UN-fused algorithm's: the persistent kernel keeps that state in LDS by design, so
    # what actually bounds it is instruction issue / latency of one wavefront per SIMD.  Its compute side, for the record:
    # MACs of the network forward (every nn.Linear of the module, as the reference evaluates it) x forwards per iteration.
    net, pack = gen.axl_network, loop.pack
    hidden, n_hidden = net._hyper_params.hidden_dimensions_size, len(net.mlp_layers)
    if pack.folded is not None and pack.folded_out is not None and not gen.fused_sampler_options:
        # executed: folded input layer (padded to quads) + middle layers + folded output layer
        macs = (pack.folded.numel() - hidden) + hidden * hidden * (n_hidden - 2) + \
            (pack.folded_out.numel() - (n * c + n * 3 + 6))
    else:
        macs = sum(mod.in_features * mod.out_features for mod in net.modules() if isinstance(mod, torch.nn.Linear))
    tflops = 2.0 * macs * (1 + m) * batch * iterations / (ms * 1e-3) / 1e12
    return dict(kernel=f"mlp_pc_sample_kernel (persistent: {iterations} iterations of MLP forward + fused update per launch)",
                ms=ms, bytes=bytes_per_launch,
                compute=dict(tflops=round(tflops, 2), peak=MFMA_F32_PEAK_TFLOPS, frac_of_fp32_vector_peak=round(tflops / MFMA_F32_PEAK_TFLOPS, 4),
                             note="latency / issue bound: one wavefront per SIMD at B = 1024 (DESIGN.md section 4)"))


def time_launches(launch, device, launches):
    """Average GPU-side duration of `launch`: `launches` back-to-back launches are captured into one hipGraph (so the
    host's ctypes call overhead is out of the picture) and the replay is bracketed by HIP events on its stream.
    The figure includes the ~1.5 us same-stream dependency boundary between consecutive kernels."""
    for _ in range(3):
        launch()
    torch.cuda.synchronize(device)
    graph = torch.cuda.CUDAGraph()
    import gc
    gc.collect()                            # (torch >= 2.9 no longer does this on entry; no finaliser may call HIP in a capture)
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            for _ in range(launches):
                launch()
    finally:
        if was_enabled:
            gc.enable()
    graph.replay()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    start.record()
    graph.replay()
    stop.record()
    torch.cuda.synchronize(device)
    return start.elapsed_time(stop) / launches


def time_update_kernel(gen, batch, w, device, launches=200):
    """Average duration of one fused predictor-update launch (P2 + P1), HIP events on the launch stream."""
    n, c = w["n_atoms"], w["num_atom_types"] + 1
    sched = gen._prepare(device)
    a = torch.full((batch, n), c - 1, dtype=torch.int64, device=device)
    x = torch.rand(batch, n, 3, device=device)
    lat = torch.tensor([w["cell"]] * 3 + [0.0] * 3, device=device).repeat(batch, 1)
    logits = torch.randn(batch, n, c, device=device)
    logits[..., -1] = -torch.inf
    score = torch.randn(batch, n, 3, device=device)
    a_out, x_out = torch.empty_like(a), torch.empty_like(x)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    flags = gen._flags(True)
    rng = Rng(BASE_SEED, 0, w["M"] + 1, 0)
    T = w["noise"]["total_time_steps"]

    def launch():
        kernels.pc_step_update(sched, MDX_PREDICTOR, T // 2, None, flags, a, x, lat, logits, score, None, None, None,
                               None, None, rng, a_out, x_out, lat, status)
    ms = time_launches(launch, device, launches)
    bytes_per_launch = batch * n * (52 + 4 * c)         # predictor, device RNG
    return dict(kernel="pc_step_kernel (fused P2+P1 predictor update)", ms=ms, bytes=bytes_per_launch)


def time_radius_graph(batch, w, device, launches=50):
    """Average duration of one graph build as the sampler runs it (N1 + N2: mdx_egnn_radius_graph on relative coordinates, hit
    masks + emission = two launches) at the workload's size, and of its count / scan / fill form (three launches) beside it."""
    n = w["n_atoms"]
    x = torch.rand(batch, n, 3, device=device)
    lattice = torch.tensor([w["cell"]] * 3 + [0.0] * 3).repeat(batch, 1).to(device)
    capacity = batch * n * (n - 1)
    ms = {}
    for form, two in (("masks_emit", True), ("count_scan_fill", False)):
        keep = []

        def launch():
            keep.append(kernels.egnn_radius_graph(x, lattice, 2.2 * 7.5, 7.5, capacity, two_launches=two))
            del keep[:-2]
        ms[form] = time_launches(launch, device, launches)
    n_edges = int(keep[-1]["n_edges"].item())
    bytes_per_launch = batch * n * (12 + 8 + 8) + 16 * n_edges    # read X, write counts + offsets, write 16 B per edge
    return dict(kernel="mdx_egnn_radius_graph (N1 + N2 as the sampler runs it: egnn_graph_mask_kernel + egnn_graph_emit_kernel)",
                ms=ms["masks_emit"], bytes=bytes_per_launch, edges_per_atom=n_edges / (batch * n),
                extra=dict(launches_per_build=2, count_scan_fill_form_us=round(ms["count_scan_fill"] * 1e3, 2)))


MFMA_F32_PEAK_TFLOPS = 157.3    # dense fp32-input MFMA peak of MI355X (MI355X_MICROARCH.md; no xf32/TF32 on gfx950)


def time_edge_gemm(n_edges, device, hidden=256, launches=10):
    """`--egnn-precision library`: one hidden layer of the edge MLPs as the per-layer PyTorch path runs it,
    out[E,256] = SiLU(x[E,256] W^T + b) in fp32 -- torch.nn.functional.linear (torch's ROCm GEMM) + SiLU."""
    x = torch.randn(n_edges, hidden, device=device)
    wgt = torch.randn(hidden, hidden, device=device) / hidden ** 0.5
    bias = torch.zeros(hidden, device=device)
    ms = time_launches(lambda: torch.nn.functional.silu(torch.nn.functional.linear(x, wgt, bias)), device, launches)
    tflops = 2.0 * n_edges * hidden * hidden / (ms * 1e-3) / 1e12
    return dict(bound="mfma", achieved=round(tflops, 2), peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                frac=round(tflops / MFMA_F32_PEAK_TFLOPS, 4), traffic=None,
                kernel=f"PyTorch fp32 linear + SiLU {n_edges} x {hidden} x {hidden} (library kernels through torch; 36 of these "
                       "per network forward)", avg_launch_us=round(ms * 1e3, 2))


TRAFFIC_FILE = "traffic_r05.json"      # the latest committed PMC record of the edge chain's HBM traffic
MFMA_F16_PEAK_TFLOPS = 2500.0   # dense f16 / bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PF headline includes 2:1 sparsity)


def time_edge_chain(net, n_edges, n_nodes, device, launches=5):
    """The dominant kernel of the EGNN workloads: one launch of the hand-written fused edge chain (csrc/mdx_egnn_chain.hip)
    of the network's first graph layer over a synthetic sorted edge list of the workload's size, HIP events on the launch
    stream.  Algorithmic FLOPs = 2 E H^2 per H -> H layer (the per-edge MLPs); in the split-f16 mode the matrix
    cores execute three f16 products per algorithmic one, and `achieved` counts those against the f16 peak."""
    layer = net.egnn.graph_layers[0]
    pack = layer._edge_chain_pack()
    H, n_layers = pack.hidden, pack.c_struct.n_message_layers + pack.c_struct.n_coord_layers
    deg = max(1, n_edges // n_nodes)
    src = torch.arange(n_nodes, device=device).repeat_interleave(deg)
    dst = (src // 64) * 64 + torch.randint(0, 64, (src.numel(),), device=device)
    edges = torch.stack([src, dst], 1).contiguous()
    proj = torch.randn(n_nodes, 2 * H, device=device)
    coord = torch.rand(n_nodes, 6, device=device)
    pieces = bool(pack.piece_sums_ok)                 # the mode the network's forward uses (models/egnn.py)
    ms = time_launches(lambda: kernels.egnn_edge_chain(pack, proj, coord, edges, piece_sums=pieces), device, launches)
    flops = 2.0 * edges.shape[0] * H * H * n_layers
    split = pack.precision in ("f16x3", "f16x3_32x32")
    shape16 = pack.precision == "f16x3"
    executed = flops * (3 if split else 1) / (ms * 1e-3) / 1e12
    peak = MFMA_F16_PEAK_TFLOPS if split else MFMA_F32_PEAK_TFLOPS
    # HBM-side bytes per launch: NOT measured by this run -- counters need their own rocprofv3 --pmc passes -- but read from
    # the committed record of those passes (C3 shape only) and scaled by the edge count; `traffic_from` says so on the line
    traffic = traffic_from = None
    try:
        entry = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))[f"C3/edge_chain/{pack.precision}"]
        pmc_edges = int(entry.get("edges_per_launch", 819200))
        if abs(edges.shape[0] - pmc_edges) < 0.06 * pmc_edges and H == 256 and n_layers == 9:
            traffic = int(round(entry["bytes_per_launch"] * edges.shape[0] / pmc_edges))
            traffic_from = (f"profiles/{TRAFFIC_FILE}: rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, corrected per MI355X_MICROARCH.md) "
                            f"of this kernel at {pmc_edges} edges per launch = {entry['bytes_per_launch']} B, scaled to the "
                            f"{edges.shape[0]} edges of the launch timed here; not a measurement of this run")
    except (OSError, KeyError, ValueError):
        pass
    prec_index = {"f32": 0, "f16x3_32x32": 1, "f16x3": 2}[pack.precision]
    mfma = {"f32": "v_mfma_f32_32x32x2_f32", "f16x3_32x32": "split-f16: 3 x v_mfma_f32_32x32x16_f16",
            "f16x3": "split-f16: 3 x v_mfma_f32_16x16x32_f16"}[pack.precision]
    note = ""
    if split:
        note = ("peak = datasheet dense f16 MFMA rate at 2.4 GHz, counting the 3 executed products per algorithmic one.  Measured "
                "inside the kernel (s_memtime / s_memrealtime) the chip holds " +
                ("1.96-2.11 GHz on the 16x16x32 shape (2.32 GHz, same cycle count, on all-zero activations: power sets the clock); "
                 "the launch is bounded by its energy, not its cycles: 2.9 % fewer cycles per wavefront in round 4 (bit-identical "
                 "outputs) left the wall clock where it was (profiles/r04_chain_ablation.md); the card's sensor reads 1346 W "
                 "mean of its 1400 W cap during these launches, clock level 2141 MHz (profiles/r04_chain_power.json)"
                 if shape16 else "1.80-1.82 GHz on the 32x32x16 shape: power-bound (profiles/r03_chain_ablation.md)"))
    return dict(bound="mfma", achieved=round(executed, 2), peak=peak, unit="TFLOP/s", frac=round(executed / peak, 4),
                # the strict figure: ALGORITHMIC flops (one product per multiply-add of the reference's layers) against the
                # same peak -- `frac` counts the three executed f16 products per algorithmic one in the split mode
                frac_algorithmic=round(flops / (ms * 1e-3) / 1e12 / peak, 4),
                traffic=traffic, traffic_from=traffic_from,
                kernel=f"egnn_edge_chain_kernel<{H},{prec_index},{2 if pieces else 0}> ({mfma}"
                f" per product; {n_layers} fused H->H layers + per-node message sums, {edges.shape[0]} edges per launch; 4 launches per "
                f"network forward)",
                avg_launch_us=round(ms * 1e3, 2), algorithmic_flops_per_launch=flops,
                algorithmic_tflops=round(flops / (ms * 1e-3) / 1e12, 2), note=note)


def cpu_baseline(w, name, budget_s=15.0, resampling=0):
    """The CPU oracle on this host's cores over a bounded sample of the same workload."""
    import nets as test_nets
    from oracle import mdx_oracle
    cores = min(16, len(os.sched_getaffinity(0)))      # a 1-GPU box's CPU share; more threads only oversubscribe
    torch.set_num_threads(cores)
    from oracle.reference_sampler import OracleLangevinGenerator, PhiloxNoise
    mdx_oracle.build()
    batch = w["batch"] if w["net"] == "mlp" else 16
    gen, noise, sampling, net = build_generator(w, torch.device("cpu"), 0, batch, False,
                                                edge_builder=test_nets.oracle_edge_builder, resampling=resampling)
    constraint = None
    if "repaint" in w:
        k = w["repaint"]
        constraint = dict(constrained_relative_coordinates=diamond_sites(3)[:k].numpy(),
                          constrained_atom_types=torch.zeros(k, dtype=torch.long).numpy(), constrained_indices=None)
    ora = OracleLangevinGenerator(noise, sampling, net, constraint=constraint, noise=PhiloxNoise(BASE_SEED, 0))
    comp = ora.initialize(batch)
    T = noise.total_time_steps

    def run(i0, count):
        nonlocal comp
        t0 = time.perf_counter()
        for i in range(i0, i0 - count, -1):       # one sampler iteration = what OracleLangevinGenerator's loop does
            comp = ora.sample_from_noisy_composition(comp, i + 1, i)
        return time.perf_counter() - t0

    probe = 2 if w["net"] == "egnn" else 10
    t_probe = run(T - 1, probe)
    per_iter = t_probe / probe
    count = int(max(1, min(T - probe, budget_s / per_iter)))
    elapsed = run(T - 1 - probe, count)
    per_iter = elapsed / count
    value = batch / (T * per_iter)
    out = dict(value=value, unit="structures/s", cores=cores, kind="port",
               sample=f"{count} of {T} iterations of workload {name} at batch {batch} "
                      f"({elapsed:.1f} s, {per_iter * 1e3:.2f} ms/iteration), extrapolated to the {T}-step job")
    # the port against the REFERENCE ITSELF, both timed on one host (the reference cannot run on the GPU box): read from the
    # committed calibration (tools/time_reference.py --port), and only for the workloads it was measured on
    try:
        calibration = json.load(open(os.path.join(ROOT, "profiles", REFERENCE_TIMING_FILE)))
        ratio = float(calibration["port"][name]["port_over_reference"])
        out["reference_equivalent"] = dict(
            value=value / ratio, port_over_reference=ratio,
            source=f"profiles/{REFERENCE_TIMING_FILE}: the reference and this port timed on the build container's "
                   f"{calibration['host']['cores']} cores, workload {name}")
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return out


REFERENCE_TIMING_FILE = "r05_reference_cpu_timing.json"      # (round 4: port / reference 1.30 at C3; round 5: 1.70 -- the container's own timing of the reference moves between 0.0105 and 0.0142 structures/s)


import contextlib
import datetime
import threading


@contextlib.contextmanager
def stdout_to_stderr():
    """gloo prints a connection banner on the process's stdout (file descriptor 1, from C++): keep this program's stdout
    to its one JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


RENDEZVOUS_TIMEOUT = datetime.timedelta(seconds=120)     # process-group set-up and every collective: fail, do not hang


class CardSensor:
    """Socket power (W), power cap (W) and current shader-clock level (MHz) of the card THIS rank computes on, polled from
    sysfs (hwmon power1_average / power1_cap, pp_dpm_sclk of the device's PCI address) by a thread while the timed regions
    run: 10 polls per second, a few microseconds each.  Why it is on the bench line: the dominant kernel of the EGNN workloads
    runs at the card's power cap (profiles/r04_chain_power.json: 1 346 W of 1 400 W, 2.14 GHz), so on a node of eight cards a
    rank that is given less power or runs hotter is slower -- the per-rank record shows which, without a second run.
    Values are None where the sensor files are absent or unreadable (nothing is guessed)."""

    def __init__(self, device_index):
        import glob
        self.power, self.cap, self.sclk, self.bdf = None, None, None, None
        self.samples, self.clocks, self._stop, self._thread = [], [], False, None
        try:
            # the card's PCI address from torch's own device properties (no second handle on the HIP runtime)
            p = torch.cuda.get_device_properties(int(device_index))
            self.bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        except (AttributeError, RuntimeError, AssertionError):
            pass
        if self.bdf is None:
            return
        for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") +
                        glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")):
            if os.path.realpath(f.split("/hwmon/")[0]).lower().endswith(self.bdf):
                self.power = f
                self.cap = f.rsplit("/", 1)[0] + "/power1_cap"
                self.sclk = f.split("/hwmon/")[0] + "/pp_dpm_sclk"
                break

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return f.read()
        except (OSError, TypeError):
            return None

    def _poll(self):
        while not self._stop:
            v = self._read(self.power)
            if v and v.strip().isdigit() and int(v) > 0:
                self.samples.append(int(v) / 1e6)
            for line in (self._read(self.sclk) or "").splitlines():
                if line.rstrip().endswith("*"):
                    try:
                        self.clocks.append(float(line.split(":")[1].replace("Mhz", "").replace("*", "").strip()))
                    except (IndexError, ValueError):
                        pass
            time.sleep(0.1)

    def start(self):
        if self.power is not None and self._thread is None:
            self._stop = False
            self._thread = threading.Thread(target=self._poll, daemon=True)
            self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._stop = True
            self._thread.join()
            self._thread = None

    def summary(self):
        """(mean shader clock MHz, mean power W, cap W, number of samples); nan where unknown."""
        nan = float("nan")
        cap = self._read(self.cap)
        return (sum(self.clocks) / len(self.clocks) if self.clocks else nan,
                sum(self.samples) / len(self.samples) if self.samples else nan,
                int(cap) / 1e6 if cap and cap.strip().isdigit() else nan, float(len(self.samples)))


PER_RANK_FIELDS = ("rank", "ms_per_step", "trajectory_ms", "sclk_mhz_mean", "power_w_mean", "power_cap_w", "sensor_samples",
                   "f16_range_fallbacks")


def gather_per_rank(dist, world, record, device=None):
    """ONE small all-gather (len(PER_RANK_FIELDS) doubles per rank) after the timed regions -> the list of per-rank records
    of rank 0's line, with nan -> None."""
    mine = torch.tensor([float(v) for v in record], dtype=torch.float64, device=device)
    if dist is not None and world > 1:
        out = torch.empty(world * mine.numel(), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(out, mine)
    else:
        out = mine
    rows = out.cpu().reshape(-1, len(PER_RANK_FIELDS)).tolist()
    clean = []
    for row in rows:
        d = {k: (None if v != v else v) for k, v in zip(PER_RANK_FIELDS, row)}
        d["rank"], d["f16_range_fallbacks"] = int(d["rank"]), int(d["f16_range_fallbacks"] or 0)
        d["sensor_samples"] = int(d["sensor_samples"] or 0)
        for k in ("ms_per_step", "trajectory_ms", "sclk_mhz_mean", "power_w_mean", "power_cap_w"):
            d[k] = None if d[k] is None else round(d[k], 4)
        clean.append(d)
    return clean


def per_rank_summary(per_rank, batch, T, gather_ms):
    """slowest_rank and the N = 1 equivalents of an N-rank line: what ONE card of this job did, to be held against the
    N = 1 line of the same bench (BENCH_rNN.json): if the FASTEST rank's figure equals the N = 1 value, the loss at N is the
    spread between cards (power, temperature); if every rank is slower than N = 1, it is the node."""
    def job_ms(r):
        return (r["trajectory_ms"] if r["trajectory_ms"] is not None else T * r["ms_per_step"]) + gather_ms
    slowest = max(per_rank, key=job_ms)
    fastest = min(per_rank, key=job_ms)
    return slowest["rank"], dict(
        value_per_gpu_slowest_rank=round(batch / (job_ms(slowest) * 1e-3), 4),
        value_per_gpu_fastest_rank=round(batch / (job_ms(fastest) * 1e-3), 4),
        fastest_rank=fastest["rank"], spread=round(job_ms(slowest) / job_ms(fastest) - 1.0, 5),
        note="structures/s of ONE card over its own timed regions; compare with the N = 1 line's value")


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`--gpus N` with N > 1 and no WORLD_SIZE: start the ranks as a CHILD process.  Nothing in this process has initialised
    the GPU (importing torch and this package does not) and it never will -- it only relays: the child's stdout and stderr are
    inherited, its return code becomes this process's, and when it is not zero the rank that failed is named on stderr
    (every rank reports its own exception as a `bench_error` line before it dies; torch.distributed.run ends the other ranks as
    soon as one has failed, and every collective carries RENDEZVOUS_TIMEOUT, so a dead rank cannot hold the job)."""
    import subprocess
    import tempfile
    port = args.master_port if args.master_port else free_port()
    errors = tempfile.mkdtemp(prefix="bench_errors_")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               BENCH_ERROR_DIR=errors)
    t0 = time.perf_counter()
    code = subprocess.call(cmd, env=env)
    if code != 0:
        reports = []
        for name in sorted(os.listdir(errors)):
            with open(os.path.join(errors, name)) as f:
                reports.append(json.loads(f.read())["bench_error"])
        reports.sort(key=lambda r: r.get("at", 0.0))       # the rank that failed FIRST first: the others usually die of its death
        print(json.dumps({"bench_launcher": {"exit_code": code, "elapsed_s": round(time.perf_counter() - t0, 1),
                                             "rank_errors": reports or ["no rank left a report (killed by a signal?)"]}}),
              file=sys.stderr, flush=True)
    import shutil
    shutil.rmtree(errors, ignore_errors=True)
    raise SystemExit(code if code else 0)


def report_rank_error(rank, exc):
    """A rank's own account of why it is about to die: one line on stderr, and a file the launcher reads."""
    import traceback
    text = json.dumps({"bench_error": {"rank": rank, "at": time.time(), "error": repr(exc),
                                       "where": traceback.format_exception(type(exc), exc, exc.__traceback__)[-2].strip()}})
    print(text, file=sys.stderr, flush=True)
    folder = os.environ.get("BENCH_ERROR_DIR")
    if folder and os.path.isdir(folder):
        with open(os.path.join(folder, f"rank{rank}.json"), "w") as f:
            f.write(text)


def rehearse_launch(args, w, world, rank):
    """The multi-rank control flow of the job on host tensors: what the CPU tests of `--gpus N` exercise -- rendezvous with
    the time-out, the packed all-gather of synthetic compositions, the MAX reduction, the per-rank record's all-gather,
    rank 0's JSON line; `--rehearse-fail` makes one rank raise or die in the middle."""
    import torch.distributed as dist
    from diffusion_for_multi_scale_molecular_dynamics_amd.namespace import AXL
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if world > 1:
        with stdout_to_stderr():
            dist.init_process_group(backend="gloo", timeout=RENDEZVOUS_TIMEOUT)
            dist.barrier()
    if args.rehearse_fail:
        kind, who = args.rehearse_fail.split(":")
        if int(who) == rank:
            if kind == "raise":
                raise RuntimeError(f"rehearsal: rank {rank} was asked to fail")
            import signal
            os.kill(os.getpid(), signal.SIGKILL)
    batch, n = 4, w["n_atoms"]
    gen = torch.Generator().manual_seed(BASE_SEED + rank)
    comp = AXL(A=torch.randint(0, 2, (batch, n), generator=gen), X=torch.rand(batch, n, 3, generator=gen),
               L=torch.rand(batch, 6, generator=gen))
    rows = pack_compositions(comp)
    out = torch.empty((world * batch, rows.shape[1]), dtype=torch.uint8)
    t0 = time.perf_counter()
    if world > 1:
        dist.all_gather_into_tensor(out, rows)
    else:
        out.copy_(rows)
    local_ms = (time.perf_counter() - t0) * 1e3
    t = torch.tensor([local_ms], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = unpack_compositions(out, n, 3)
    ok = True
    for r in range(world):                        # every rank's block equals what that rank's seed generates
        g = torch.Generator().manual_seed(BASE_SEED + r)
        a = torch.randint(0, 2, (batch, n), generator=g)
        x = torch.rand(batch, n, 3, generator=g)
        ok = ok and torch.equal(gathered.A[r * batch:(r + 1) * batch], a) and \
            torch.equal(gathered.X[r * batch:(r + 1) * batch], x)
    # the per-rank record of the real job, with this rank's rehearsal figures (no card: the sensor fields are nan)
    nan = float("nan")
    per_rank = gather_per_rank(dist if world > 1 else None, world, (rank, 1.0 + rank, local_ms + 10.0 * rank, nan, nan, nan, 0, 0))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        slowest, n1 = per_rank_summary(per_rank, batch, w["noise"]["total_time_steps"], 0.0)
        print(json.dumps({"rehearsal": True, "n_gpus": world, "gather_ok": bool(ok), "gather_ms": float(t[0]),
                          "collectives": 1, "workload": args.workload, "per_rank": per_rank, "slowest_rank": slowest,
                          "n1_equivalent": n1}), flush=True)
    if not ok:
        raise SystemExit(1)


class Job:
    """What one process of the bench shares between the workloads it measures: its device, its process group and the timing
    protocol (barrier + synchronize | work | each rank stamps its clock | barrier; MAX over ranks)."""

    def __init__(self, args, device, dist, world, rank):
        self.args, self.device, self.dist, self.world, self.rank = args, device, dist, world, rank
        self.coll = (lambda t: t) if args.backend == "nccl" else (lambda t: t.cpu())   # gloo rehearsal: host copies

    def wait_for_gpu(self):
        """The GPU is awaited by polling an event before the blocking synchronize: a blocking synchronize alone wakes
        the host tens of microseconds late, which matters when K steps take ~1 ms."""
        done = torch.cuda.Event()
        done.record()
        while not done.query():
            pass
        torch.cuda.synchronize(self.device)

    def barrier(self):
        """torch.cuda.synchronize + dist.barrier + torch.cuda.synchronize."""
        self.wait_for_gpu()
        if self.dist is not None:
            self.dist.barrier()
            torch.cuda.synchronize(self.device)

    def timed(self, fn):
        """(MAX over ranks, this rank's own) elapsed seconds of fn: barrier + synchronize | fn | each rank stamps its clock
        when its own work is complete | barrier.  The time is the MAX over ranks (= when the last rank finished), so the
        closing barrier's own latency -- an RCCL all-reduce of ~50 us -- is not booked as sampling time."""
        self.barrier()
        t0 = time.perf_counter()
        fn()
        self.wait_for_gpu()
        local = time.perf_counter() - t0
        self.barrier()
        elapsed = local
        if self.dist is not None:
            t = self.coll(torch.tensor([local], dtype=torch.float64, device=self.device))
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t[0])
        return elapsed, local


def measure_workload(job, name, steps, warmup, whole_job_budget_s, egnn_precision, resampling_arg=None, batch_arg=None,
                     no_graph=False, forward_arg=None, other_mode=True, families=True, sensor=None):
    """One workload on this process's card: the K-step region, the whole-job region where it fits the budget, the job's one
    collective, the dominant kernels' launch durations.  Returns the pieces of a JSON line (every rank runs it; rank 0 alone
    times the stand-alone kernel launches)."""
    args, device, dist, world, rank = job.args, job.device, job.dist, job.world, job.rank
    w = WORKLOADS[name]
    batch = batch_arg or w["batch"]
    T = w["noise"]["total_time_steps"]
    mlp = w["net"] == "mlp"
    # defaults: the MLP workload runs its whole T-step trajectory (what the product does in one launch); the EGNN ones 3
    steps = steps if steps is not None else (T if mlp else 3)
    warmup = warmup if warmup is not None else (T if mlp else 1)
    forward = forward_arg or ("fused" if mlp else "pytorch")
    assert forward == "pytorch" or mlp, "the fused forward exists for the MLP score network only"
    # the EGNN iteration is capturable when its radius graph needs no host read: the fused edge chain in every layer
    use_graph = w["graph"] and not no_graph and forward == "pytorch" and (mlp or egnn_precision != "library")
    resampling = (resampling_arg if resampling_arg is not None else w.get("resampling", 0)) if "repaint" in w else 0
    gen, noise, sampling, net = build_generator(w, device, rank, batch, use_graph, resampling=resampling)
    gen.fused_score_network = forward == "fused"
    if not mlp:
        net.edge_chain_precision = None if egnn_precision == "library" else egnn_precision
    timed = job.timed

    with torch.no_grad():
        gen._prepare(device)
        gen._begin_call(device)                      # Philox seed = BASE_SEED + rank
        start = gen.initialize(batch, device)

        def new_loop():
            return FusedLoop(gen, start, T) if forward == "fused" else IterationLoop(gen, start, T, use_graph=use_graph)
        loop = new_loop()
        advance(loop, warmup, T)
        if sensor is not None:
            sensor.start()
        elapsed, local = timed(lambda: advance(loop, steps, T))
        ms_per_step, ms_per_step_local = elapsed * 1e3 / steps, local * 1e3 / steps
        # MLP workloads: the product runs the whole trajectory as ONE launch (4 ms); K iterations of it pay the launch's
        # fixed cost once per K.  So the job time is measured directly: one whole T-iteration trajectory, timed the same way.
        trajectory_ms = trajectory_ms_local = None
        if not mlp and 0 < T * ms_per_step * 1e-3 <= whole_job_budget_s:
            # EGNN workloads: the job itself, measured -- a fresh loop over the same start (capture of the iteration is a
            # one-off of the process and stays outside, like the warm-up), T replays, then the one host read of the status word
            loop = new_loop()

            def whole_job():
                advance(loop, T, T)
                gen.check_status()
            trajectory_ms, trajectory_ms_local = (1e3 * t for t in timed(whole_job))
        if mlp:
            # (one untimed trajectory first: the first T-iteration launch of a process also sizes and first-touches its
            # 328-MB noise workspace)
            loop = new_loop()
            advance(loop, T, T)
            loop = new_loop()
            trajectory_ms, trajectory_ms_local = (1e3 * t for t in timed(lambda: advance(loop, T, T)))
        if sensor is not None:
            sensor.stop()
        # the single collective of the job: ONE all-gather of the packed final compositions
        comp = loop.composition
        gather_ms = 0.0
        if dist is not None:
            rows = job.coll(pack_compositions(comp))
            out = torch.empty((world * rows.shape[0], rows.shape[1]), dtype=torch.uint8, device=rows.device)
            dist.all_gather_into_tensor(out, rows)      # untimed: RCCL sets up its rings / channels at the first collective
            gather_ms = timed(lambda: dist.all_gather_into_tensor(out, rows))[0] * 1e3
            gathered = unpack_compositions(out, w["n_atoms"], 3)
            mine = slice(rank * batch, (rank + 1) * batch)
            assert torch.equal(gathered.A[mine].to(device), comp.A) and torch.equal(gathered.X[mine].to(device), comp.X)
        gen.check_status()
        job_ms = (trajectory_ms if trajectory_ms is not None else T * ms_per_step) + gather_ms
        value = (batch * world) / (job_ms * 1e-3)
        other = None
        if other_mode and not mlp and egnn_precision in ("f32", "f16x3", "f16x3_32x32"):
            # the same job through the other arithmetic mode of the edge chain (2 iterations, same timing protocol)
            other_name = "f32" if egnn_precision != "f32" else "f16x3"
            net.edge_chain_precision = other_name
            loop_o = new_loop()
            advance(loop_o, 1, T)
            ms_o = timed(lambda: advance(loop_o, 2, T))[0] * 1e3 / 2
            net.edge_chain_precision = egnn_precision
            other = dict(egnn_edge_chain=other_name, ms_per_step=round(ms_o, 5),
                         value=round((batch * world) / ((T * ms_o + gather_ms) * 1e-3), 4), unit="structures/s")
            del loop_o
        generic_path = None
        if families and forward == "fused":
            # the same job through the generic instantiation of the persistent kernel (any MLP shape takes this path;
            # the dimension-specialised, folded instantiation above is selected when the network matches a template)
            from diffusion_for_multi_scale_molecular_dynamics_amd import _hip
            generic = {}
            for key, options in (("folded", _hip.MLP_SAMPLE_GENERIC_KERNEL),
                                 ("layer_by_layer", _hip.MLP_SAMPLE_GENERIC_KERNEL | _hip.MLP_SAMPLE_UNFOLDED),
                                 ("padded_family", _hip.MLP_SAMPLE_PADDED_FAMILY)):
                gen.fused_sampler_options = options
                loop_g = new_loop()
                advance(loop_g, T, T)
                loop_g = new_loop()
                generic[key] = timed(lambda: advance(loop_g, T, T))[0] * 1e3
            gen.fused_sampler_options = 0
            generic_path = dict(kernel="mlp_pc_sample_kernel<G,true,0> (generic instantiation: any MLP shape; folded input / output "
                                       "layers + hardware sin / cos, what shapes outside the register-resident family run)",
                                trajectory_ms=round(generic["folded"], 4),
                                value=round((batch * world) / ((generic["folded"] + gather_ms) * 1e-3), 2), unit="structures/s",
                                layer_by_layer=dict(trajectory_ms=round(generic["layer_by_layer"], 4),
                                                    value=round((batch * world) / ((generic["layer_by_layer"] + gather_ms) * 1e-3), 2)),
                                # the same network through the PADDED register-resident family (mlp_pc_sample_kernel<8,true,213>:
                                # run-time structure dimensions, fixed padded layer sizes): what every MLP configuration of the
                                # reference outside the exact template runs (hidden <= 64, N <= 8, <= 192 folded inputs)
                                padded_family=dict(trajectory_ms=round(generic["padded_family"], 4),
                                                   value=round((batch * world) / ((generic["padded_family"] + gather_ms) * 1e-3), 2)))

        roofline = forward_gemm = None
        if rank == 0:
            if forward == "fused":
                m = time_fused_kernel(gen, loop, batch, w, device)
            elif w["dominant"] == "pc_step_kernel":
                m = time_update_kernel(gen, batch, w, device)
            else:
                m = time_radius_graph(batch, w, device)
                n_e = int(round(m["edges_per_atom"] * batch * w["n_atoms"]))
                if egnn_precision == "library":
                    forward_gemm = time_edge_gemm(n_e, device)
                else:
                    forward_gemm = time_edge_chain(net, n_e, batch * w["n_atoms"], device)
            achieved = m["bytes"] / (m["ms"] * 1e-3) / 1e9
            traffic = None          # HBM bytes per launch measured with PMC counters in a separate rocprofv3 pass
            try:
                table = {}
                for fname in ("traffic_r01.json", "traffic_r02.json"):        # later rounds override
                    path = os.path.join(ROOT, "profiles", fname)
                    if os.path.exists(path):
                        table.update(json.load(open(path)))
                if m["kernel"].startswith("mdx_egnn_radius_graph"):
                    # the committed counter record of the two graph-build kernels at this workload's shape (same edge density:
                    # uniform random coordinates), scaled to the edge count of the build timed here
                    records = json.load(open(os.path.join(ROOT, "profiles", "traffic_graph_r05.json")))
                    entry = next(e for k, e in records.items() if k != "_comment" and e["batch"] == batch and
                                 e["number_of_atoms"] == w["n_atoms"])       # (C4 builds C3's graph: same shape and density)
                    traffic = int(entry["bytes_per_build"] * m["bytes"] / entry["algorithmic_bytes"])
                    m.setdefault("extra", {})["traffic_from"] = (
                        "profiles/traffic_graph_r05.json: rocprofv3 --pmc passes of the two kernels at this shape "
                        f"({entry['bytes_per_build']} B per build against {entry['algorithmic_bytes']} B algorithmic), scaled to the "
                        "edge count of the build timed here; not a measurement of this run")
                else:
                    entry = table[f"{name}/{forward}"]
                    if entry["kernel"] == m["kernel"].split()[0].split("<")[0] and batch == w["batch"]:
                        traffic = entry["bytes_per_launch"]
            except (OSError, KeyError, ValueError, StopIteration):
                pass
            roofline = dict(bound="hbm", achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, kernel=m["kernel"],
                            avg_launch_us=round(m["ms"] * 1e3, 3), algorithmic_bytes_per_launch=m["bytes"])
            if "compute" in m:
                roofline["compute"] = m["compute"]
            roofline.update(m.get("extra", {}))

    out = dict(
        name=name, w=w, T=T, mlp=mlp, batch=batch, steps=steps, warmup=warmup, forward=forward, use_graph=use_graph,
        resampling=resampling, ms_per_step=ms_per_step, ms_per_step_local=ms_per_step_local, trajectory_ms=trajectory_ms,
        trajectory_ms_local=trajectory_ms_local, gather_ms=gather_ms, job_ms=job_ms, value=value, other_mode=other,
        generic_path=generic_path, roofline=roofline, forward_gemm=forward_gemm,
        f16_range_fallbacks=int(gen.f16_range_fallbacks), peak_memory=int(torch.cuda.max_memory_allocated(device)))
    del loop, gen, net
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def as_line(m, world, egnn_precision, backend):
    """The JSON fields of one measured workload (the whole line for the primary one; an `also_measured` entry otherwise)."""
    w, T, mlp = m["w"], m["T"], m["mlp"]
    line = {
        "metric": "sampled structures/sec (%d-step predictor-corrector SDE sampling)" % T,
        "value": round(m["value"], 4), "unit": "structures/s", "n_gpus": world, "steps": m["steps"], "warmup": m["warmup"],
        "ms_per_step": round(m["ms_per_step"], 5), "job_ms": round(m["job_ms"], 4),
        "value_from": ("one whole %d-iteration trajectory timed end to end (trajectory_ms %.4f) + gather" % (T, m["trajectory_ms"]))
        if m["trajectory_ms"] is not None else "total_time_steps x ms_per_step + gather",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if mlp or egnn_precision not in ("f16x3", "f16x3_32x32") else
        "f32 (per-edge matrix products as split-f16 hi/lo x3 MFMA terms with f32 accumulation: 22-bit products; state, "
        "updates, reductions and every other layer in f32)", "data": "synthetic (random-init score network, uniform-random initial structures)",
        "config": {"workload": f"{m['name']}: {w['desc']}", "batch_per_gpu": m["batch"], "global_batch": m["batch"] * world,
                   "number_of_atoms": w["n_atoms"], "total_time_steps": T, "corrector_steps": w["M"],
                   "repaint_resampling_steps": m["resampling"],
                   "rng": "device Philox4x32-10", "hip_graph": bool(m["use_graph"]),
                   "score_network_forward": "fused HIP (one persistent kernel per launch of K iterations)"
                   if m["forward"] == "fused" else "PyTorch-ROCm module (plugin API)", "gather_ms": round(m["gather_ms"], 4),
                   "parallelism": f"independent batches x{world}, one all-gather at the end",
                   "collective_backend": backend,
                   # every switch of the library is an explicit argument; MDX_* variables are not read by the product
                   # and are listed only so that a stray one is visible
                   "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("MDX_")},
                   "peak_device_memory_bytes": m["peak_memory"],
                   "f16_range_fallbacks": m["f16_range_fallbacks"]},
        "roofline": m["roofline"],
    }
    if m["generic_path"] is not None:
        line["generic_path"] = m["generic_path"]
    if m["other_mode"] is not None:
        line["other_edge_chain_mode"] = m["other_mode"]
    if m["forward_gemm"] is not None:
        # EGNN workloads: the step is the per-edge MLP chain (matrix cores); the streaming kernels are < 1 % of it.  The
        # dominant kernel's roofline is `roofline`; the largest HBM-bound kernel (radius graph, N1) is `roofline_hbm`.
        line["roofline_hbm"], line["roofline"] = m["roofline"], m["forward_gemm"]
        line["config"]["egnn_edge_chain"] = egnn_precision
    return line


# The other BASELINE configurations, measured after the primary workload on the same card when the bench is run with its
# defaults at N = 1: (workload, steps, warmup, resampling, whole-job budget in seconds).  C2's job is one 4 ms launch and is
# timed whole, with its three kernel families; C4 is a 20-iteration region (its whole job is C3's, 31 s); C5 a 3-iteration
# region without and with the resampling pass (its whole jobs take 6 and 11 minutes: profiles/r04_bench_c5_whole_job.json).
ALSO_MEASURED = (("C2", None, None, None, 0.0), ("C4", 20, 2, None, 0.0), ("C5", 3, 1, 0, 0.0), ("C5", 3, 1, 1, 0.0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="C3", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the iteration into a hipGraph")
    ap.add_argument("--forward", choices=["fused", "pytorch"], default=None,
                    help="score-network forward: 'pytorch' (plugin API, any network) or 'fused' (MLP only: network "
                         "forward + update in one persistent HIP kernel); default: fused for MLP workloads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--whole-job-budget-s", type=float, default=150.0,
                    help="EGNN workloads: after the K-step region, time ONE WHOLE T-iteration trajectory (graph replays + the "
                         "status read) and report `value` from it, when T x ms_per_step is below this many seconds "
                         "(C3 / C4: ~31 s; C5's 2000 x 0.17 .. 0.34 s is not, its value stays T x ms_per_step); 0 = never")
    ap.add_argument("--resampling", type=int, default=None,
                    help="C5 only: RePaint resampling passes per time index (BASELINE configs[4] 'with resampling'; "
                         "default 1; 0 = the reference's loop, which has no resampling)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo (with ranks sharing a GPU) exists to rehearse the multi-rank control "
                         "flow on a one-GPU box")
    ap.add_argument("--egnn-precision", choices=["f32", "f16x3", "f16x3_32x32", "library"], default="f16x3",
                    help="EGNN workloads: arithmetic of the fused per-edge MFMA kernel -- 'f16x3' (the product's default) "
                         "split-f16 three-product form with binary32 accumulation, binary32-level accuracy (error against fp64 "
                         "equal to the f32 paths': tests/test_egnn_chain_gpu.py); 'f32' exact binary32 MFMA; 'library' = "
                         "per-layer PyTorch modules (no hand-written chain).  The other MFMA mode is timed too and reported beside it.")
    ap.add_argument("--master-port", type=int, default=None,
                    help="rendezvous port when bench.py starts its own ranks (default: a free port of 127.0.0.1)")
    ap.add_argument("--also-measured", choices=["auto", "yes", "no"], default="auto",
                    help="after the primary workload, measure the other BASELINE configurations (ALSO_MEASURED) on the same card "
                         "and report them under `also_measured`; auto = at N = 1 with the default workload")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="no GPU work: the ranks only run the job's control flow on host tensors (gloo) -- rendezvous, the "
                         "packed all-gather of synthetic compositions, the MAX reduction, the per-rank record, rank 0's JSON "
                         "line; used by the CPU tests of the launcher")
    ap.add_argument("--rehearse-fail", default=None, metavar="raise:R|kill:R",
                    help="rehearsal only: rank R raises an exception (raise) or is killed by a signal (kill) after the rendezvous")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)

    w = WORKLOADS[args.workload]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    try:
        run_rank(args, w, world, rank)
    except BaseException as exc:                     # noqa: BLE001  (report, then die with the same exception)
        if not isinstance(exc, SystemExit) or exc.code not in (0, None):
            report_rank_error(rank, exc)
        raise


def run_rank(args, w, world, rank):
    if args.rehearse_launch:
        return rehearse_launch(args, w, world, rank)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py measures the GPU hot path: no GPU is visible (there is no CPU fallback)")
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or "RANK" in os.environ:
        # (under torchrun the process group is initialised at ANY world size: `torchrun --nproc-per-node 1 bench.py` runs the
        # job's RCCL calls -- group set-up, the packed all-gather, the MAX all-reduce, barriers, teardown -- on one GPU)
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device, timeout=RENDEZVOUS_TIMEOUT)      # RCCL over xGMI
        else:
            with stdout_to_stderr():
                dist.init_process_group(backend="gloo", timeout=RENDEZVOUS_TIMEOUT)
                dist.barrier()
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    job = Job(args, device, dist, world, rank)
    sensor = CardSensor(device.index)

    m = measure_workload(job, args.workload, args.steps, args.warmup, args.whole_job_budget_s, args.egnn_precision,
                         resampling_arg=args.resampling, batch_arg=args.batch, no_graph=args.no_graph,
                         forward_arg=args.forward, sensor=sensor)
    # the per-rank record: ONE small all-gather after the timed regions
    nan = float("nan")
    record = (rank, m["ms_per_step_local"], m["trajectory_ms_local"] if m["trajectory_ms_local"] is not None else nan,
              *sensor.summary(), m["f16_range_fallbacks"])
    per_rank = gather_per_rank(dist, world, record, device=device if args.backend == "nccl" else None)

    also = None
    wanted = args.also_measured == "yes" or (args.also_measured == "auto" and world == 1 and args.workload == "C3"
                                             and args.batch is None and args.forward is None and not args.no_graph)
    if wanted:
        also = []
        for name, steps, warmup, resampling, budget in ALSO_MEASURED:
            t0 = time.perf_counter()
            try:
                other = measure_workload(job, name, steps, warmup, budget, args.egnn_precision, resampling_arg=resampling,
                                         other_mode=False, families=True)
                entry = as_line(other, world, args.egnn_precision, args.backend if dist is not None else None)
            except Exception as exc:          # noqa: BLE001  (a side measurement must not take the primary line with it: say so)
                if world > 1:
                    raise                     # (ranks must stay in step: the launcher reports the failing rank)
                entry = dict(config=dict(workload=name), error=f"{type(exc).__name__}: {exc}"[:500])
                torch.cuda.synchronize(device)
            if rank == 0:
                entry["measured_in_s"] = round(time.perf_counter() - t0, 1)
                also.append(entry)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    result = as_line(m, world, args.egnn_precision, args.backend if dist is not None else None)
    slowest, n1 = per_rank_summary(per_rank, m["batch"], m["T"], m["gather_ms"])
    result["per_rank"], result["slowest_rank"], result["n1_equivalent"] = per_rank, slowest, n1
    if also is not None:
        result["also_measured"] = also
    if world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(w, args.workload, resampling=m["resampling"])
        except Exception as exc:              # noqa: BLE001  (the measured line is printed whatever the CPU leg does: say what it did)
            result["cpu_baseline"] = dict(value=None, unit="structures/s", kind="port", error=f"{type(exc).__name__}: {exc}"[:500])
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
