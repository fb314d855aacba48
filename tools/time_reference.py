"""Time the REFERENCE itself (container-only: it imports /root/reference; nothing here travels to the GPU box).

    PYTHONPATH=/root/reference/src python tools/time_reference.py [--repeats 3]

BASELINE.md section 3 step 1: `create_batch_of_samples` wall time on this container's cores, warm-up excluded, >= 3
repeats, for C1 (configs[0] exactly), C2 on CPU (B = 1024, T = 1000) and C3 on CPU (Si 2x2x2, EGNN 4 x 256, rc 7.5,
M = 2; a few iterations at B = 16 and B = 64, extrapolated to T = 1000 -- one iteration is seconds).  The third-party
packages the reference imports but this image lacks are stubbed exactly as for the golden vectors
(tests/golden/make_golden.py: pykeops.LazyTensor as a dense torch evaluation).  Output: one JSON document,
committed as profiles/r02_reference_cpu_timing.json and quoted in DESIGN.md next to bench.py's `cpu_baseline`."""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as G  # noqa: E402  (installs the stubs, imports the reference)


def egnn_experiment():
    # experiments/.../Si_2x2x2/config_diffusion_egnn.yaml:44-60
    torch.manual_seed(1234)
    p = G.EGNNScoreNetworkParameters(
        num_atom_types=1, n_layers=4, coordinate_hidden_dimensions_size=256, coordinate_n_hidden_dimensions=4,
        message_hidden_dimensions_size=256, message_n_hidden_dimensions=4, node_hidden_dimensions_size=256,
        node_n_hidden_dimensions=4, coords_agg="mean", message_agg="mean", attention=False, normalize=False,
        residual=True, tanh=False, edges="radial_cutoff", radial_cutoff=7.5)
    return G.EGNNScoreNetwork(p).eval()


def timed_runs(gen, sampling_parameters, batch, repeats):
    sampling_parameters.number_of_samples = batch
    sampling_parameters.sample_batchsize = batch
    walls = []
    with torch.no_grad():
        for k in range(repeats + 1):                       # first run = warm-up, excluded
            torch.manual_seed(20250815)
            t0 = time.perf_counter()
            out = G.create_batch_of_samples(gen, sampling_parameters, torch.device("cpu"))
            walls.append(time.perf_counter() - t0)
            assert out["original_axl"].X.shape[0] == batch
    return walls[1:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--port", action="store_true", help="also time bench.py's cpu_baseline (the oracle port) on this host")
    args = ap.parse_args()
    result = {"host": {"cores": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__},
              "protocol": "create_batch_of_samples wall time, first run excluded, %d repeats" % args.repeats, "configs": {}}
    exp = dict(sigma_min=1e-4, sigma_max=0.25)
    lin = dict(sigma_min=1e-4, sigma_max=0.2, schedule_type="linear", corrector_step_epsilon=2.5e-8)
    cases = [("C1", dict(T=100, N=8, nat=1, M=1, greedy=True, one=True, cell=[5.43] * 3, noise=exp, net="mlp"), 16, 100),
             ("C2_cpu", dict(T=1000, N=8, nat=1, M=1, greedy=True, one=True, cell=[5.43] * 3, noise=exp, net="mlp"), 1024, 1000),
             ("C3_cpu_B16", dict(T=3, N=64, nat=1, M=2, greedy=False, one=False, cell=[10.86] * 3, noise=lin, net="egnn"), 16, 1000),
             ("C3_cpu_B64", dict(T=2, N=64, nat=1, M=2, greedy=False, one=False, cell=[10.86] * 3, noise=lin, net="egnn"), 64, 1000)]
    for name, c, batch, full_T in cases:
        net = G._mlp(c["N"], c["nat"]) if c["net"] == "mlp" else egnn_experiment()
        gen, _, sp = G.make_generator(c["T"], c["N"], c["nat"], M=c["M"], greedy=c["greedy"], one=c["one"], cell=c["cell"],
                                      noise_kw=c["noise"], net=net)
        walls = timed_runs(gen, sp, batch, args.repeats)
        per_iteration = [w / c["T"] for w in walls]
        job = [p * full_T for p in per_iteration]
        result["configs"][name] = {
            "batch": batch, "timed_iterations": c["T"], "job_iterations": full_T, "wall_s": [round(w, 4) for w in walls],
            "s_per_iteration_median": round(statistics.median(per_iteration), 5),
            "structures_per_s": round(batch / statistics.median(job), 5),
            "structures_per_s_min_max": [round(batch / max(job), 5), round(batch / min(job), 5)],
            "extrapolated": c["T"] != full_T}
        print(name, result["configs"][name], flush=True)
    if args.port:
        # the build's own CPU restatement -- the `cpu_baseline` leg of bench.py, same bounded sample of C3 -- on
        # THIS host, so that the `cpu_baseline` bench.py reports from the GPU box's cores can be read against the reference
        # figure above (same machine, same day): calibration = port / reference.
        sys.path.insert(0, ROOT)
        import bench
        for name in ("C3", "C2"):
            port = bench.cpu_baseline(bench.WORKLOADS[name], name)
            ref = result["configs"]["C3_cpu_B16" if name == "C3" else "C2_cpu"]["structures_per_s"]
            result.setdefault("port", {})[name] = dict(port, port_over_reference=round(port["value"] / ref, 3))
            print(name, result["port"][name], flush=True)
    print(json.dumps(result))


if __name__ == "__main__":
    main()
