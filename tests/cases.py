"""The golden trajectory cases of tests/golden/make_golden.py, restated as (noise, sampling, network) builders."""
from types import SimpleNamespace

import nets

LIN = dict(sigma_min=1e-4, sigma_max=0.2, schedule_type="linear", corrector_step_epsilon=2.5e-8)


def noise_ns(T, **kw):
    d = dict(total_time_steps=T, schedule_type="exponential", time_delta=1e-5, sigma_min=0.005, sigma_max=0.5,
             corrector_step_epsilon=2e-5, corrector_r=0.17)
    d.update(kw)
    return d


def sampling_ns(N, num_atom_types, M=1, greedy=True, one=True, in_corr=False, eps=1e-8, fixed=True, cell=None, d=3):
    s = dict(algorithm="predictor_corrector", number_of_atoms=N, num_atom_types=num_atom_types, number_of_samples=1,
             spatial_dimension=d, number_of_corrector_steps=M, atom_type_greedy_sampling=greedy,
             one_atom_type_transition_per_step=one, atom_type_transition_in_corrector=in_corr, small_epsilon=eps,
             use_fixed_lattice_parameters=fixed)
    if fixed:
        s["cell_dimensions"] = cell or [5.43] * d
    return s


# name -> (noise kwargs, sampling kwargs, network factory(edge_builder) or None for the echo network)
TRAJECTORIES = {
    "traj_fake_c2": (noise_ns(12), sampling_ns(8, 1), None),
    "traj_fake_c3_m2": (noise_ns(10, schedule_type="linear"), sampling_ns(8, 2, M=2), None),
    "traj_fake_c5_nogreedy": (noise_ns(10), sampling_ns(8, 4, greedy=False, one=False), None),
    "traj_fake_c5_test": (noise_ns(10, time_delta=0.1, sigma_min=0.15, corrector_step_epsilon=0.25),
                          sampling_ns(8, 4, M=2, eps=1e-6, in_corr=True), None),
    "traj_fake_free_lattice": (noise_ns(8), sampling_ns(8, 1, fixed=False), None),
    "traj_mlp_c1": (noise_ns(20, sigma_min=1e-4, sigma_max=0.25), sampling_ns(8, 1), lambda eb: nets.mlp_net(8, 1)),
    "traj_mlp_c3": (noise_ns(16, **LIN), sampling_ns(8, 2, M=2, cell=[5.5421] * 3), lambda eb: nets.mlp_net(8, 2)),
    "traj_egnn_fc": (noise_ns(6, **LIN), sampling_ns(8, 1, one=False, greedy=False),
                     lambda eb: nets.egnn_net(1, "fully_connected", None, edge_builder=eb)),
    "traj_egnn_rc": (noise_ns(4, **LIN), sampling_ns(64, 1, M=2, one=False, greedy=False, cell=[10.86] * 3),
                     lambda eb: nets.egnn_net(1, "radial_cutoff", 7.5, edge_builder=eb)),
}
REPAINT = {
    "traj_repaint_fake": (noise_ns(10), sampling_ns(8, 2), None),
    "traj_repaint_mlp": (noise_ns(12, **LIN), sampling_ns(8, 1, M=2), lambda eb: nets.mlp_net(8, 1)),
}


def as_objects(noise_kw, sampling_kw):
    return SimpleNamespace(**noise_kw), SimpleNamespace(**sampling_kw)

ADAPTIVE = {
    "traj_adaptive_fake": (noise_ns(8, corrector_r=0.5), dict(sampling_ns(8, 2, M=2), algorithm="adaptive_corrector"), None),
    "traj_adaptive_mlp": (noise_ns(10, sigma_min=1e-3, sigma_max=0.2, schedule_type="linear"),
                          dict(sampling_ns(8, 1), algorithm="adaptive_corrector"), lambda eb: nets.mlp_net(8, 1)),
}


# BASELINE configs[0] at its exact settings (tests/golden/traj_c1_exact.npz): T = 100, batch 16, MLP template
C1_EXACT = (noise_ns(100, sigma_min=1e-4, sigma_max=0.25), sampling_ns(8, 1), lambda eb: nets.mlp_net(8, 1))


# BASELINE configs[2]'s network shape and sampler settings (tests/golden/traj_egnn_c3_{top,bottom}.npz: two indices of the
# T = 1000 schedule at the top and at the bottom, B = 4, the production EGNN with formula weights)
C3_SHAPE = (noise_ns(1000, **LIN), sampling_ns(64, 1, M=2, one=False, greedy=False, cell=[10.86] * 3),
            lambda eb: nets.egnn_c3_net(1, edge_builder=eb))

# BASELINE configs[3]'s settings at the same network shape (SiGe: two atom types, greedy sampling + one transition per step,
# cell 11.084): tests/golden/traj_egnn_c4_{top,mid}.npz
C4_SHAPE = (noise_ns(1000, **LIN), sampling_ns(64, 2, M=2, one=True, greedy=True, cell=[11.084] * 3),
            lambda eb: nets.egnn_c3_net(2, edge_builder=eb))


# The same two settings with the "live" network (formula weights at 2 x the default range, where a 256 x 256 hidden matrix
# matters to the output: tests/golden/make_golden.py::golden_live): traj_egnn_c3_live.npz (500 -> 498),
# traj_egnn_c4_live_bottom.npz (2 -> 0: the last predictor step with C = 3)
C3_LIVE_SHAPE = (C3_SHAPE[0], C3_SHAPE[1], lambda eb: nets.egnn_c3_net(1, edge_builder=eb, scale=nets.LIVE_SCALE))
C4_LIVE_SHAPE = (C4_SHAPE[0], C4_SHAPE[1], lambda eb: nets.egnn_c3_net(2, edge_builder=eb, scale=nets.LIVE_SCALE))


def shape_of(name):
    """(noise, sampling, network factory) of a production-width trajectory fixture, by its name."""
    if "_c4_live" in name:
        return C4_LIVE_SHAPE
    if "_c3_live" in name:
        return C3_LIVE_SHAPE
    return C4_SHAPE if "_c4_" in name else C3_SHAPE


def diamond_sites(n_cells):
    """The 8 n^3 sites of the diamond structure in an n x n x n supercell, relative coordinates, cell-major order (the
    constraint of BASELINE configs[4]: the first 108 of the 216 sites of Si 3x3x3 are pinned)."""
    import torch
    base = torch.tensor([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                         [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]])
    cells = torch.cartesian_prod(*[torch.arange(n_cells)] * 3).float()
    return ((cells[:, None, :] + base[None]) / n_cells).reshape(-1, 3)


# BASELINE configs[4] at the production network (tests/golden/net_egnn_c5.npz, traj_egnn_c5_{top,bottom}.npz): Si 3x3x3,
# N = 216, cell 16.29, T = 2000 linear, M = 2, ConstrainedLangevinGenerator with K = 108 pinned diamond sites, B = 2
C5_SHAPE = (noise_ns(2000, **LIN), sampling_ns(216, 1, M=2, one=False, greedy=False, cell=[16.29] * 3),
            lambda eb: nets.egnn_c3_net(1, edge_builder=eb))


def periodic_well_mlp_state(sites, amplitude=4.0, offset=24.0, factor=1.0, reference_shapes=None):
    """Weights for the MLP template (N = 8, one atom type, hidden 64 x 3: tests/nets.py::mlp_net(8, 1) and the reference's
    MLPScoreNetwork of tests/golden/make_golden.py::_mlp(8, 1)) that make it a KNOWN function with a strong, structured score:

        out.X[i] = - factor * amplitude * sin(2 pi (x_i - site_i)),     logits and lattice output zero

    -- a periodic well around every site.  The network's own input features are cos / sin of 2 pi x
    (src/.../models/score_networks/mlp_score_network.py:286-293), of which the target is a linear function; the two SiLUs between
    the three layers are passed in their linear regime: the 24 live neurons carry e_i + offset >= 20, where SiLU(z) = z (1 - e^-z)
    differs from z by < 5e-8.  Every other weight and bias is zero.  Returns {parameter name: float32 array}; `reference_shapes`
    (name -> shape, e.g. of a state_dict) is checked when given."""
    import numpy as np
    s = np.asarray(sites, np.float64).reshape(-1)                      # 24 = (atom, axis) in the network's flattening order
    n = s.size
    z = lambda *shape: np.zeros(shape, np.float32)                     # noqa: E731
    state = {"relative_coordinates_embedding_layer.weight": z(32, 2 * n), "relative_coordinates_embedding_layer.bias": z(32),
             "noise_embedding_layer.weight": z(16, 1), "noise_embedding_layer.bias": z(16),
             "time_embedding_layer.weight": z(16, 1), "time_embedding_layer.bias": z(16),
             "atom_type_embedding_layer.weight": z(1, 2), "atom_type_embedding_layer.bias": z(1),
             "lattice_parameters_embedding_layer.weight": z(1, 6), "lattice_parameters_embedding_layer.bias": z(1),
             "condition_embedding_layer.weight": z(64, n), "condition_embedding_layer.bias": z(64),
             "output_A_layer.weight": z(2 * (n // 3), 64), "output_A_layer.bias": z(2 * (n // 3)),
             "output_X_layer.weight": z(n, 64), "output_X_layer.bias": z(n),
             "output_L_layer.weight": z(6, 64), "output_L_layer.bias": z(6)}
    width_in = 32 + 16 + 16 + (n // 3) + 1
    for k in range(3):
        state[f"mlp_layers.{k}.weight"] = z(64, width_in if k == 0 else 64)
        state[f"mlp_layers.{k}.bias"] = z(64)
        state[f"conditional_layers.{k}.weight"] = z(64, 64)
        state[f"conditional_layers.{k}.bias"] = z(64)
    a = factor * amplitude
    for i in range(n):                                                 # input = [cos(2 pi x) (n) | sin(2 pi x) (n)]
        state["relative_coordinates_embedding_layer.weight"][i, i] = a * np.sin(2 * np.pi * s[i])
        state["relative_coordinates_embedding_layer.weight"][i, n + i] = -a * np.cos(2 * np.pi * s[i])
        for k in range(3):
            state[f"mlp_layers.{k}.weight"][i, i] = 1.0
        state["mlp_layers.0.bias"][i] = offset
        state["mlp_layers.2.bias"][i] = -offset
        state["output_X_layer.weight"][i, i] = 1.0
    if reference_shapes is not None:
        assert {k: tuple(v.shape) for k, v in state.items()} == {k: tuple(v) for k, v in reference_shapes.items()}
    return state
