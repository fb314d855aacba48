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
