"""Tensor-level wrappers of the C ABI (include/mdx_hip.h).

Each function validates shapes/dtypes, passes raw device pointers + torch's current HIP stream to the shared
library, and returns torch tensors that PyTorch owns.  Nothing here computes on the CPU.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _hip
from ._hip import Mlp, PcFlags, Rng, Schedule, check, lib, ptr, stream_handle

F32, I64, I32 = torch.float32, torch.int64, torch.int32


# ----------------------------------------------------------------------------------------------------------------
# S1
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class DeviceSchedule:
    """Device-resident schedule tables + the mdx_schedule_t view handed to the kernels."""

    total_time_steps: int
    num_classes: int
    sigma_min: float
    time: torch.Tensor
    sigma: torch.Tensor
    sigma_squared: torch.Tensor
    g: torch.Tensor
    g_squared: torch.Tensor
    epsilon: torch.Tensor
    sqrt_2_epsilon: torch.Tensor
    beta: torch.Tensor
    alpha_bar: torch.Tensor
    q_matrix: torch.Tensor
    q_bar_matrix: torch.Tensor
    q_bar_tm1_matrix: torch.Tensor

    def __post_init__(self):
        self.c_struct = Schedule(self.total_time_steps, self.num_classes, float(self.sigma_min),
                                 self.time.data_ptr(), self.sigma.data_ptr(), self.g.data_ptr(),
                                 self.g_squared.data_ptr(), self.epsilon.data_ptr(), self.q_matrix.data_ptr(),
                                 self.q_bar_matrix.data_ptr(), self.q_bar_tm1_matrix.data_ptr())

    @property
    def device(self):
        return self.time.device


def noise_schedule_build(total_time_steps: int, schedule_type: str, time_delta: float, sigma_min: float,
                         sigma_max: float, corrector_step_epsilon: float, num_classes: int,
                         device: torch.device) -> DeviceSchedule:
    """S1: build the variance-exploding schedule tables on the device (mdx_noise_schedule_build)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise _hip.MdxError(f"schedule tables are built on the GPU; got device {device}")
    T, Cn = int(total_time_steps), int(num_classes)
    st = {"exponential": 0, "linear": 1}[schedule_type]
    with torch.cuda.device(device):
        vec = [torch.empty(T, dtype=F32, device=device) for _ in range(9)]
        mats = [torch.empty(T, Cn, Cn, dtype=F32, device=device) for _ in range(3)]
        rc = lib().mdx_noise_schedule_build(T, st, float(time_delta), float(sigma_min), float(sigma_max),
                                            float(corrector_step_epsilon), Cn,
                                            *[C.c_void_p(t.data_ptr()) for t in vec + mats], stream_handle())
    check(rc, "mdx_noise_schedule_build")
    time, sigma, sigma2, g, g2, eps, s2e, beta, ab = vec
    q, qb, qbm = mats
    return DeviceSchedule(T, Cn, float(sigma_min), time, sigma, sigma2, g, g2, eps, s2e, beta, ab, q, qb, qbm)


# ----------------------------------------------------------------------------------------------------------------
# step index / network inputs
# ----------------------------------------------------------------------------------------------------------------
def index_set(d_index: torch.Tensor, value: int):
    check(lib().mdx_index_set(ptr(d_index, I32, "d_index"), int(value), stream_handle()), "mdx_index_set")


def index_add(d_index: torch.Tensor, delta: int):
    check(lib().mdx_index_add(ptr(d_index, I32, "d_index"), int(delta), stream_handle()), "mdx_index_add")


def fill_time_sigma(sched: DeviceSchedule, mode: int, index_i: int, d_index: Optional[torch.Tensor],
                    time_out: torch.Tensor, sigma_out: torch.Tensor):
    batch = time_out.numel()
    assert sigma_out.numel() == batch
    rc = lib().mdx_fill_time_sigma(C.byref(sched.c_struct), mode, int(index_i), ptr(d_index, I32, "d_index"),
                                   ptr(time_out, F32, "time_out"), ptr(sigma_out, F32, "sigma_out"), batch,
                                   stream_handle())
    check(rc, "mdx_fill_time_sigma")


# ----------------------------------------------------------------------------------------------------------------
# P1 / P2 / P3 with explicit operands (mirror the reference's private update methods)
# ----------------------------------------------------------------------------------------------------------------
def relative_coordinates_update(x, s, z, score_weight=None, gaussian_noise_weight=None, sigma=None, out=None, weights=None):
    """P1.  The three scalars as host numbers, or -- `weights`, float32 [3] on the device: {score weight, noise weight,
    sigma} -- read by the kernel (no host synchronisation: AdaptiveCorrectorGenerator)."""
    assert x.shape == s.shape == z.shape
    out = torch.empty_like(x) if out is None else out
    if weights is not None:
        assert weights.numel() == 3
        rc = lib().mdx_relative_coordinates_update_dev(ptr(x, F32, "x"), ptr(s, F32, "sigma_normalized_scores"), ptr(z, F32, "z"),
                                                       ptr(weights, F32, "weights"), x.numel(), ptr(out, F32, "out"),
                                                       stream_handle())
        check(rc, "mdx_relative_coordinates_update_dev")
        return out
    rc = lib().mdx_relative_coordinates_update(ptr(x, F32, "x"), ptr(s, F32, "sigma_normalized_scores"),
                                               ptr(z, F32, "z"), float(score_weight), float(gaussian_noise_weight),
                                               float(sigma), x.numel(), ptr(out, F32, "out"), stream_handle())
    check(rc, "mdx_relative_coordinates_update")
    return out


def lattice_parameters_update(l, s, z, score_weight=None, gaussian_noise_weight=None, sigma_n=None, out=None, weights=None):
    """P3; `weights` as in relative_coordinates_update."""
    assert l.shape == s.shape == z.shape
    out = torch.empty_like(l) if out is None else out
    if weights is not None:
        assert weights.numel() == 3
        rc = lib().mdx_lattice_parameters_update_dev(ptr(l, F32, "l"), ptr(s, F32, "sigma_normalized_scores"), ptr(z, F32, "z"),
                                                     ptr(weights, F32, "weights"), l.numel(), ptr(out, F32, "out"),
                                                     stream_handle())
        check(rc, "mdx_lattice_parameters_update_dev")
        return out
    rc = lib().mdx_lattice_parameters_update(ptr(l, F32, "l"), ptr(s, F32, "sigma_normalized_scores"),
                                             ptr(z, F32, "z"), float(score_weight), float(gaussian_noise_weight),
                                             float(sigma_n), l.numel(), ptr(out, F32, "out"), stream_handle())
    check(rc, "mdx_lattice_parameters_update")
    return out


def atom_types_update(logits, atom_types, q, q_bar, q_bar_tm1, gumbel, u, small_epsilon: float, greedy: bool,
                      one_transition: bool, return_probabilities: bool = False):
    B, N, Cn = logits.shape
    assert atom_types.shape == (B, N) and gumbel.shape == (B, N, Cn)
    assert q.shape == q_bar.shape == q_bar_tm1.shape == (Cn, Cn)
    out = torch.empty_like(atom_types)
    p_out = torch.empty_like(logits) if return_probabilities else None
    rc = lib().mdx_atom_types_update(ptr(logits, F32, "logits"), ptr(atom_types, I64, "atom_types"),
                                     ptr(q, F32, "q"), ptr(q_bar, F32, "q_bar"), ptr(q_bar_tm1, F32, "q_bar_tm1"),
                                     ptr(gumbel, F32, "gumbel"), ptr(u, F32, "u"), B, N, Cn, float(small_epsilon),
                                     int(bool(greedy)), int(bool(one_transition)), ptr(out, I64, "out"),
                                     ptr(p_out, F32, "p_out"), stream_handle())
    check(rc, "mdx_atom_types_update")
    return (out, p_out) if return_probabilities else out


# ----------------------------------------------------------------------------------------------------------------
# fused per-step update
# ----------------------------------------------------------------------------------------------------------------
def pc_step_update(sched: DeviceSchedule, mode: int, index_i: int, d_index: Optional[torch.Tensor], flags: PcFlags,
                   atom_types, x, l, logits, score_x, score_l, z_coordinates, gumbel, u, z_lattice, rng: Rng,
                   atom_types_out, x_out, l_out, status: Optional[torch.Tensor]):
    B, N, d = x.shape
    rc = lib().mdx_pc_step_update(
        C.byref(sched.c_struct), int(mode), int(index_i), ptr(d_index, I32, "d_index"), C.byref(flags),
        ptr(atom_types, I64, "atom_types"), ptr(x, F32, "x"), ptr(l, F32, "l"), ptr(logits, F32, "logits"),
        ptr(score_x, F32, "score_x"), ptr(score_l, F32, "score_l"), ptr(z_coordinates, F32, "z_coordinates"),
        ptr(gumbel, F32, "gumbel"), ptr(u, F32, "u"), ptr(z_lattice, F32, "z_lattice"), rng, B, N, d,
        ptr(atom_types_out, I64, "atom_types_out"), ptr(x_out, F32, "x_out"), ptr(l_out, F32, "l_out"),
        ptr(status, I32, "status"), stream_handle())
    check(rc, "mdx_pc_step_update")


# ----------------------------------------------------------------------------------------------------------------
# F1 / F2 / R1
# ----------------------------------------------------------------------------------------------------------------
def noise_relative_coordinates(x0, z, sigma, out=None):
    """wrap(x0 + sigma z); sigma: one number, or a tensor of x0's shape (the reference's per-element sigmas)."""
    assert x0.shape == z.shape
    out = torch.empty_like(x0) if out is None else out
    if isinstance(sigma, torch.Tensor):
        assert sigma.shape == x0.shape, "sigmas array is expected to be of the same shape as the real_relative_coordinates array"
        rc = lib().mdx_noise_relative_coordinates_sigmas(ptr(x0, F32, "x0"), ptr(z, F32, "z"), ptr(sigma, F32, "sigmas"),
                                                         x0.numel(), ptr(out, F32, "out"), stream_handle())
        check(rc, "mdx_noise_relative_coordinates_sigmas")
        return out
    rc = lib().mdx_noise_relative_coordinates(ptr(x0, F32, "x0"), ptr(z, F32, "z"), float(sigma), x0.numel(),
                                              ptr(out, F32, "out"), stream_handle())
    check(rc, "mdx_noise_relative_coordinates")
    return out


def noise_lattice_parameters(l0, z, sigmas_n):
    """sigmas_n * z + l0, element by element (mdx_noise_lattice_parameters)."""
    assert l0.shape == z.shape == sigmas_n.shape
    out = torch.empty_like(l0)
    rc = lib().mdx_noise_lattice_parameters(ptr(l0, F32, "l0"), ptr(z, F32, "z"), ptr(sigmas_n, F32, "sigmas_n"), l0.numel(),
                                            ptr(out, F32, "out"), stream_handle())
    check(rc, "mdx_noise_lattice_parameters")
    return out


def noise_atom_types(a0, q_bar, u):
    """q_bar: one [C, C] matrix for the call, or one per atom [*a0.shape, C, C]."""
    Cn = u.shape[-1]
    assert u.shape[:-1] == a0.shape
    out = torch.empty_like(a0)
    if q_bar.dim() > 2:
        assert q_bar.shape == tuple(a0.shape) + (Cn, Cn), "q_bar array first dimensions should match real_atom_types array"
        rc = lib().mdx_noise_atom_types_per_atom(ptr(a0, I64, "a0"), ptr(q_bar, F32, "q_bar"), ptr(u, F32, "u"), a0.numel(),
                                                 Cn, ptr(out, I64, "out"), stream_handle())
        check(rc, "mdx_noise_atom_types_per_atom")
        return out
    assert q_bar.shape == (Cn, Cn)
    rc = lib().mdx_noise_atom_types(ptr(a0, I64, "a0"), ptr(q_bar, F32, "q_bar"), ptr(u, F32, "u"), a0.numel(), Cn,
                                    ptr(out, I64, "out"), stream_handle())
    check(rc, "mdx_noise_atom_types")
    return out


def repaint_constrained_rows(sched: DeviceSchedule, index_i: int, d_index, constrained_x, constrained_a,
                             constrained_indices, z, u, rng: Rng, x_inout, a_inout):
    B, N, d = x_inout.shape
    K = constrained_x.shape[0]
    rc = lib().mdx_repaint_constrained_rows(
        C.byref(sched.c_struct), int(index_i), ptr(d_index, I32, "d_index"), ptr(constrained_x, F32, "constrained_x"),
        ptr(constrained_a, I64, "constrained_a"), ptr(constrained_indices, I64, "constrained_indices"), K,
        ptr(z, F32, "z"), ptr(u, F32, "u"), rng, B, N, d, ptr(x_inout, F32, "x"), ptr(a_inout, I64, "a"),
        stream_handle())
    check(rc, "mdx_repaint_constrained_rows")


def forward_diffusion_step(sched: DeviceSchedule, index_i: int, d_index, z, u, rng: Rng, x_inout, a_inout):
    """RePaint resampling: one forward-process step i -> i+1 on every atom, in place (mdx_forward_diffusion_step)."""
    B, N, d = x_inout.shape
    rc = lib().mdx_forward_diffusion_step(C.byref(sched.c_struct), int(index_i), ptr(d_index, I32, "d_index"),
                                          ptr(z, F32, "z"), ptr(u, F32, "u"), rng, B, N, d, ptr(x_inout, F32, "x"),
                                          ptr(a_inout, I64, "a"), stream_handle())
    check(rc, "mdx_forward_diffusion_step")


# ----------------------------------------------------------------------------------------------------------------
# N1
# ----------------------------------------------------------------------------------------------------------------
def radius_graph(cartesian_positions, basis_vectors, radial_cutoff: float, unique: bool,
                 status: Optional[torch.Tensor] = None, want_shifts: bool = True):
    """Two-call radius graph.  Returns dict(counts [B,N], edges [E,2], image [E] | None, shifts [E,3] | None).

    The only host synchronisation is reading E = counts.sum() to size the outputs.
    """
    B, N, d = cartesian_positions.shape
    assert d == 3 and basis_vectors.shape == (B, 3, 3)
    dev = cartesian_positions.device
    counts = torch.empty(B, N, dtype=I64, device=dev)
    L = lib()
    rc = L.mdx_radius_graph_count(ptr(cartesian_positions, F32, "cartesian_positions"),
                                  ptr(basis_vectors, F32, "basis_vectors"), float(radial_cutoff), B, N,
                                  int(bool(unique)), ptr(counts, I64, "counts"), ptr(status, I32, "status"),
                                  stream_handle())
    check(rc, "mdx_radius_graph_count")
    inclusive = torch.cumsum(counts.view(-1), 0)
    offsets = inclusive - counts.view(-1)
    E = int(inclusive[-1].item()) if B * N > 0 else 0
    edges = torch.empty(E, 2, dtype=I64, device=dev)
    image = None if unique else torch.empty(E, dtype=I32, device=dev)
    shifts = None if (unique or not want_shifts) else torch.empty(E, 3, dtype=F32, device=dev)
    if E > 0:
        rc = L.mdx_radius_graph_fill(ptr(cartesian_positions, F32, "cartesian_positions"),
                                     ptr(basis_vectors, F32, "basis_vectors"), float(radial_cutoff), B, N,
                                     int(bool(unique)), ptr(offsets, I64, "offsets"), ptr(edges, I64, "edges"),
                                     ptr(image, I32, "image"), ptr(shifts, F32, "shifts"), stream_handle())
        check(rc, "mdx_radius_graph_fill")
    return dict(counts=counts, edges=edges, image=image, shifts=shifts)


def radius_graph_static(cartesian_positions, basis_vectors, radial_cutoff: float, capacity: int,
                        status: Optional[torch.Tensor] = None):
    """Unique-pair radius graph into a caller-sized edge list, with NO host synchronisation (capturable into a hipGraph):
    returns dict(counts [B*N], offsets [B*N], edges [capacity, 2], n_edges int64 [1] on the device).  Rows beyond
    n_edges are uninitialised; more than `capacity` edges sets STATUS_GRAPH_CAPACITY in `status`."""
    B, N, d = cartesian_positions.shape
    assert d == 3 and basis_vectors.shape == (B, 3, 3)
    dev = cartesian_positions.device
    counts = torch.empty(B * N, dtype=I64, device=dev)
    L = lib()
    check(L.mdx_radius_graph_count(ptr(cartesian_positions, F32, "cartesian_positions"),
                                   ptr(basis_vectors, F32, "basis_vectors"), float(radial_cutoff), B, N, 1,
                                   ptr(counts, I64, "counts"), ptr(status, I32, "status"), stream_handle()),
          "mdx_radius_graph_count")
    inclusive = torch.cumsum(counts, 0)
    offsets = inclusive - counts
    edges = torch.empty(int(capacity), 2, dtype=I64, device=dev)
    check(L.mdx_radius_graph_fill_capped(ptr(cartesian_positions, F32, "cartesian_positions"),
                                         ptr(basis_vectors, F32, "basis_vectors"), float(radial_cutoff), B, N, 1,
                                         ptr(offsets, I64, "offsets"), int(capacity), ptr(edges, I64, "edges"), None, None,
                                         ptr(status, I32, "status"), stream_handle()), "mdx_radius_graph_fill_capped")
    return dict(counts=counts, offsets=offsets, edges=edges, n_edges=inclusive[-1:])


def egnn_radius_graph(relative_coordinates, lattice_parameters, clip_min: float, radial_cutoff: float, capacity: int,
                      status: Optional[torch.Tensor] = None, two_launches: bool = True):
    """radius_graph_static for the graph EGNNScoreNetwork builds (egnn_score_network.py:236-247): relative coordinates
    [B,N,3] in the cell diag(max(lattice_parameters[:, :3], clip_min)) behind ONE call, no library kernel, no host read
    (mdx_egnn_radius_graph): hit masks + emission (two launches, every pair tested once) where that form applies
    (N <= 1024, B <= 2048) and `two_launches`, else count, device-side scan and fill.  Same dict as radius_graph_static."""
    B, N, d = relative_coordinates.shape
    assert d == 3 and lattice_parameters.dim() == 2 and lattice_parameters.shape[0] == B and lattice_parameters.shape[1] >= 3
    dev = relative_coordinates.device
    counts = torch.empty(B * N, dtype=I64, device=dev)
    offsets = torch.empty(B * N, dtype=I64, device=dev)
    n_edges = torch.empty(1, dtype=I64, device=dev)
    edges = torch.empty(int(capacity), 2, dtype=I64, device=dev)
    words = int(lib().mdx_egnn_radius_graph_workspace_words(B, N)) if two_launches else 0
    workspace = torch.empty(words, dtype=I64, device=dev) if words else None
    check(lib().mdx_egnn_radius_graph(ptr(relative_coordinates, F32, "relative_coordinates"),
                                      ptr(lattice_parameters, F32, "lattice_parameters"), lattice_parameters.shape[1],
                                      float(clip_min), float(radial_cutoff), B, N, int(capacity), ptr(counts, I64, "counts"),
                                      ptr(offsets, I64, "offsets"), ptr(n_edges, I64, "n_edges"), ptr(edges, I64, "edges"),
                                      ptr(status, I32, "status"), ptr(workspace, I64, "workspace"), words, stream_handle()),
          "mdx_egnn_radius_graph")
    return dict(counts=counts, offsets=offsets, edges=edges, n_edges=n_edges)


# ----------------------------------------------------------------------------------------------------------------
# fused MLP score network
# ----------------------------------------------------------------------------------------------------------------
class MlpPack:
    """Device copy of an MLPScoreNetwork's parameters in the layout of mdx_mlp_t (transposed weights, [in, out]).

    Built from the module's state at construction; rebuild it if the parameters change."""

    def __init__(self, network, device):
        hp = network._hyper_params
        if getattr(network, "use_permutation_invariance", False) or getattr(network, "use_time_dependent_prefactor", False):
            raise _hip.MdxError("the fused MLP kernels implement the plain unconditional forward "
                                "(no permutation symmetrisation, no time prefactor)")
        n_hidden = len(network.mlp_layers)
        if n_hidden > _hip.MLP_MAX_HIDDEN:
            raise _hip.MdxError(f"at most {_hip.MLP_MAX_HIDDEN} hidden layers are supported by the fused MLP kernels")
        self._keep = []

        def wt(linear):
            t = linear.weight.detach().to(device=device, dtype=F32).t().contiguous()
            self._keep.append(t)
            return t.data_ptr()

        def bias(linear):
            t = linear.bias.detach().to(device=device, dtype=F32).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        m = Mlp()
        m.number_of_atoms, m.spatial_dimension, m.num_classes = network._natoms, network.spatial_dimension, network.num_classes
        m.hidden_size, m.n_hidden = hp.hidden_dimensions_size, n_hidden
        m.e_coordinates = hp.relative_coordinates_embedding_dimensions_size
        m.e_noise, m.e_time = hp.noise_embedding_dimensions_size, hp.time_embedding_dimensions_size
        m.e_atom_type = hp.atom_type_embedding_dimensions_size
        m.e_lattice = hp.lattice_parameters_embedding_dimensions_size
        for name, layer in (("coordinates", network.relative_coordinates_embedding_layer),
                            ("noise", network.noise_embedding_layer), ("time", network.time_embedding_layer),
                            ("atom_type", network.atom_type_embedding_layer),
                            ("lattice", network.lattice_parameters_embedding_layer)):
            setattr(m, f"w_{name}_t", wt(layer))
            setattr(m, f"b_{name}", bias(layer))
        for k, layer in enumerate(network.mlp_layers):
            m.w_hidden_t[k] = wt(layer)
            m.b_hidden[k] = bias(layer)
        for name, layer in (("a", network.output_A_layer), ("x", network.output_X_layer), ("l", network.output_L_layer)):
            setattr(m, f"w_out_{name}_t", wt(layer))
            setattr(m, f"b_out_{name}", bias(layer))
        m.packed_image = None
        m.folded_input = None
        self.folded = _fold_input_layers(network, m, device)
        m.folded_output = None
        self.folded_out = _fold_output_layers(network, device) if self.folded is not None else None
        m.folded_padded = None
        if self.folded is not None and self.folded_out is not None:
            m.folded_input = self.folded.data_ptr()
            m.folded_output = self.folded_out.data_ptr()
            self.folded_padded = _pad_folded_layers(network, m, device)
            if self.folded_padded is not None:
                m.folded_padded = self.folded_padded.data_ptr()
        n_floats = lib().mdx_mlp_image_floats(C.byref(m))
        if n_floats > 0:      # the kernels' own layout, built once: kernel start-up becomes one coalesced copy
            self.image = torch.empty(n_floats, dtype=F32, device=device)
            with torch.cuda.device(device):
                check(lib().mdx_mlp_pack_image(C.byref(m), ptr(self.image, F32, "image"), stream_handle()),
                      "mdx_mlp_pack_image")
            m.packed_image = self.image.data_ptr()
        self.c_struct = m
        self.device = torch.device(device)
        self.number_of_atoms, self.num_classes, self.spatial_dimension = m.number_of_atoms, m.num_classes, m.spatial_dimension


def _fold_input_layers(network, m, device):
    """The five embedding layers folded into hidden layer 0 (mdx_mlp_t.folded_input): they are linear and no activation
    lies between them and the first hidden layer (mlp_score_network.py:299-344).  Products in binary64, rounded once.
    Input of the folded layer: [cos (N d) | sin (N d) | sigma | t | atom-type embeddings (N e_a) | lattice emb. (e_l)]."""
    f64 = torch.float64
    with torch.no_grad():
        w0 = network.mlp_layers[0].weight.detach().to(f64).cpu()           # [H, ec + en + et + N ea + el]
        b0 = network.mlp_layers[0].bias.detach().to(f64).cpu()
        ec, en, et = m.e_coordinates, m.e_noise, m.e_time
        na, el = m.number_of_atoms * m.e_atom_type, m.e_lattice
        if w0.shape[1] != ec + en + et + na + el:
            return None
        o1, o2, o3, o4 = ec, ec + en, ec + en + et, ec + en + et + na
        wc, bc = (t.detach().to(f64).cpu() for t in (network.relative_coordinates_embedding_layer.weight,
                                                       network.relative_coordinates_embedding_layer.bias))
        wn, bn = (t.detach().to(f64).cpu() for t in (network.noise_embedding_layer.weight,
                                                       network.noise_embedding_layer.bias))
        wt_, bt = (t.detach().to(f64).cpu() for t in (network.time_embedding_layer.weight,
                                                        network.time_embedding_layer.bias))
        columns = torch.cat([w0[:, :o1] @ wc,                     # [H, 2 N d]: cos block then sin block, as the module
                             w0[:, o1:o2] @ wn,                   # [H, 1]  sigma
                             w0[:, o2:o3] @ wt_,                  # [H, 1]  time
                             w0[:, o3:o4],                        # [H, N ea]
                             w0[:, o4:]], dim=1)                  # [H, el]
        bias = b0 + w0[:, :o1] @ bc + w0[:, o1:o2] @ bn + w0[:, o2:o3] @ bt
        return torch.cat([_quad_image(columns), bias]).to(device=device, dtype=F32).contiguous()


def _folded_matrices(network, m):
    """(W_first [H, F], b_first [H]), [(W_k, b_k) of the middle hidden layers], (W_out [outputs, H], b_out) in binary64: the
    network as n_hidden linear maps (see _fold_input_layers / _fold_output_layers)."""
    f64 = torch.float64
    with torch.no_grad():
        w0 = network.mlp_layers[0].weight.detach().to(f64).cpu()
        b0 = network.mlp_layers[0].bias.detach().to(f64).cpu()
        ec, en, et = m.e_coordinates, m.e_noise, m.e_time
        na = m.number_of_atoms * m.e_atom_type
        o1, o2, o3, o4 = ec, ec + en, ec + en + et, ec + en + et + na
        cpu = lambda t: t.detach().to(f64).cpu()       # noqa: E731
        wc, bc = cpu(network.relative_coordinates_embedding_layer.weight), cpu(network.relative_coordinates_embedding_layer.bias)
        wn, bn = cpu(network.noise_embedding_layer.weight), cpu(network.noise_embedding_layer.bias)
        wt_, bt = cpu(network.time_embedding_layer.weight), cpu(network.time_embedding_layer.bias)
        first = torch.cat([w0[:, :o1] @ wc, w0[:, o1:o2] @ wn, w0[:, o2:o3] @ wt_, w0[:, o3:o4], w0[:, o4:]], dim=1)
        first_bias = b0 + w0[:, :o1] @ bc + w0[:, o1:o2] @ bn + w0[:, o2:o3] @ bt
        mids = [(cpu(layer.weight), cpu(layer.bias)) for layer in list(network.mlp_layers)[1:-1]]
        last = network.mlp_layers[-1]
        heads = (network.output_A_layer, network.output_X_layer, network.output_L_layer)
        w_heads = torch.cat([cpu(h.weight) for h in heads], dim=0)
        b_heads = torch.cat([cpu(h.bias) for h in heads], dim=0)
        return (first, first_bias), mids, (w_heads @ cpu(last.weight), w_heads @ cpu(last.bias) + b_heads)


def _pad_folded_layers(network, m, device):
    """mdx_mlp_t.folded_padded: the folded layers zero-padded to the fixed sizes of the padded register-resident family
    (64 neurons, 16 / 32 / 48 first-layer quads, 64 outputs), or None when the network is outside that family's limits.
    The same binary64 products as folded_input / folded_output, rounded once: the padded family and the generic folded forward
    compute the same bits."""
    n_hidden = len(network.mlp_layers)
    n_in = 2 * m.number_of_atoms * m.spatial_dimension + 2 + m.number_of_atoms * m.e_atom_type + m.e_lattice
    d = m.spatial_dimension
    n_out = m.number_of_atoms * m.num_classes + m.number_of_atoms * d + d * (d + 1) // 2
    if m.hidden_size > 64 or m.number_of_atoms > 8 or not 2 <= n_hidden <= 4 or n_out > 64 or n_in > 192:
        return None
    (w_first, b_first), mids, (w_out, b_out) = _folded_matrices(network, m)
    if w_first.shape[1] != n_in:
        return None

    def padded(w, b, inputs):
        wp = torch.zeros(64, inputs, dtype=torch.float64)
        wp[: w.shape[0], : w.shape[1]] = w
        bp = torch.zeros(64, dtype=torch.float64)
        bp[: b.shape[0]] = b
        return [_quad_image(wp), bp]

    parts = padded(w_first, b_first, 64 * ((n_in + 63) // 64))
    for w, b in mids:
        parts += padded(w, b, 64)
    parts += padded(w_out, b_out, 64)
    return torch.cat(parts).to(device=device, dtype=F32).contiguous()


def _quad_image(matrix64: torch.Tensor) -> torch.Tensor:
    """[outputs, inputs] matrix -> the kernels' [ceil(inputs/4)][outputs][4] layout (inputs zero-padded)."""
    n_out, n_in = matrix64.shape
    quads = (n_in + 3) // 4
    padded = torch.zeros(n_out, quads * 4, dtype=matrix64.dtype)
    padded[:, :n_in] = matrix64
    return padded.t().reshape(quads, 4, n_out).permute(0, 2, 1).contiguous().reshape(-1)


def _fold_output_layers(network, device):
    """The last hidden layer (no activation follows it, mlp_score_network.py:337-344) folded into the three output
    heads (mdx_mlp_t.folded_output): outputs ordered logits | score_x | score_l.  Needs >= 2 hidden layers."""
    if len(network.mlp_layers) < 2:
        return None
    f64 = torch.float64
    with torch.no_grad():
        last = network.mlp_layers[-1]
        w_last, b_last = last.weight.detach().to(f64).cpu(), last.bias.detach().to(f64).cpu()
        heads = (network.output_A_layer, network.output_X_layer, network.output_L_layer)
        w_heads = torch.cat([h.weight.detach().to(f64).cpu() for h in heads], dim=0)        # [N C + N d + nl, H]
        b_heads = torch.cat([h.bias.detach().to(f64).cpu() for h in heads], dim=0)
        folded = w_heads @ w_last                                                           # [outputs, H_in of last]
        bias = w_heads @ b_last + b_heads
        return torch.cat([_quad_image(folded), bias]).to(device=device, dtype=F32).contiguous()



def mlp_forward(pack: MlpPack, atom_types, x, l, time, sigma):
    """Fused forward of the MLP score network: returns (logits [B,N,C] with MASK = -inf, score_x, score_l)."""
    B, N, d = x.shape
    assert N == pack.number_of_atoms and d == pack.spatial_dimension
    logits = torch.empty(B, N, pack.num_classes, dtype=F32, device=x.device)
    score_x = torch.empty_like(x)
    score_l = torch.empty_like(l)
    rc = lib().mdx_mlp_forward(C.byref(pack.c_struct), ptr(atom_types, I64, "atom_types"), ptr(x, F32, "x"),
                               ptr(l, F32, "l"), ptr(time, F32, "time"), ptr(sigma, F32, "sigma"), B,
                               ptr(logits, F32, "logits"), ptr(score_x, F32, "score_x"), ptr(score_l, F32, "score_l"),
                               stream_handle())
    check(rc, "mdx_mlp_forward")
    return logits, score_x, score_l


NOISE_WORKSPACE_MAX_FLOATS = 1 << 28        # 1 GiB cap; longer segments are split into several launches by the library
NOISE_WORKSPACE_MIN_ITERATIONS = 1024       # first allocation holds at least this many iterations (within the cap)


class NoiseWorkspace:
    """The pre-drawn-noise buffer of mdx_mlp_pc_sample: sized by the library, grown on demand, owned by ONE caller (a
    generator or a bench loop), so that two samplers on different streams never share records.  A buffer that is
    replaced is handed to the allocator with record_stream, i.e. it is not reused before the launches that read it
    have completed."""

    def __init__(self):
        self.buffer = None

    def floats_needed(self, pack: "MlpPack", number_of_corrector_steps: int, atom_type_transition_in_corrector: bool,
                      n_iterations: int, batch: int) -> int:
        need = lib().mdx_mlp_pc_sample_workspace_floats(C.byref(pack.c_struct), int(number_of_corrector_steps),
                                                        int(bool(atom_type_transition_in_corrector)), int(n_iterations),
                                                        int(batch))
        if need < 0:
            raise _hip.MdxError("mdx_mlp_pc_sample_workspace_floats: invalid argument")
        return need

    def get(self, pack: "MlpPack", number_of_corrector_steps: int, atom_type_transition_in_corrector: bool,
            n_iterations: int, batch: int, device) -> torch.Tensor:
        need = self.floats_needed(pack, number_of_corrector_steps, atom_type_transition_in_corrector, n_iterations, batch)
        per_iteration = need // max(int(n_iterations), 1)
        need = max(min(need, NOISE_WORKSPACE_MAX_FLOATS), per_iteration, 1)
        buf = self.buffer
        if buf is None or buf.numel() < need or buf.device != torch.device(device):
            # allocate with headroom (a device allocation costs ~100 us, more than a short segment's kernels): room for
            # NOISE_WORKSPACE_MIN_ITERATIONS iterations of this shape, within the cap
            roomy = min(max(need, NOISE_WORKSPACE_MIN_ITERATIONS * per_iteration), max(NOISE_WORKSPACE_MAX_FLOATS, need))
            if buf is not None and buf.is_cuda:
                buf.record_stream(torch.cuda.current_stream(buf.device))
            buf = self.buffer = torch.empty(roomy, dtype=F32, device=device)
        return buf


def mlp_pc_sample(sched: DeviceSchedule, pack: MlpPack, flags: PcFlags, number_of_corrector_steps: int,
                  atom_type_transition_in_corrector: bool, start_index: int, n_iterations: int, rng: Rng,
                  atom_types, x, l, status, workspace: Optional[NoiseWorkspace] = None, options: int = 0,
                  caller_records: Optional[torch.Tensor] = None):
    """n_iterations x (predictor + M correctors) in one launch, composition updated in place.

    workspace: the caller's NoiseWorkspace -- the segment's draws are generated by the chip-filling pre-pass kernel
    into it (default product path); None: every wavefront draws in-kernel.  Both evaluate the same Philox
    specification: identical results.  caller_records: a float32 device tensor already holding the records (layout in
    include/mdx_hip.h) -- parity tests replay the reference's recorded draws this way (MLP_SAMPLE_CALLER_NOISE).
    options: MLP_SAMPLE_* bits of _hip.py, passed through to the library."""
    B = x.shape[0]
    work = None
    if caller_records is not None:
        work = caller_records
        options |= _hip.MLP_SAMPLE_CALLER_NOISE
    elif workspace is not None and B > 0 and n_iterations > 0:
        work = workspace.get(pack, number_of_corrector_steps, atom_type_transition_in_corrector, n_iterations, B,
                             x.device)
    rc = lib().mdx_mlp_pc_sample(C.byref(sched.c_struct), C.byref(pack.c_struct), C.byref(flags),
                                 int(number_of_corrector_steps), int(bool(atom_type_transition_in_corrector)),
                                 int(start_index), int(n_iterations), rng, B, ptr(atom_types, I64, "atom_types"),
                                 ptr(x, F32, "x"), ptr(l, F32, "l"), ptr(work, F32, "noise_workspace"),
                                 0 if work is None else work.numel(), int(options), ptr(status, I32, "status"),
                                 stream_handle())
    check(rc, "mdx_mlp_pc_sample")


# ----------------------------------------------------------------------------------------------------------------
# EGNN helpers around the MFMA kernels: fused first message layer, sorted-segment reductions
# ----------------------------------------------------------------------------------------------------------------
def egnn_message_input(node_proj, edges, radial, bias, w_radial, silu: bool = True) -> torch.Tensor:
    """First message layer on an edge list: SiLU(P[src,:H] + P[dst,H:] + b + r w_r)  (mdx_egnn_message_input)."""
    E = edges.shape[0]
    H = bias.shape[0]
    assert node_proj.shape[1] == 2 * H and radial.numel() == E
    out = torch.empty(E, H, dtype=F32, device=node_proj.device)
    rc = lib().mdx_egnn_message_input(ptr(node_proj, F32, "node_proj"), ptr(edges, I64, "edges"),
                                      ptr(radial, F32, "radial"), ptr(bias, F32, "bias"), ptr(w_radial, F32, "w_radial"),
                                      E, H, int(bool(silu)), ptr(out, F32, "out"), stream_handle())
    check(rc, "mdx_egnn_message_input")
    return out


def egnn_coord_head(hidden, w_out, coord_diff, offsets, degree, mean: bool) -> torch.Tensor:
    """Last coordinate-MLP layer (H -> 1, no bias) x coord_diff, summed (or averaged) over each node's sorted edges."""
    n_nodes, H, d = degree.shape[0], hidden.shape[1], coord_diff.shape[1]
    if hidden.shape[0] == 0:                                   # no edges at all
        return torch.zeros(n_nodes, d, dtype=F32, device=hidden.device)
    trans = torch.empty(n_nodes, d, dtype=F32, device=hidden.device)
    rc = lib().mdx_egnn_coord_head(ptr(hidden, F32, "hidden"), ptr(w_out, F32, "w_out"), ptr(coord_diff, F32, "coord_diff"),
                                   ptr(offsets, I64, "offsets"), ptr(degree, I64, "degree"), n_nodes, H, d,
                                   int(bool(mean)), ptr(trans, F32, "trans"), stream_handle())
    check(rc, "mdx_egnn_coord_head")
    return trans


def segment_rows(data, offsets, degree, mean: bool) -> torch.Tensor:
    """Sum (or mean) of the rows of data [E,H] over each node's sorted edge segment -> [n_nodes, H]."""
    n_nodes, H = degree.shape[0], data.shape[1]
    if data.shape[0] == 0:
        return torch.zeros(n_nodes, H, dtype=F32, device=data.device)
    out = torch.empty(n_nodes, H, dtype=F32, device=data.device)
    rc = lib().mdx_segment_rows(ptr(data, F32, "data"), ptr(offsets, I64, "offsets"), ptr(degree, I64, "degree"), n_nodes,
                                H, int(bool(mean)), ptr(out, F32, "out"), stream_handle())
    check(rc, "mdx_segment_rows")
    return out


# ----------------------------------------------------------------------------------------------------------------
# fused EGNN edge chain on the matrix cores (csrc/mdx_egnn_chain.hip)
# ----------------------------------------------------------------------------------------------------------------
# "f32": exact binary32 MFMA (v_mfma_f32_32x32x2_f32); "f16x3": split-f16, three products per term, on
# v_mfma_f32_16x16x32_f16 (the default since round 3: the chip holds a higher clock on this shape); "f16x3_32x32": the same
# arithmetic on v_mfma_f32_32x32x16_f16 (the round-2 kernel, kept for A/B runs and as a second implementation in the tests)
EDGE_CHAIN_PRECISIONS = {"f32": 0, "f16x3": 2, "f16x3_32x32": 1}


def _pack_chain_image(weights, w_out, H: int, precision: str, tied_layers: int = 0):
    """(image, exponents) of mdx_egnn_chain_pack for device matrices `weights` ([H, H] each, nn.Linear layout) and the
    optional head row w_out [H].  exponents: int32 [len(weights) + 1] on the device -- the per-layer powers of two the
    split-f16 image is scaled by (zeros for "f32"), chosen and written by the library without a host read."""
    dev = weights[0].device
    keep = [w.detach().to(F32).contiguous() for w in weights]
    image = torch.empty(lib().mdx_egnn_chain_image_bytes(H, len(keep)), dtype=torch.uint8, device=dev)
    exponents = torch.empty(len(keep) + 1, dtype=I32, device=dev)
    array = (C.c_void_p * len(keep))(*[w.data_ptr() for w in keep])
    head = None if w_out is None else w_out.detach().reshape(-1).to(F32).contiguous()
    with torch.cuda.device(dev):
        check(lib().mdx_egnn_chain_pack(array, len(keep), None if head is None else C.c_void_p(head.data_ptr()), H,
                                        EDGE_CHAIN_PRECISIONS[precision], tied_layers, C.c_void_p(image.data_ptr()),
                                        C.c_void_p(exponents.data_ptr()), stream_handle()), "mdx_egnn_chain_pack")
    return image, exponents      # (the temporaries are freed in stream order: the image holds its own copy)


F16_ACTIVATION_EXPONENT = 6        # MDX_EGNN_F16_ACTIVATION_EXPONENT


class ActivationScales:
    """The per-position powers of two the split-f16 kernels carry a chain's activations with, and the maxima the exact-f32
    kernels collect for them (mdx_egnn_chain_t.activation_exponents / activation_maxima): two small device arrays owned by
    whoever owns the chain's parameters and SHARED by the chain's packs of every precision -- the f32 pass that follows a
    range report fills the maxima, adapt() turns them into exponents (on the device, no host read), and the split-f16
    launches that follow -- captured ones included: the kernels read the array at every launch -- carry the hot positions
    with more headroom."""

    def __init__(self, n_layers: int, device):
        self.count = n_layers + 2
        self.exponents = torch.full((self.count,), F16_ACTIVATION_EXPONENT, dtype=I32, device=device)
        self.maxima = torch.zeros(self.count, dtype=I32, device=device)

    def adapt(self):
        with torch.cuda.device(self.exponents.device):
            check(lib().mdx_egnn_chain_adapt_activation_exponents(C.c_void_p(self.maxima.data_ptr()), self.count,
                                                                  C.c_void_p(self.exponents.data_ptr()), stream_handle()),
                  "mdx_egnn_chain_adapt_activation_exponents")

    def reset(self):
        """The state of a new object: default exponents, no maxima.  (The exponents only ever go DOWN otherwise -- after a
        fallback the split-f16 kernels carry the hot positions with more headroom and fewer low bits, so a later sample() with
        the same (seed, call index) can differ in the last bits from one made before the fallback: results are a function of
        (seed, call index, exponents), and reset() restores the exponents' initial value.)"""
        self.exponents.fill_(F16_ACTIVATION_EXPONENT)
        self.maxima.zero_()

    def pointers(self, precision: str):
        """(activation_exponents, activation_maxima) of a pack of `precision`: the split kernels read the exponents, the
        exact-f32 kernels write the maxima."""
        return (None, self.maxima.data_ptr()) if precision == "f32" else (self.exponents.data_ptr(), None)


CHAIN_WIDTHS = (32, 64, 128, 256)       # the widths egnn_edge_chain_kernel is instantiated for


def chain_width(*widths) -> int:
    """The instantiated chain width that holds layers of these widths (0: none does)."""
    need = max(widths)
    return next((w for w in CHAIN_WIDTHS if w >= need), 0)


def _pad2(w: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    w = w.detach().to(F32)
    if tuple(w.shape) == (rows, cols):
        return w.contiguous()
    out = torch.zeros(rows, cols, dtype=F32, device=w.device)
    out[:w.shape[0], :w.shape[1]] = w
    return out


def _pad1(v: torch.Tensor, n: int) -> torch.Tensor:
    return _pad2(v.reshape(1, -1), 1, n).reshape(-1)


class EdgeChainPack:
    """Device image of one E_GCL layer's per-edge MLP chain for mdx_egnn_edge_chain: the H -> H weight matrices of the
    message MLP (after its first layer) and of the coordinate MLP, re-laid out by mdx_egnn_chain_pack for `precision`,
    plus the small vectors.  Built from the modules' parameters at construction; `stamp` tells when to rebuild.
    scales (ActivationScales, optional): shared with the layer's packs of the other precisions.

    Widths.  The kernel runs square layers of one width H in {32, 64, 128, 256}.  A message MLP of width m and a coordinate
    MLP of width c (the reference's DEFAULT hyper-parameters are m = 16, c = 32: models/score_networks/egnn_score_network.py:
    23-45) run at H = chain_width(m, c) with every matrix, bias and vector ZERO-PADDED: a padded neuron has zero weights and a
    zero bias, so its pre-activation is 0, SiLU(0) = 0, and it feeds zeros on -- the same function, bit for bit in exact
    arithmetic, at the price of multiplying zeros.  `hidden` = H, `message_width` = m: the message sums come out [.., H] with
    columns m .. H-1 zero."""

    def __init__(self, first_message_layer, message_layers, coord_layers, coord_out_layer, input_size: int, precision: str,
                 scales=None, attention_layer=None):
        """attention_layer: E_GCL.att_mlp's nn.Linear(m, 1) (its Sigmoid is the kernel's), or None."""
        dev = first_message_layer.weight.device
        message_layers, coord_layers = list(message_layers), list(coord_layers)
        layers = message_layers + coord_layers
        if precision not in EDGE_CHAIN_PRECISIONS:
            raise _hip.MdxError(f"edge-chain precision must be one of {sorted(EDGE_CHAIN_PRECISIONS)}; got {precision!r}")
        if not self.supported(first_message_layer, message_layers, coord_layers, coord_out_layer):
            raise _hip.MdxError("this E_GCL shape is not covered by the fused edge chain (see mdx_egnn_edge_chain)")
        m = first_message_layer.out_features
        H = chain_width(m, coord_out_layer.in_features)
        self.precision, self.hidden, self.message_width = precision, H, m
        self.image, self.exponents = _pack_chain_image([_pad2(layer.weight, H, H) for layer in layers],
                                                       _pad1(coord_out_layer.weight, H), H, precision)
        self.biases = torch.stack([_pad1(layer.bias, H) for layer in layers]).contiguous()
        self.bias_in = _pad1(first_message_layer.bias, H)
        self.w_radial = _pad1(first_message_layer.weight.detach()[:, 2 * input_size], H)
        # [2H, n_in]: the per-node projections of the first message layer (source half | destination half) as ONE matrix
        w0 = first_message_layer.weight.detach().to(F32)
        self.proj_weight = torch.cat([_pad2(w0[:, :input_size], H, input_size),
                                      _pad2(w0[:, input_size:2 * input_size], H, input_size)], dim=0).contiguous()
        self.scales = scales
        act = scales.pointers(precision) if scales is not None else (None, None)
        assert scales is None or scales.count == len(layers) + 2
        self.att_w = self.att_b = None
        if attention_layer is not None:
            if attention_layer.in_features != m or attention_layer.out_features != 1 or attention_layer.bias is None:
                raise _hip.MdxError("the attention gate of the fused edge chain is nn.Linear(message width, 1) with a bias")
            self.att_w = _pad1(attention_layer.weight, H)
            self.att_b = attention_layer.bias.detach().to(F32).reshape(-1).contiguous()
        self.c_struct = _hip.EgnnChain(H, len(message_layers), len(coord_layers),
                                       EDGE_CHAIN_PRECISIONS[precision], 0, 0, self.image.data_ptr(),
                                       self.biases.data_ptr(), self.bias_in.data_ptr(), self.w_radial.data_ptr(),
                                       self.exponents.data_ptr(), *act,
                                       None if self.att_w is None else self.att_w.data_ptr(),
                                       None if self.att_b is None else self.att_b.data_ptr())
        # the kernel's LDS: weight ring + small vectors + per-layer scale table + the source ids of the in-kernel aggregation
        # (+ the attention gate's weight row)
        lds = 4 * 32 * H * 4 + 4 * (len(layers) * H + 2 * H) + 16 * (_hip.EGNN_CHAIN_MAX_LAYERS + 4) + 4 * 4 * 32 + 4 * (_hip.EGNN_CHAIN_MAX_LAYERS + 2) + \
            (16 + 4 * (H + 4) if self.att_w is not None else 0)
        self.piece_sums_ok = lds <= 160 * 1024
        self.device = dev

    @staticmethod
    def supported(first_message_layer, message_layers, coord_layers, coord_out_layer) -> bool:
        """nn.Linear stacks m -> m (message, after the first layer), m -> c -> c ... (coordinate), c -> 1 without a bias (head),
        with chain_width(m, c) an instantiated width."""
        message_layers, coord_layers = list(message_layers), list(coord_layers)
        if len(message_layers) < 1 or len(coord_layers) < 1 or \
                len(message_layers) + len(coord_layers) > _hip.EGNN_CHAIN_MAX_LAYERS:
            return False
        m, c = first_message_layer.out_features, coord_out_layer.in_features
        widths_in = [m] * len(message_layers) + [m] + [c] * (len(coord_layers) - 1)
        widths_out = [m] * len(message_layers) + [c] * len(coord_layers)
        return (chain_width(m, c) != 0 and first_message_layer.bias is not None and
                all(l.in_features == i and l.out_features == o and l.bias is not None
                    for l, i, o in zip(message_layers + coord_layers, widths_in, widths_out)) and
                coord_out_layer.out_features == 1 and coord_out_layer.bias is None)


class RowChainPack:
    """Device image of a chain of H -> H nn.Linear layers applied to the rows of a matrix (mdx_mlp_chain_rows): every layer
    but the last is followed by SiLU.  Used for the per-node MLP of an EGNN layer after its first (2H -> H) layer."""

    def __init__(self, layers, precision: str, scales=None):
        layers = list(layers)
        H = layers[0].in_features
        if precision not in EDGE_CHAIN_PRECISIONS:
            raise _hip.MdxError(f"chain precision must be one of {sorted(EDGE_CHAIN_PRECISIONS)}; got {precision!r}")
        if not self.supported(layers):
            raise _hip.MdxError("this layer stack is not covered by mdx_mlp_chain_rows")
        dev = layers[0].weight.device
        self.precision, self.hidden = precision, H
        self.image, self.exponents = _pack_chain_image([layer.weight for layer in layers], None, H, precision)
        self.biases = torch.stack([layer.bias.detach().to(F32) for layer in layers]).contiguous()
        self.scales = scales
        act = scales.pointers(precision) if scales is not None else (None, None)
        assert scales is None or scales.count == len(layers) + 2
        self.c_struct = _hip.EgnnChain(H, len(layers), 0, EDGE_CHAIN_PRECISIONS[precision], 0, 0, self.image.data_ptr(),
                                       self.biases.data_ptr(), None, None, self.exponents.data_ptr(), *act, None, None)

    @staticmethod
    def supported(layers) -> bool:
        layers = list(layers)
        if not layers:
            return False
        H = layers[0].in_features
        return (H in (32, 64, 128, 256) and len(layers) <= _hip.EGNN_CHAIN_MAX_LAYERS and
                all(l.in_features == H and l.out_features == H and l.bias is not None for l in layers))


class NodeMlpPack:
    """Device image of a whole EGNN node MLP -- Linear(2H, H), SiLU, [Linear(H, H), SiLU]*, Linear(H, H) -- for
    mdx_node_mlp_rows: the first layer's [H, 2H] weight as two H x H chain layers."""

    def __init__(self, layers, precision: str, next_projection=None, scales=None):
        """next_projection: [2H, H] = the next graph layer's per-node projection weight (EdgeChainPack.proj_weight): its
        two H x H halves follow the MLP in the image, and node_mlp_rows also returns out @ next_projection.T.
        scales: ActivationScales(n_chain_layers(layers), device), shared between the precisions."""
        layers = list(layers)
        if precision not in EDGE_CHAIN_PRECISIONS:
            raise _hip.MdxError(f"chain precision must be one of {sorted(EDGE_CHAIN_PRECISIONS)}; got {precision!r}")
        if not self.supported(layers):
            raise _hip.MdxError("this layer stack is not covered by mdx_node_mlp_rows")
        H = layers[0].out_features
        dev = layers[0].weight.device
        self.precision, self.hidden = precision, H
        w0 = layers[0].weight.detach().to(F32)
        keep = [w0[:, :H].contiguous(), w0[:, H:].contiguous()] + [l.weight.detach().to(F32).contiguous() for l in layers[1:]]
        n = len(keep)
        self.projects = next_projection is not None and n + 2 <= _hip.EGNN_CHAIN_MAX_LAYERS
        if self.projects:
            assert tuple(next_projection.shape) == (2 * H, H)
            keep += [next_projection[:H].detach().to(F32).contiguous(), next_projection[H:].detach().to(F32).contiguous()]
        # tied: the two halves of the wide first layer (their accumulators continue one another) and the two halves of the
        # projection share a power of two each
        tied = (1 << 1) | ((1 << (n + 1)) if self.projects else 0)
        self.image, self.exponents = _pack_chain_image(keep, None, H, precision, tied_layers=tied)
        zeros = torch.zeros(H, dtype=F32, device=dev)
        self.biases = torch.stack([layers[0].bias.detach().to(F32), zeros] +
                                  [l.bias.detach().to(F32) for l in layers[1:]]).contiguous()
        self.scales = scales
        act = scales.pointers(precision) if scales is not None else (None, None)
        assert scales is None or scales.count == n + 2
        self.c_struct = _hip.EgnnChain(H, n, 0, EDGE_CHAIN_PRECISIONS[precision], 0, 0, self.image.data_ptr(),
                                       self.biases.data_ptr(), None, None, self.exponents.data_ptr(), *act)

    @staticmethod
    def n_chain_layers(layers) -> int:
        """chain layers of the MLP part of the image: the wide first layer counts twice"""
        return len(list(layers)) + 1

    @staticmethod
    def supported(layers) -> bool:
        layers = list(layers)
        if len(layers) < 2:
            return False
        H = layers[0].out_features
        return (H in (32, 64, 128, 256) and layers[0].in_features == 2 * H and len(layers) + 1 <= _hip.EGNN_CHAIN_MAX_LAYERS and
                all(l.in_features == H and l.out_features == H for l in layers[1:]) and all(l.bias is not None for l in layers))


def node_mlp_rows(pack: NodeMlpPack, node_in, add_residual: bool, status=None, agg=None):
    """(h if add_residual) + MLP([h | agg]) over the rows; with a pack built with next_projection: (out, out @
    next_projection.T [M, 2H]).  node_in [M, 2H] = [h | agg] (mdx_node_mlp_rows), or node_in = h [M, H] with agg [M, H] given
    separately (mdx_node_mlp_rows_split: the concatenation is never formed)."""
    M, W = node_in.shape
    out = torch.empty(M, pack.hidden, dtype=F32, device=node_in.device)
    proj = torch.empty(M, 2 * pack.hidden, dtype=F32, device=node_in.device) if pack.projects else None
    if agg is not None:
        assert W == pack.hidden and agg.shape == node_in.shape
        rc = lib().mdx_node_mlp_rows_split(C.byref(pack.c_struct), ptr(node_in, F32, "h"), ptr(agg, F32, "agg"),
                                           int(bool(add_residual)), M, None, ptr(out, F32, "out"), ptr(proj, F32, "proj_out"),
                                           ptr(status, I32, "status"), stream_handle())
        check(rc, "mdx_node_mlp_rows_split")
        return (out, proj) if pack.projects else out
    assert W == 2 * pack.hidden
    rc = lib().mdx_node_mlp_rows(C.byref(pack.c_struct), ptr(node_in, F32, "node_in"), int(bool(add_residual)), M, None,
                                 ptr(out, F32, "out"), ptr(proj, F32, "proj_out"), ptr(status, I32, "status"), stream_handle())
    check(rc, "mdx_node_mlp_rows")
    return (out, proj) if pack.projects else out


def mlp_chain_rows(pack: RowChainPack, x, residual=None, status=None) -> torch.Tensor:
    """residual + chain(x) over the rows of x [M, H] (mdx_mlp_chain_rows)."""
    M, H = x.shape
    assert H == pack.hidden and (residual is None or residual.shape == x.shape)
    out = torch.empty_like(x)
    rc = lib().mdx_mlp_chain_rows(C.byref(pack.c_struct), ptr(x, F32, "x"), ptr(residual, F32, "residual"), M, None,
                                  ptr(out, F32, "out"), ptr(status, I32, "status"), stream_handle())
    check(rc, "mdx_mlp_chain_rows")
    return out


def egnn_edge_chain(pack: EdgeChainPack, node_proj, coord, edges, status=None, n_edges_dev=None, piece_sums: bool = False):
    """messages [E,H], edge_scalar [E] of the fused per-edge chain (mdx_egnn_edge_chain); edges sorted by source.
    n_edges_dev (int64 [1], device): the actual number of edge rows when `edges` is a capacity-sized list.
    piece_sums: the first output holds per-node piece sums instead of the messages, in the compact layout of
    mdx_egnn_piece_rows(E, n_nodes) = ceil(E / 16) + n_nodes rows (feed it to segment_combine with n_edges=E): no [E, H]
    buffer exists in that mode."""
    E, H = edges.shape[0], pack.hidden
    pack.c_struct.message_mode = 1 if piece_sums else 0
    assert node_proj.shape[1] == 2 * H and coord.shape[0] == node_proj.shape[0]
    rows = lib().mdx_egnn_piece_rows(E, node_proj.shape[0]) if piece_sums else E
    messages = torch.empty(rows, H, dtype=F32, device=edges.device)
    scalar = torch.empty(E, dtype=F32, device=edges.device)
    rc = lib().mdx_egnn_edge_chain(C.byref(pack.c_struct), ptr(node_proj, F32, "node_proj"), ptr(coord, F32, "coord"),
                                   coord.shape[1], ptr(edges, I64, "edges"), E, ptr(n_edges_dev, I64, "n_edges_dev"),
                                   ptr(messages, F32, "messages"), ptr(scalar, F32, "edge_scalar"),
                                   ptr(status, I32, "status"), stream_handle())
    check(rc, "mdx_egnn_edge_chain")
    return messages, scalar


def segment_combine(pieces, n_edges: int, offsets, degree, mean: bool, left=None) -> torch.Tensor:
    """Sum (or mean) over each node's edges from the piece sums of egnn_edge_chain(..., piece_sums=True) over `n_edges` edge
    rows (the capacity that call was given) -> [n_nodes, H]; with `left` [n_nodes, H]: [left | sums], [n_nodes, 2H] (the
    node MLP's input, without a separate concatenation)."""
    n_nodes, H = degree.shape[0], pieces.shape[1]
    assert left is None or tuple(left.shape) == (n_nodes, H)
    assert pieces.shape[0] == lib().mdx_egnn_piece_rows(n_edges, n_nodes), "pieces: not the compact layout of n_edges, n_nodes"
    out = torch.empty(n_nodes, H if left is None else 2 * H, dtype=F32, device=pieces.device)
    rc = lib().mdx_segment_combine(ptr(pieces, F32, "pieces"), n_edges, ptr(offsets, I64, "offsets"), ptr(degree, I64, "degree"),
                                   n_nodes, H, int(bool(mean)), ptr(left, F32, "left"), ptr(out, F32, "out"), stream_handle())
    check(rc, "mdx_segment_combine")
    return out


def coord_flags(normalize: bool, tanh: bool) -> int:
    """MDX_EGNN_COORD_* bits of an E_GCL layer's coordinate update (models/egnn.py: `normalize`, `tanh`)."""
    return (_hip.EGNN_COORD_NORMALIZE if normalize else 0) | (_hip.EGNN_COORD_TANH if tanh else 0)


def egnn_node_gather(pieces, n_edges: int, offsets, degree, mean_messages: bool, left, edge_scalar, coord, edges,
                     mean_coords: bool, flags: int = 0):
    """segment_combine(..., left=left) and egnn_coord_aggregate(...) in one launch (mdx_egnn_node_gather):
    ([left | message sums] [n_nodes, 2H] (or the sums [n_nodes, H] when left is None), coord_out [n_nodes, D])."""
    n_nodes, H = degree.shape[0], pieces.shape[1]
    assert left is None or tuple(left.shape) == (n_nodes, H)
    assert pieces.shape[0] == lib().mdx_egnn_piece_rows(n_edges, n_nodes), "pieces: not the compact layout of n_edges, n_nodes"
    out = torch.empty(n_nodes, H if left is None else 2 * H, dtype=F32, device=pieces.device)
    coord_out = torch.empty_like(coord)
    rc = lib().mdx_egnn_node_gather(ptr(pieces, F32, "pieces"), n_edges, ptr(offsets, I64, "offsets"), ptr(degree, I64, "degree"),
                                    n_nodes, H, int(bool(mean_messages)), ptr(left, F32, "left"), ptr(out, F32, "out"),
                                    ptr(edge_scalar, F32, "edge_scalar"), ptr(coord, F32, "coord"), coord.shape[1],
                                    ptr(edges, I64, "edges"), int(bool(mean_coords)), int(flags),
                                    ptr(coord_out, F32, "coord_out"), stream_handle())
    check(rc, "mdx_egnn_node_gather")
    return out, coord_out


def egnn_node_inputs(x, k_vectors, sigma, atom_types, emb_weight, emb_bias, second=None):
    """z [n_nodes, 2 n_k] (torus uplift) and h [n_nodes, H] (embedding of [sigma | one_hot]) of EGNNScoreNetwork, one launch.
    x [B, N, 3] relative coordinates, sigma [B] (or [B,1]), atom_types [B, N] int64.
    second = (W2 [H2, F], b2 [H2]): a third output, the same input through that linear map (see mdx_egnn_node_inputs)."""
    B, N, d = x.shape
    assert d == 3 and k_vectors.shape[1] == 3
    n_nodes, n_k, (H, F) = B * N, k_vectors.shape[0], emb_weight.shape
    z = torch.empty(n_nodes, 2 * n_k, dtype=F32, device=x.device)
    h = torch.empty(n_nodes, H, dtype=F32, device=x.device)
    w2, b2 = second if second is not None else (None, None)
    assert second is None or (w2.shape[1] == F and b2.shape[0] == w2.shape[0])
    H2 = w2.shape[0] if second is not None else 0
    h2 = torch.empty(n_nodes, H2, dtype=F32, device=x.device) if second is not None else None
    rc = lib().mdx_egnn_node_inputs(ptr(x, F32, "x"), ptr(k_vectors, F32, "k_vectors"), n_k, ptr(sigma, F32, "sigma"), N,
                                    ptr(atom_types, I64, "atom_types"), ptr(emb_weight, F32, "emb_weight"),
                                    ptr(emb_bias, F32, "emb_bias"), F, H, n_nodes, ptr(z, F32, "z"), ptr(h, F32, "h"),
                                    ptr(w2, F32, "second_weight"), ptr(b2, F32, "second_bias"), H2, ptr(h2, F32, "second_out"),
                                    stream_handle())
    check(rc, "mdx_egnn_node_inputs")
    return (z, h) if second is None else (z, h, h2)


def egnn_scores(z, x_hat, k_vectors):
    """S^alpha = z . Gamma^alpha . x_hat per node -> [n_nodes, 3]  (mdx_egnn_scores)."""
    n_nodes, n_k = z.shape[0], k_vectors.shape[0]
    assert z.shape == x_hat.shape == (n_nodes, 2 * n_k)
    out = torch.empty(n_nodes, 3, dtype=F32, device=z.device)
    rc = lib().mdx_egnn_scores(ptr(z, F32, "z"), ptr(x_hat, F32, "x_hat"), ptr(k_vectors, F32, "k_vectors"), n_k, n_nodes,
                               ptr(out, F32, "scores"), stream_handle())
    check(rc, "mdx_egnn_scores")
    return out


def egnn_outputs(z, x_hat, k_vectors, h, class_weight, class_bias, mask_class: int, n_zero: int):
    """(scores [n_nodes,3], logits [n_nodes,C] with the MASK logit at -inf, zeros [n_zero]) in one launch (mdx_egnn_outputs)."""
    n_nodes, n_k = z.shape[0], k_vectors.shape[0]
    C, H = class_weight.shape
    assert z.shape == x_hat.shape == (n_nodes, 2 * n_k) and h.shape == (n_nodes, H) and class_bias.shape == (C,)
    scores = torch.empty(n_nodes, 3, dtype=F32, device=z.device)
    logits = torch.empty(n_nodes, C, dtype=F32, device=z.device)
    zeros = torch.empty(int(n_zero), dtype=F32, device=z.device)
    rc = lib().mdx_egnn_outputs(ptr(z, F32, "z"), ptr(x_hat, F32, "x_hat"), ptr(k_vectors, F32, "k_vectors"), n_k,
                                ptr(h, F32, "h"), ptr(class_weight, F32, "class_weight"), ptr(class_bias, F32, "class_bias"),
                                H, C, int(mask_class), n_nodes, ptr(scores, F32, "scores"), ptr(logits, F32, "logits"),
                                ptr(zeros, F32, "zeros") if n_zero else None, int(n_zero), stream_handle())
    check(rc, "mdx_egnn_outputs")
    return scores, logits, zeros


def egnn_coord_aggregate(edge_scalar, coord, edges, offsets, degree, mean: bool, flags: int = 0) -> torch.Tensor:
    """coord + segment sum/mean of (coord_i - coord_dst) * edge_scalar over each node's sorted edges (flags: coord_flags())."""
    out = torch.empty_like(coord)
    rc = lib().mdx_egnn_coord_aggregate(ptr(edge_scalar, F32, "edge_scalar"), ptr(coord, F32, "coord"), coord.shape[1],
                                        ptr(edges, I64, "edges"), ptr(offsets, I64, "offsets"), ptr(degree, I64, "degree"),
                                        coord.shape[0], int(bool(mean)), int(flags), ptr(out, F32, "coord_out"),
                                        stream_handle())
    check(rc, "mdx_egnn_coord_aggregate")
    return out


# ----------------------------------------------------------------------------------------------------------------
# RNG fills / probes
# ----------------------------------------------------------------------------------------------------------------
RNG_UNIFORM, RNG_NORMAL, RNG_GUMBEL = 0, 1, 2


def rng_fill(kind: int, seed: int, call: int, draw: int, tag: int, n_items: int, width: int, device) -> torch.Tensor:
    out = torch.empty(n_items, width, dtype=F32, device=device)
    rc = lib().mdx_rng_fill(kind, int(seed) & 0xFFFFFFFFFFFFFFFF, int(call), int(draw), int(tag), n_items, width,
                            ptr(out, F32, "out"), stream_handle())
    check(rc, "mdx_rng_fill")
    return out


def math_probe(fn: int, x: torch.Tensor) -> torch.Tensor:
    y = torch.empty_like(x)
    check(lib().mdx_math_probe(fn, ptr(x, F32, "x"), x.numel(), ptr(y, F32, "y"), stream_handle()), "mdx_math_probe")
    return y
