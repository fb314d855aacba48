"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (container-only).

Run here (the reference never travels to the GPU box; only the .npz files this script writes do):

    PYTHONPATH=/root/reference/src python tests/golden/make_golden.py

What is stored is DATA ONLY: inputs, the noise draws consumed, and the outputs the reference's own
functions returned for them. Reference symbols exercised (file:line under
/root/reference/src/diffusion_for_multi_scale_molecular_dynamics/):

  schedule_*.npz      NoiseScheduler.get_all_sampling_parameters      noise_schedulers/noise_scheduler.py:112-378
  p1_coordinates.npz  LangevinGenerator._relative_coordinates_update  generators/langevin_generator.py:155-201
                      map_relative_coordinates_to_unit_cell            utils/basis_transformations.py:95-119
  p2_atom_types.npz   LangevinGenerator._atom_types_update             generators/langevin_generator.py:247-439
  p3_lattice.npz      LangevinGenerator._lattice_parameters_update     generators/langevin_generator.py:441-490
  noisers.npz         RelativeCoordinatesNoiser / AtomTypesNoiser      noisers/*.py
  neighbors.npz       get_periodic_adjacency_information, get_edges_with_radial_cutoff
                                                                      utils/neighbors.py:36-224, models/egnn_utils.py:107-144
  traj_*.npz          LangevinGenerator.sample / ConstrainedLangevinGenerator.sample with record_samples=True
                                                                      generators/*.py

Third-party packages the reference imports but this image lacks (pykeops, e3nn, torch_geometric) are stubbed
below *for this script only*; the KeOps LazyTensor stand-in is a dense torch evaluation of the same symbolic
expression. torch.rand / torch.randn are wrapped so every draw the reference consumes is
recorded in order: torch's CPU randn stream differs in the last bits between AVX2 and AVX512 hosts, so the
fixtures carry the draws themselves rather than a seed.
"""
import itertools
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("MDX_GOLDEN_OUT", HERE)      # (tests/test_golden_reproducible.py regenerates into a scratch directory)

# --------------------------------------------------------------------------------------------------------------
# container-only stubs for absent third-party modules
# --------------------------------------------------------------------------------------------------------------
if not hasattr(np, "NaN"):
    np.NaN = np.nan


class _Lazy:
    """Dense stand-in for pykeops.torch.LazyTensor (only what utils/neighbors.py:155-186 uses)."""

    def __init__(self, t):
        self.t = t

    def __sub__(self, o):
        return _Lazy(self.t - o.t)

    def __pow__(self, p):
        return _Lazy(self.t ** p)

    def sum(self, dim):
        return _Lazy(self.t.sum(dim=dim))

    def __le__(self, v):
        return _Lazy((self.t <= v).to(torch.float32))

    def sum_reduction(self, dim):
        return self.t.sum(dim=dim, keepdim=True).transpose(dim, -1).squeeze(dim) if False else self.t.sum(dim=dim).unsqueeze(-1)

    def Kmin_argKmin(self, K, dim):
        v, i = torch.topk(self.t, K, dim=dim, largest=False)
        return v, i


def _install_stubs():
    pk = types.ModuleType("pykeops")
    pkt = types.ModuleType("pykeops.torch")
    pkt.LazyTensor = _Lazy
    pk.torch = pkt
    sys.modules["pykeops"] = pk
    sys.modules["pykeops.torch"] = pkt
    e3 = types.ModuleType("e3nn")
    o3 = types.ModuleType("e3nn.o3")
    o3.Irreps = object
    e3.o3 = o3
    sys.modules["e3nn"] = e3
    sys.modules["e3nn.o3"] = o3
    tg = types.ModuleType("torch_geometric")
    tgd = types.ModuleType("torch_geometric.data")
    tgd.Data = object
    tg.data = tgd
    sys.modules["torch_geometric"] = tg
    sys.modules["torch_geometric.data"] = tgd


_install_stubs()

from diffusion_for_multi_scale_molecular_dynamics.generators.constrained_langevin_generator import \
    ConstrainedLangevinGenerator  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics.generators.langevin_generator import \
    LangevinGenerator  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics.generators.predictor_corrector_axl_generator import \
    PredictorCorrectorSamplingParameters  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics.generators.sampling_constraint import \
    SamplingConstraint  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics.models.egnn_utils import \
    get_edges_with_radial_cutoff  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics.models.score_networks.egnn_score_network import (  # noqa: E402
    EGNNScoreNetwork, EGNNScoreNetworkParameters)
from diffusion_for_multi_scale_molecular_dynamics.models.score_networks.mlp_score_network import (  # noqa: E402
    MLPScoreNetwork, MLPScoreNetworkParameters)
from diffusion_for_multi_scale_molecular_dynamics.models.score_networks.score_network import (  # noqa: E402
    ScoreNetwork, ScoreNetworkParameters)
from diffusion_for_multi_scale_molecular_dynamics.namespace import (  # noqa: E402
    AXL, CARTESIAN_FORCES, NOISE, NOISY_AXL_COMPOSITION, TIME)
from diffusion_for_multi_scale_molecular_dynamics.noise_schedulers.noise_parameters import \
    NoiseParameters  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics.noise_schedulers.noise_scheduler import \
    NoiseScheduler  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics.noisers.atom_types_noiser import \
    AtomTypesNoiser  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics.noisers.relative_coordinates_noiser import \
    RelativeCoordinatesNoiser  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics.sampling.diffusion_sampling import \
    create_batch_of_samples  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics.utils.d3pm_utils import \
    class_index_to_onehot  # noqa: E402
from diffusion_for_multi_scale_molecular_dynamics.utils.neighbors import \
    get_periodic_adjacency_information  # noqa: E402


def _np(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    arrays["torch_version"] = np.array(torch.__version__)
    arrays["cpu_capability"] = np.array(torch.backends.cpu.get_cpu_capability())
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


# --------------------------------------------------------------------------------------------------------------
# RNG recorder: wraps torch.rand / torch.randn for the duration of a reference call
# --------------------------------------------------------------------------------------------------------------
class DrawRecorder:
    def __init__(self):
        self.kinds = []   # 0 = rand, 1 = randn
        self.shapes = []
        self.values = []

    def __enter__(self):
        self._rand, self._randn = torch.rand, torch.randn
        rec = self

        def rand(*size, **kw):
            out = rec._rand(*size, **kw)
            rec.kinds.append(0)
            rec.shapes.append(tuple(out.shape))
            rec.values.append(_np(out).ravel().copy())
            return out

        def randn(*size, **kw):
            out = rec._randn(*size, **kw)
            rec.kinds.append(1)
            rec.shapes.append(tuple(out.shape))
            rec.values.append(_np(out).ravel().copy())
            return out

        torch.rand, torch.randn = rand, randn
        return self

    def __exit__(self, *a):
        torch.rand, torch.randn = self._rand, self._randn

    def pack(self):
        shapes = np.full((len(self.shapes), 4), -1, dtype=np.int64)
        for i, s in enumerate(self.shapes):
            shapes[i, : len(s)] = s
        offsets = np.cumsum([0] + [v.size for v in self.values]).astype(np.int64)
        flat = np.concatenate(self.values).astype(np.float32) if self.values else np.zeros(0, np.float32)
        return dict(draw_kinds=np.array(self.kinds, dtype=np.int64), draw_shapes=shapes,
                    draw_offsets=offsets, draw_values=flat)


# --------------------------------------------------------------------------------------------------------------
# S1: schedule tables
# --------------------------------------------------------------------------------------------------------------
def golden_schedules():
    cases = {
        "T3_default": (dict(total_time_steps=3), 2),
        "T10_default": (dict(total_time_steps=10), 3),
        "T17_default": (dict(total_time_steps=17), 5),
        "c1_T100_exp": (dict(total_time_steps=100, sigma_min=1e-4, sigma_max=0.25, schedule_type="exponential"), 2),
        "c2_T1000_exp": (dict(total_time_steps=1000, sigma_min=1e-4, sigma_max=0.25, schedule_type="exponential"), 2),
        "c3_T1000_lin": (dict(total_time_steps=1000, sigma_min=1e-4, sigma_max=0.2, schedule_type="linear",
                              corrector_step_epsilon=2.5e-8), 2),
        "c4_T1000_lin": (dict(total_time_steps=1000, sigma_min=1e-4, sigma_max=0.2, schedule_type="linear",
                              corrector_step_epsilon=2.5e-8), 3),
        "c5_T2000_lin": (dict(total_time_steps=2000, sigma_min=1e-4, sigma_max=0.2, schedule_type="linear",
                              corrector_step_epsilon=2.5e-8), 2),
        "test_T10": (dict(total_time_steps=10, time_delta=0.1, sigma_min=0.15, corrector_step_epsilon=0.25), 5),
    }
    out = {}
    names = []
    for name, (kw, num_classes) in cases.items():
        p = NoiseParameters(**kw)
        noise, ld = NoiseScheduler(p, num_classes=num_classes).get_all_sampling_parameters()
        names.append(name)
        out[f"{name}/params"] = np.array([p.total_time_steps, 0 if p.schedule_type == "exponential" else 1,
                                          p.time_delta, p.sigma_min, p.sigma_max, p.corrector_step_epsilon,
                                          num_classes], dtype=np.float64)
        for k in ["time", "sigma", "sigma_squared", "g", "g_squared", "beta", "alpha_bar", "q_matrix",
                  "q_bar_matrix", "q_bar_tm1_matrix"]:
            out[f"{name}/{k}"] = _np(getattr(noise, k))
        out[f"{name}/epsilon"] = _np(ld.epsilon)
        out[f"{name}/sqrt_2_epsilon"] = _np(ld.sqrt_2_epsilon)
    out["names"] = np.array(names)
    save("schedules.npz", **out)


# --------------------------------------------------------------------------------------------------------------
# helpers to build a generator whose noise draws are pinned
# --------------------------------------------------------------------------------------------------------------
class FakeAXLNetwork(ScoreNetwork):
    """Same behaviour as the reference tests' fake network (tests/generators/conftest.py:14-26): echo the input."""

    def _forward_unchecked(self, batch, conditional=False):
        return AXL(A=class_index_to_onehot(batch[NOISY_AXL_COMPOSITION].A, num_classes=self.num_atom_types + 1),
                   X=batch[NOISY_AXL_COMPOSITION].X, L=batch[NOISY_AXL_COMPOSITION].L)


def make_generator(T, N, num_atom_types, M=1, greedy=True, one=True, in_corr=False, eps=1e-8, fixed=True,
                   cell=None, d=3, noise_kw=None, net=None, record=False, constraint=None):
    nkw = dict(total_time_steps=T)
    nkw.update(noise_kw or {})
    noise_parameters = NoiseParameters(**nkw)
    skw = dict(number_of_corrector_steps=M, number_of_atoms=N, number_of_samples=1, spatial_dimension=d,
               num_atom_types=num_atom_types, one_atom_type_transition_per_step=one,
               atom_type_greedy_sampling=greedy, atom_type_transition_in_corrector=in_corr, small_epsilon=eps,
               record_samples=record, record_samples_corrector_steps=record)
    if fixed:
        skw.update(use_fixed_lattice_parameters=True, cell_dimensions=cell or [5.43] * d)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sampling_parameters = PredictorCorrectorSamplingParameters(**skw)
    if net is None:
        net = FakeAXLNetwork(ScoreNetworkParameters(architecture="dummy", spatial_dimension=d,
                                                    num_atom_types=num_atom_types))
    if constraint is not None:
        gen = ConstrainedLangevinGenerator(noise_parameters=noise_parameters, sampling_parameters=sampling_parameters,
                                           axl_network=net, sampling_constraints=constraint)
    else:
        gen = LangevinGenerator(noise_parameters=noise_parameters, sampling_parameters=sampling_parameters,
                                axl_network=net)
    return gen, noise_parameters, sampling_parameters


# --------------------------------------------------------------------------------------------------------------
# P1 / P3
# --------------------------------------------------------------------------------------------------------------
def golden_p1_p3():
    g = torch.Generator().manual_seed(101)
    gen, _, _ = make_generator(T=10, N=8, num_atom_types=1)
    out = {}
    B, N, d = 6, 8, 3
    x = torch.rand(B, N, d, generator=g)
    # edge cases of the wrap (tests/utils/test_basis_transformations.py:76-110)
    x[0, 0] = torch.tensor([0.0, 1.0 - 2.0 ** -24, 0.5])
    s = torch.randn(B, N, d, generator=g) * 3.0
    z = torch.randn(B, N, d, generator=g)
    scal = []
    outs = []
    for (w, n, sig) in [(0.01, 0.1, 0.05), (2.5e-5, 7.07e-3, 1e-3), (0.3, 0.55, 0.5), (1e-9, 4.5e-5, 1e-4),
                        (0.0, 0.0, 1.0)]:
        w_, n_, sig_ = torch.tensor(w), torch.tensor(n), torch.tensor(sig)
        xo = gen._relative_coordinates_update(x.clone(), s, sig_, w_, n_, z)
        scal.append([_np(w_), _np(n_), _np(sig_)])
        outs.append(_np(xo))
    out["x"], out["s"], out["z"] = _np(x), _np(s), _np(z)
    out["scalars"] = np.array(scal, dtype=np.float32)
    out["x_out"] = np.stack(outs)
    # pure wrap edge cases
    e = torch.tensor([-1e-8, -1e-12, 1.0, 2.0, -1.0, 1.0 - 2.0 ** -24, -0.25, 1.75, 3.999999, -2.0000002, 0.0,
                      -0.0, 1e-30, -1e-30, 5e-8, -5e-8, -3e-8, 123.456, -123.456], dtype=torch.float32)
    from diffusion_for_multi_scale_molecular_dynamics.utils.basis_transformations import \
        map_relative_coordinates_to_unit_cell
    out["wrap_in"] = _np(e)
    out["wrap_out"] = _np(map_relative_coordinates_to_unit_cell(e.clone()))
    save("p1_coordinates.npz", **out)

    # P3 lattice (not fixed)
    gen, _, _ = make_generator(T=10, N=8, num_atom_types=1, fixed=False)
    lat = torch.randn(B, 6, generator=g) + 5.0
    sl = torch.randn(B, 6, generator=g)
    zl = torch.randn(B, 6, generator=g)
    res = []
    scal = []
    for (w, n, sig) in [(0.01, 0.1, 0.05), (2.5e-5, 7.07e-3, 1e-3)]:
        sig_t = torch.tensor(sig)
        sigma_n = sig_t / (N ** (1 / d))  # langevin_generator.py:567-569
        lo = gen._lattice_parameters_update(lat, sl, sigma_n, torch.tensor(w), torch.tensor(n), zl)
        res.append(_np(lo))
        scal.append([w, n, sig, float(sigma_n)])
    save("p3_lattice.npz", l=_np(lat), s=_np(sl), z=_np(zl), scalars=np.array(scal, dtype=np.float32),
         l_out=np.stack(res), n_atoms=np.array(N))


# --------------------------------------------------------------------------------------------------------------
# P2
# --------------------------------------------------------------------------------------------------------------
def golden_p2():
    g = torch.Generator().manual_seed(202)
    out = {}
    case_names = []
    B, N = 12, 8
    T = 10
    for num_atom_types in (1, 2, 4):
        C = num_atom_types + 1
        for greedy, one in itertools.product((False, True), (False, True)):
            for idx in (0, 4, 9):  # idx = index_i - 1 (0 is the last denoising step)
                gen, _, _ = make_generator(T=T, N=N, num_atom_types=num_atom_types, greedy=greedy, one=one)
                logits = torch.randn(B, N, C, generator=g) * 2.0
                logits[..., -1] = -torch.inf
                a = torch.randint(0, C, (B, N), generator=g)
                a[0] = C - 1  # fully masked sample
                a[1] = C - 1
                a[2, ::2] = C - 1
                a[3] = torch.randint(0, num_atom_types, (N,), generator=g)  # no mask at all
                if idx == T - 1:
                    a[:] = C - 1
                u_g = torch.rand(B, N, C, generator=g)
                gumbel = -torch.log(-torch.log(u_g.clip(min=gen.small_epsilon)))
                u_b = torch.rand(B, N, generator=g)
                gen._draw_gumbel_sample = lambda n, gumbel=gumbel: gumbel.clone()
                gen._draw_binary_sample = lambda n, u_b=u_b: u_b.clone()
                gen.record_atom_type_update = True
                from diffusion_for_multi_scale_molecular_dynamics.utils.sample_trajectory import SampleTrajectory
                gen.sample_trajectory_recorder = SampleTrajectory()
                import einops
                q = einops.repeat(gen.noise.q_matrix[idx], "i j -> b n i j", b=B, n=N)
                qb = einops.repeat(gen.noise.q_bar_matrix[idx], "i j -> b n i j", b=B, n=N)
                qbm = einops.repeat(gen.noise.q_bar_tm1_matrix[idx], "i j -> b n i j", b=B, n=N)
                one_eff = one and idx != 0  # langevin_generator.py:601-604
                a_out = gen._atom_types_update(logits, a, q, qb, qbm, atom_type_greedy_sampling=greedy,
                                               one_atom_type_transition_per_step=one_eff)
                rec = gen.sample_trajectory_recorder._internal_data["atom_type_update"][0]
                name = f"C{C}_g{int(greedy)}_o{int(one)}_i{idx}"
                case_names.append(name)
                out[f"{name}/logits"] = _np(logits)
                out[f"{name}/a"] = _np(a)
                out[f"{name}/gumbel"] = _np(gumbel)
                out[f"{name}/u"] = _np(u_b)
                out[f"{name}/q"] = _np(gen.noise.q_matrix[idx])
                out[f"{name}/qbar"] = _np(gen.noise.q_bar_matrix[idx])
                out[f"{name}/qbar_tm1"] = _np(gen.noise.q_bar_tm1_matrix[idx])
                out[f"{name}/flags"] = np.array([int(greedy), int(one_eff), idx, T], dtype=np.int64)
                out[f"{name}/p"] = _np(rec["one_step_transition_probabilities"])
                out[f"{name}/gumbel_used"] = _np(rec["gumbel_sample"])
                out[f"{name}/a_out"] = _np(a_out)
    out["names"] = np.array(case_names)
    out["small_epsilon"] = np.array(1e-8)
    save("p2_atom_types.npz", **out)


# --------------------------------------------------------------------------------------------------------------
# F1 / F2
# --------------------------------------------------------------------------------------------------------------
def golden_noisers():
    g = torch.Generator().manual_seed(303)
    B, N, d = 5, 8, 3
    out = {}
    orig_gauss = RelativeCoordinatesNoiser.__dict__["_get_gaussian_noise"]
    orig_unif = AtomTypesNoiser.__dict__["_get_uniform_noise"]
    x0 = torch.rand(B, N, d, generator=g)
    sig = torch.rand(B, 1, 1, generator=g).expand(B, N, d).contiguous() * 0.5
    z = torch.randn(B, N, d, generator=g)
    RelativeCoordinatesNoiser._get_gaussian_noise = staticmethod(lambda shape: z.clone())
    xt = RelativeCoordinatesNoiser.get_noisy_relative_coordinates_sample(x0, sig)
    out.update(f1_x0=_np(x0), f1_sigma=_np(sig), f1_z=_np(z), f1_xt=_np(xt))
    names = []
    for C in (2, 3, 5):
        T = 10
        sched = NoiseScheduler(NoiseParameters(total_time_steps=T), num_classes=C)
        noise, _ = sched.get_all_sampling_parameters()
        for idx in (0, 3, 9):
            a0 = torch.randint(0, C - 1, (B, N), generator=g)
            u = torch.rand(B, N, C, generator=g)
            u[0, 0, 0] = 0.0  # un-clipped uniform (quirk 6)
            AtomTypesNoiser._get_uniform_noise = staticmethod(lambda shape, u=u: u.clone())
            import einops
            qb = einops.repeat(noise.q_bar_matrix[idx], "i j -> b n i j", b=B, n=N)
            at = AtomTypesNoiser.get_noisy_atom_types_sample(class_index_to_onehot(a0, C), qb)
            nm = f"f2_C{C}_i{idx}"
            names.append(nm)
            out[f"{nm}/a0"] = _np(a0)
            out[f"{nm}/u"] = _np(u)
            out[f"{nm}/qbar"] = _np(noise.q_bar_matrix[idx])
            out[f"{nm}/at"] = _np(at)
    out["f2_names"] = np.array(names)
    # the operands the training transform hands over (noising_transform.py:140-195): a time index PER STRUCTURE, hence a
    # cumulative transition matrix per atom (F2) and a sigma per element that differs between structures (F1 above, F3 here)
    C, T = 3, 10
    sched = NoiseScheduler(NoiseParameters(total_time_steps=T), num_classes=C)
    noise, _ = sched.get_all_sampling_parameters()
    indices = torch.randint(0, T, (B,), generator=g)
    a0 = torch.randint(0, C, (B, N), generator=g)                  # MASK as a starting class included
    u = torch.rand(B, N, C, generator=g)
    AtomTypesNoiser._get_uniform_noise = staticmethod(lambda shape, u=u: u.clone())
    qb = noise.q_bar_matrix[indices][:, None, :, :].expand(B, N, C, C).contiguous()
    at = AtomTypesNoiser.get_noisy_atom_types_sample(class_index_to_onehot(a0, C), qb)
    out.update(f2b_a0=_np(a0), f2b_u=_np(u), f2b_qbar=_np(qb), f2b_at=_np(at))
    from diffusion_for_multi_scale_molecular_dynamics.noisers.lattice_noiser import LatticeDataParameters, LatticeNoiser
    orig_lattice = LatticeNoiser.__dict__["_get_gaussian_noise"]
    l0 = torch.rand(B, 6, generator=g) * 4.0 + 3.0
    sigmas_n = (torch.rand(B, 1, generator=g) * 0.3).expand(B, 6).contiguous()
    zl = torch.randn(B, 6, generator=g)
    LatticeNoiser._get_gaussian_noise = staticmethod(lambda shape: zl.clone())
    lt = LatticeNoiser(LatticeDataParameters(spatial_dimension=3, use_fixed_lattice_parameters=False)) \
        .get_noisy_lattice_parameters(l0, sigmas_n)
    fixed = LatticeNoiser(LatticeDataParameters(spatial_dimension=3, use_fixed_lattice_parameters=True)) \
        .get_noisy_lattice_parameters(l0, sigmas_n)
    assert torch.equal(fixed, l0)
    out.update(f3_l0=_np(l0), f3_sigmas_n=_np(sigmas_n), f3_z=_np(zl), f3_lt=_np(lt))
    LatticeNoiser._get_gaussian_noise = orig_lattice
    RelativeCoordinatesNoiser._get_gaussian_noise = orig_gauss
    AtomTypesNoiser._get_uniform_noise = orig_unif
    save("noisers.npz", **out)


def golden_d3pm_utils():
    """The five callables of utils/d3pm_utils.py on random operands: one-hot and soft distributions, matrices shared by the
    batch (the sampler's expand of one time index) and matrices that differ per atom."""
    from diffusion_for_multi_scale_molecular_dynamics.utils import d3pm_utils as D
    g = torch.Generator().manual_seed(909)
    B, N, C = 4, 6, 4
    out = {}
    index = torch.randint(0, C, (B, N), generator=g)
    onehot = D.class_index_to_onehot(index, C)
    soft = torch.softmax(torch.randn(B, N, C, generator=g), dim=-1)
    logits = torch.randn(B, N, C, generator=g) * 3.0
    logits[..., -1] = -torch.inf

    def stochastic(*lead):
        m = torch.rand(*lead, C, C, generator=g) + 0.05
        return m / m.sum(dim=-1, keepdim=True)
    q_one, qb_one, qbm_one = stochastic(), stochastic(), stochastic()
    shared = [m.expand(B, N, C, C) for m in (q_one, qb_one, qbm_one)]
    per_atom = [stochastic(B, N) for _ in range(3)]
    out.update(index=_np(index), onehot=_np(onehot), soft=_np(soft), logits=_np(logits), q=_np(q_one), q_bar=_np(qb_one),
               q_bar_tm1=_np(qbm_one), q_atoms=_np(per_atom[0]), q_bar_atoms=_np(per_atom[1]), q_bar_tm1_atoms=_np(per_atom[2]))
    out["q_at_given_a0"] = _np(D.compute_q_at_given_a0(onehot, per_atom[1]))
    out["q_at_given_a0_soft"] = _np(D.compute_q_at_given_a0(soft, shared[1]))
    out["q_at_given_atm1"] = _np(D.compute_q_at_given_atm1(onehot, per_atom[0]))
    out["probability_from_logits"] = _np(D.get_probability_from_logits(logits, 1e-8))
    out["previous_logits_shared"] = _np(D.get_probability_at_previous_time_step(logits, onehot, *shared, small_epsilon=1e-8,
                                                                                probability_at_zeroth_timestep_are_logits=True))
    out["previous_logits_atoms"] = _np(D.get_probability_at_previous_time_step(logits, onehot, *per_atom, small_epsilon=1e-8,
                                                                               probability_at_zeroth_timestep_are_logits=True))
    out["previous_soft_shared"] = _np(D.get_probability_at_previous_time_step(soft, onehot, *shared, small_epsilon=1e-8))
    save("d3pm_utils.npz", **out)


# --------------------------------------------------------------------------------------------------------------
# N1
# --------------------------------------------------------------------------------------------------------------
def golden_neighbors():
    g = torch.Generator().manual_seed(404)
    out = {}
    names = []

    def run(name, X, cell, rc):
        cart = torch.matmul(X, cell)
        info = get_periodic_adjacency_information(cart, cell, rc)
        adj = _np(info.adjacency_matrix)
        eb = _np(info.edge_batch_indices)
        shifts = _np(info.shifts)
        # canonical order for set comparison: (batch, src, dst, shift)
        key = np.lexsort((shifts[:, 2], shifts[:, 1], shifts[:, 0], adj[1], adj[0], eb))
        names.append(name)
        out[f"{name}/X"] = _np(X)
        out[f"{name}/cell"] = _np(cell)
        out[f"{name}/cart"] = _np(cart)
        out[f"{name}/rc"] = np.array(rc, dtype=np.float64)
        out[f"{name}/adj_sorted"] = adj[:, key].astype(np.int32)
        out[f"{name}/edge_batch_sorted"] = eb[key].astype(np.int32)
        out[f"{name}/shifts_sorted"] = shifts[key]
        out[f"{name}/number_of_edges"] = _np(info.number_of_edges)
        edges = get_edges_with_radial_cutoff(X, cell, rc, drop_duplicate_edges=True)
        out[f"{name}/unique_edges"] = _np(edges).astype(np.int32)

    B = 4
    # cubic Si 1x1x1-like, Si 2x2x2 with the EGNN clipped cell (quirk N2), small rc
    run("n8_cubic", torch.rand(B, 8, 3, generator=g), torch.diag(torch.tensor([5.43] * 3)).repeat(B, 1, 1), 2.5)
    run("n64_clip", torch.rand(B, 64, 3, generator=g), torch.diag(torch.tensor([16.5] * 3)).repeat(B, 1, 1), 7.5)
    run("n64_cubic", torch.rand(B, 64, 3, generator=g), torch.diag(torch.tensor([10.86] * 3)).repeat(B, 1, 1), 5.0)
    # slightly triclinic cells as in tests/utils/test_neighbors.py:180-236 (5-10 A, rc in {1.1, 2.2, 3.3})
    for rc in (1.1, 2.2, 3.3):
        diag = 5.0 + 5.0 * torch.rand(B, 3, generator=g)
        cell = torch.diag_embed(diag) + 0.1 * (torch.rand(B, 3, 3, generator=g) - 0.5)
        run(f"n32_tric_rc{rc}", torch.rand(B, 32, 3, generator=g), cell, rc)
    run("n216_clip", torch.rand(2, 216, 3, generator=g), torch.diag(torch.tensor([16.5] * 3)).repeat(2, 1, 1), 7.5)
    # coincident atoms (quirk 3): 0 < d^2 excludes them
    X = torch.rand(2, 8, 3, generator=g)
    X[0, 1] = X[0, 0]
    run("n8_coincident", X, torch.diag(torch.tensor([6.0] * 3)).repeat(2, 1, 1), 2.9)
    out["names"] = np.array(names)
    save("neighbors.npz", **out)


# --------------------------------------------------------------------------------------------------------------
# whole trajectories
# --------------------------------------------------------------------------------------------------------------
def _pack_records(gen, with_corrector=True):
    data = gen.sample_trajectory_recorder._internal_data
    out = {}
    pred = data["predictor_step"]
    out["pred_index"] = np.array([e["time_step_index"] for e in pred], dtype=np.int64)
    for key in ("composition_i", "composition_im1", "model_predictions_i"):
        out[f"pred_{key}_A"] = np.stack([_np(e[key].A) for e in pred])
        out[f"pred_{key}_X"] = np.stack([_np(e[key].X) for e in pred])
        out[f"pred_{key}_L"] = np.stack([_np(e[key].L) for e in pred])
    corr = data.get("corrector_step", [])
    if corr:
        out["corr_index"] = np.array([e["time_step_index"] for e in corr], dtype=np.int64)
        for key in ("composition_i", "corrected_composition_i", "model_predictions_i"):
            out[f"corr_{key}_A"] = np.stack([_np(e[key].A) for e in corr])
            out[f"corr_{key}_X"] = np.stack([_np(e[key].X) for e in corr])
            out[f"corr_{key}_L"] = np.stack([_np(e[key].L) for e in corr])
    return out


def _mlp(N, num_atom_types, seed=1234, hidden=64, d=3):
    torch.manual_seed(seed)
    p = MLPScoreNetworkParameters(number_of_atoms=N, num_atom_types=num_atom_types, spatial_dimension=d,
                                  n_hidden_dimensions=3, hidden_dimensions_size=hidden,
                                  relative_coordinates_embedding_dimensions_size=32,
                                  noise_embedding_dimensions_size=16, time_embedding_dimensions_size=16,
                                  atom_type_embedding_dimensions_size=1,
                                  lattice_parameters_embedding_dimensions_size=1, condition_embedding_size=64)
    return MLPScoreNetwork(p).eval()


def _egnn(num_atom_types, edges, rc, seed=1234, hidden=32, n_layers=2):
    torch.manual_seed(seed)
    p = EGNNScoreNetworkParameters(num_atom_types=num_atom_types, n_layers=n_layers,
                                   coordinate_hidden_dimensions_size=hidden, coordinate_n_hidden_dimensions=2,
                                   message_hidden_dimensions_size=hidden, message_n_hidden_dimensions=2,
                                   node_hidden_dimensions_size=hidden, node_n_hidden_dimensions=2,
                                   edges=edges, radial_cutoff=rc)
    return EGNNScoreNetwork(p).eval()


def _state_dict_np(net):
    return {f"net/{k}": _np(v) for k, v in net.state_dict().items()}


def golden_trajectories():
    # (name, generator kwargs, network factory, B, seed)
    runs = [
        ("traj_fake_c2", dict(T=12, N=8, num_atom_types=1, M=1), None, 5, 11),
        ("traj_fake_c3_m2", dict(T=10, N=8, num_atom_types=2, M=2, noise_kw=dict(schedule_type="linear")), None, 4, 12),
        ("traj_fake_c5_nogreedy", dict(T=10, N=8, num_atom_types=4, M=1, greedy=False, one=False), None, 4, 13),
        ("traj_fake_c5_test", dict(T=10, N=8, num_atom_types=4, M=2, eps=1e-6, in_corr=True,
                                   noise_kw=dict(time_delta=0.1, sigma_min=0.15, corrector_step_epsilon=0.25)),
         None, 5, 14),
        ("traj_fake_free_lattice", dict(T=8, N=8, num_atom_types=1, M=1, fixed=False), None, 4, 15),
        ("traj_mlp_c1", dict(T=20, N=8, num_atom_types=1, M=1,
                             noise_kw=dict(sigma_min=1e-4, sigma_max=0.25)), lambda: _mlp(8, 1), 6, 16),
        ("traj_mlp_c3", dict(T=16, N=8, num_atom_types=2, M=2, cell=[5.5421] * 3,
                             noise_kw=dict(sigma_min=1e-4, sigma_max=0.2, schedule_type="linear",
                                           corrector_step_epsilon=2.5e-8)), lambda: _mlp(8, 2), 4, 17),
        ("traj_egnn_fc", dict(T=6, N=8, num_atom_types=1, M=1, one=False, greedy=False,
                              noise_kw=dict(sigma_min=1e-4, sigma_max=0.2, schedule_type="linear",
                                            corrector_step_epsilon=2.5e-8)),
         lambda: _egnn(1, "fully_connected", None), 3, 18),
        ("traj_egnn_rc", dict(T=4, N=64, num_atom_types=1, M=2, one=False, greedy=False, cell=[10.86] * 3,
                              noise_kw=dict(sigma_min=1e-4, sigma_max=0.2, schedule_type="linear",
                                            corrector_step_epsilon=2.5e-8)),
         lambda: _egnn(1, "radial_cutoff", 7.5), 2, 19),
    ]
    for name, kw, netf, B, seed in runs:
        net = netf() if netf else None
        gen, npar, spar = make_generator(record=True, net=net, **kw)
        torch.manual_seed(seed)
        with torch.no_grad(), DrawRecorder() as rec:
            axl = gen.sample(B, torch.device("cpu"))
        out = dict(final_A=_np(axl.A), final_X=_np(axl.X), final_L=_np(axl.L), batch=np.array(B))
        out.update(rec.pack())
        out.update(_pack_records(gen))
        if net is not None:
            out.update(_state_dict_np(net))
        save(name + ".npz", **out)

    # repaint (ConstrainedLangevinGenerator), K = N/2 pinned atoms, fake and MLP nets
    for name, kw, netf, B, seed, idxs in [
        ("traj_repaint_fake", dict(T=10, N=8, num_atom_types=2, M=1), None, 4, 21, None),
        ("traj_repaint_mlp", dict(T=12, N=8, num_atom_types=1, M=2,
                                  noise_kw=dict(sigma_min=1e-4, sigma_max=0.2, schedule_type="linear",
                                                corrector_step_epsilon=2.5e-8)), lambda: _mlp(8, 1), 3, 22,
         torch.tensor([6, 1, 3, 4])),
    ]:
        g = torch.Generator().manual_seed(seed)
        K = 4
        nat = kw["num_atom_types"]
        constraint = SamplingConstraint(elements=["Si", "Ge"][:nat],
                                        constrained_relative_coordinates=torch.rand(K, 3, generator=g),
                                        constrained_atom_types=torch.randint(0, nat, (K,), generator=g),
                                        constrained_indices=idxs)
        net = netf() if netf else None
        gen, npar, spar = make_generator(record=True, net=net, constraint=constraint, **kw)
        torch.manual_seed(seed)
        with torch.no_grad(), DrawRecorder() as rec:
            axl = gen.sample(B, torch.device("cpu"))
        out = dict(final_A=_np(axl.A), final_X=_np(axl.X), final_L=_np(axl.L), batch=np.array(B),
                   constrained_relative_coordinates=_np(constraint.constrained_relative_coordinates),
                   constrained_atom_types=_np(constraint.constrained_atom_types),
                   constrained_indices=_np(gen.constraint_indices))
        out.update(rec.pack())
        out.update(_pack_records(gen))
        if net is not None:
            out.update(_state_dict_np(net))
        save(name + ".npz", **out)

    # D1: create_batch_of_samples with sub-batches (sampling/diffusion_sampling.py:16-73)
    gen, npar, spar = make_generator(T=6, N=8, num_atom_types=1, M=1)
    spar.number_of_samples = 7
    spar.sample_batchsize = 3
    torch.manual_seed(31)
    with torch.no_grad(), DrawRecorder() as rec:
        batch = create_batch_of_samples(gen, spar, torch.device("cpu"))
    out = dict(cartesian_positions=_np(batch["cartesian_positions"]), A=_np(batch["original_axl"].A),
               X=_np(batch["original_axl"].X), L=_np(batch["original_axl"].L))
    out.update(rec.pack())
    save("batch_of_samples.npz", **out)


# --------------------------------------------------------------------------------------------------------------
# score-network forwards (so the build's PyTorch nets can be checked against the reference's on the same weights)
# --------------------------------------------------------------------------------------------------------------
def golden_networks():
    g = torch.Generator().manual_seed(505)
    for name, net, N, nat, cell in [
        ("net_mlp_c1", _mlp(8, 1), 8, 1, 5.43),
        ("net_mlp_c3", _mlp(8, 2), 8, 2, 5.5421),
        ("net_egnn_fc", _egnn(1, "fully_connected", None), 8, 1, 5.43),
        ("net_egnn_rc", _egnn(2, "radial_cutoff", 7.5), 64, 2, 11.084),
    ]:
        B = 3
        C = nat + 1
        batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.randint(0, C, (B, N), generator=g),
                                            X=torch.rand(B, N, 3, generator=g),
                                            L=torch.tensor([cell, cell, cell, 0, 0, 0.0]).repeat(B, 1)),
                 TIME: torch.rand(B, 1, generator=g), NOISE: torch.rand(B, 1, generator=g) * 0.2,
                 CARTESIAN_FORCES: torch.zeros(B, N, 3)}
        with torch.no_grad():
            o = net(batch, conditional=False)
        out = dict(A=_np(batch[NOISY_AXL_COMPOSITION].A), X=_np(batch[NOISY_AXL_COMPOSITION].X),
                   L=_np(batch[NOISY_AXL_COMPOSITION].L), time=_np(batch[TIME]), noise=_np(batch[NOISE]),
                   out_A=_np(o.A), out_X=_np(o.X), out_L=_np(o.L))
        out.update(_state_dict_np(net))
        save(name + ".npz", **out)


# --------------------------------------------------------------------------------------------------------------
# "next" rows of SURVEY 8(f): adaptive corrector, force-field augmentation, atom-type update recording
# (appended after the first fixtures were committed; run with `python make_golden.py next` to write only these)
# --------------------------------------------------------------------------------------------------------------
def golden_next():
    from diffusion_for_multi_scale_molecular_dynamics.generators.adaptive_corrector import AdaptiveCorrectorGenerator
    from diffusion_for_multi_scale_molecular_dynamics.models.score_networks.force_field_augmented_score_network import (
        ForceFieldAugmentedScoreNetwork, ForceFieldParameters)
    import warnings

    # adaptive corrector (generators/adaptive_corrector.py:17-148)
    for name, kw, netf, B, seed in [
        ("traj_adaptive_fake", dict(T=8, N=8, num_atom_types=2, M=2, noise_kw=dict(corrector_r=0.5)), None, 4, 41),
        ("traj_adaptive_mlp", dict(T=10, N=8, num_atom_types=1, M=1,
                                   noise_kw=dict(sigma_min=1e-3, sigma_max=0.2, schedule_type="linear")),
         lambda: _mlp(8, 1), 5, 42),
    ]:
        net = netf() if netf else None
        gen0, npar, spar = make_generator(record=True, net=net, **kw)
        spar.algorithm = "adaptive_corrector"
        gen = AdaptiveCorrectorGenerator(noise_parameters=npar, sampling_parameters=spar, axl_network=gen0.axl_network)
        torch.manual_seed(seed)
        with torch.no_grad(), DrawRecorder() as rec:
            axl = gen.sample(B, torch.device("cpu"))
        out = dict(final_A=_np(axl.A), final_X=_np(axl.X), final_L=_np(axl.L), batch=np.array(B))
        out.update(rec.pack())
        out.update(_pack_records(gen))
        if net is not None:
            out.update(_state_dict_np(net))
        save(name + ".npz", **out)

    # force-field augmented network (models/score_networks/force_field_augmented_score_network.py:44-236)
    g = torch.Generator().manual_seed(606)
    for name, N, cell, rc, strength in [("ff_n8", 8, 5.43, 1.5, 2.0), ("ff_n32", 32, 8.0, 2.2, 0.7)]:
        B = 4
        base = FakeAXLNetwork(ScoreNetworkParameters(architecture="dummy", spatial_dimension=3, num_atom_types=1))
        ff = ForceFieldAugmentedScoreNetwork(base, ForceFieldParameters(radial_cutoff=rc, strength=strength))
        batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.zeros(B, N, dtype=torch.long), X=torch.rand(B, N, 3, generator=g),
                                            L=torch.tensor([cell, cell * 1.1, cell * 1.2, 0, 0, 0.0]).repeat(B, 1)),
                 TIME: torch.rand(B, 1, generator=g), NOISE: torch.rand(B, 1, generator=g) * 0.2,
                 CARTESIAN_FORCES: torch.zeros(B, N, 3)}
        forces = ff.get_relative_coordinates_pseudo_force(batch)
        out = ff(batch, conditional=False)
        save(name + ".npz", X=_np(batch[NOISY_AXL_COMPOSITION].X), L=_np(batch[NOISY_AXL_COMPOSITION].L),
             rc=np.array(rc), strength=np.array(strength), forces=_np(forces), out_X=_np(out.X))

    # atom-type update recording (langevin_generator.py:325-335)
    gen, npar, spar = make_generator(T=6, N=8, num_atom_types=2, M=1, record=True)
    gen.record_atom_type_update = True
    torch.manual_seed(43)
    with torch.no_grad(), DrawRecorder() as rec:
        axl = gen.sample(3, torch.device("cpu"))
    entries = gen.sample_trajectory_recorder._internal_data["atom_type_update"]
    out = dict(final_A=_np(axl.A), final_X=_np(axl.X), final_L=_np(axl.L), batch=np.array(3))
    out.update(rec.pack())
    for key in ("predicted_logits", "one_step_transition_probabilities", "gumbel_sample", "a_i", "a_im1"):
        out["rec_" + key] = np.stack([_np(e[key]) for e in entries])
    save("traj_record_atom_types.npz", **out)




def _reference_compute_distances_in_batch():
    pm = types.ModuleType("pymatgen")
    pmc = types.ModuleType("pymatgen.core")
    pmc.Lattice = object
    pmc.Structure = object
    pm.core = pmc
    sys.modules.setdefault("pymatgen", pm)
    sys.modules.setdefault("pymatgen.core", pmc)
    from diffusion_for_multi_scale_molecular_dynamics.utils.structure_utils import compute_distances_in_batch
    return compute_distances_in_batch


def golden_distances():
    """utils/structure_utils.py:41-121 (pymatgen is only imported by that module for an unrelated helper: stubbed)."""
    compute_distances_in_batch = _reference_compute_distances_in_batch()
    g = torch.Generator().manual_seed(707)
    out = {}
    for name, B, N, box, rc in (("d8", 3, 8, 5.43, 4.0), ("d64", 2, 64, 10.86, 5.0)):
        X = torch.rand(B, N, 3, generator=g)
        cell = torch.diag(torch.tensor([box, box * 1.05, box * 1.1])).repeat(B, 1, 1)
        cart = torch.matmul(X, cell)
        dist = compute_distances_in_batch(cart, cell, rc)
        out[f"{name}/cart"], out[f"{name}/cell"], out[f"{name}/rc"] = _np(cart), _np(cell), np.array(rc)
        out[f"{name}/distances_sorted"] = np.sort(_np(dist))
    save("distances.npz", **out)


def golden_c1_exact():
    """BASELINE configs[0] at its exact settings: Si 1x1x1 (N = 8, one atom type), the MLP of
    config_diffusion_mlp.yaml:41-53, T = 100, sigma 1e-4 .. 0.25 exponential (:21-24), M = 1, batch 16, on the
    reference's CPU path.  Stored: every draw, the final composition, and the composition after every predictor and
    corrector step (atom types as int8) -- inputs and model predictions are not stored (each step's input is the
    previous step's output)."""
    net = _mlp(8, 1)
    gen, npar, spar = make_generator(record=True, net=net, T=100, N=8, num_atom_types=1, M=1,
                                     noise_kw=dict(sigma_min=1e-4, sigma_max=0.25))
    B = 16
    torch.manual_seed(31)
    with torch.no_grad(), DrawRecorder() as rec:
        axl = gen.sample(B, torch.device("cpu"))
    data = gen.sample_trajectory_recorder._internal_data
    out = dict(final_A=_np(axl.A), final_X=_np(axl.X), final_L=_np(axl.L), batch=np.array(B))
    out.update(rec.pack())
    pred, corr = data["predictor_step"], data["corrector_step"]
    out["pred_index"] = np.array([e["time_step_index"] for e in pred], dtype=np.int64)
    out["corr_index"] = np.array([e["time_step_index"] for e in corr], dtype=np.int64)
    out["start_A"] = _np(pred[0]["composition_i"].A).astype(np.int8)
    out["start_X"] = _np(pred[0]["composition_i"].X)
    out["pred_out_A"] = np.stack([_np(e["composition_im1"].A) for e in pred]).astype(np.int8)
    out["pred_out_X"] = np.stack([_np(e["composition_im1"].X) for e in pred])
    out["corr_out_A"] = np.stack([_np(e["corrected_composition_i"].A) for e in corr]).astype(np.int8)
    out["corr_out_X"] = np.stack([_np(e["corrected_composition_i"].X) for e in corr])
    out.update(_state_dict_np(net))
    save("traj_c1_exact.npz", **out)


def _egnn_c3(num_atom_types=1, scale=1.0):
    """The reference's production EGNN (experiments/.../Si_2x2x2/config_diffusion_egnn.yaml:44-60): 4 graph layers, 256 wide,
    4 hidden layers per MLP, radial cutoff 7.5 -- with every trainable parameter filled from tests/formula_weights.py
    (the fixture then needs no 19 MB state_dict: the tests evaluate the same formula)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from formula_weights import fill_with_formula
    p = EGNNScoreNetworkParameters(num_atom_types=num_atom_types, n_layers=4,
                                   coordinate_hidden_dimensions_size=256, coordinate_n_hidden_dimensions=4,
                                   message_hidden_dimensions_size=256, message_n_hidden_dimensions=4,
                                   node_hidden_dimensions_size=256, node_n_hidden_dimensions=4,
                                   coords_agg="mean", message_agg="mean", attention=False, normalize=False, residual=True,
                                   tanh=False, edges="radial_cutoff", radial_cutoff=7.5)
    return fill_with_formula(EGNNScoreNetwork(p).eval(), scale=scale)


def _reordered_copy(net, seed=77):
    """The SAME function with another binary32 summation order: the hidden units of every MLP of every graph layer permuted
    (rows of a Linear and its bias, columns of the Linear that follows).  Mathematically the identity; in binary32 every inner
    product adds its terms in another order -- the distance between `net` and this copy is what "the reference's output" is
    defined up to by the reference's own arithmetic (what another BLAS, thread count or device changes)."""
    import copy as _copy
    other = _copy.deepcopy(net)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for layer in other.egnn.graph_layers:
            for mlp in (layer.message_mlp, layer.coord_mlp, layer.node_mlp):
                linears = [m for m in mlp if isinstance(m, torch.nn.Linear)]
                # outputs of every Linear that feeds another Linear of the same Sequential
                for a, b in zip(linears[:-1], linears[1:]):
                    perm = torch.randperm(a.out_features, generator=g)
                    a.weight.copy_(a.weight[perm].clone())
                    if a.bias is not None:
                        a.bias.copy_(a.bias[perm].clone())
                    b.weight.copy_(b.weight[:, perm].clone())
    return other


def golden_c3_shape(only=None):
    """The network shape and sampler settings BASELINE configs[2] is quoted on (the benchmarked kernels'
    instantiation), on the reference's CPU path:
      net_egnn_c3.npz          EGNNScoreNetwork forward (models/score_networks/egnn_score_network.py:226-303), B = 8, N = 64
      traj_egnn_c3_top.npz     LangevinGenerator.sample_from_noisy_composition(1000 -> 998)  (generators/langevin_generator.py:536-805)
      traj_egnn_c3_bottom.npz  the same 2 -> 0 (index 0: the corrector's sigma_min special case, :719-725)
    with T = 1000, sigma 1e-4 .. 0.2 linear, corrector_step_epsilon 2.5e-8, M = 2, no greedy / one-transition
    (config_diffusion_egnn.yaml:84-103).  Draws and per-step compositions recorded; weights by formula."""
    c4 = only is not None                 # (the two configs[3] trajectories were added later: their own pass and seed)
    if only is None:
        only = ("traj_egnn_c3_top", "traj_egnn_c3_bottom")
    net = _egnn_c3(1)
    g = torch.Generator().manual_seed(911 if c4 else 909)
    B, N, cell = 8, 64, 10.86
    batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.randint(0, 2, (B, N), generator=g), X=torch.rand(B, N, 3, generator=g),
                                        L=torch.tensor([cell, cell, cell, 0, 0, 0.0]).repeat(B, 1)),
             TIME: torch.rand(B, 1, generator=g), NOISE: torch.rand(B, 1, generator=g) * 0.2,
             CARTESIAN_FORCES: torch.zeros(B, N, 3)}
    with torch.no_grad():
        o = net(batch, conditional=False)
    if not c4:
        save("net_egnn_c3.npz", A=_np(batch[NOISY_AXL_COMPOSITION].A), X=_np(batch[NOISY_AXL_COMPOSITION].X),
             L=_np(batch[NOISY_AXL_COMPOSITION].L), time=_np(batch[TIME]), noise=_np(batch[NOISE]),
             out_A=_np(o.A), out_X=_np(o.X), out_L=_np(o.L))

    kw = dict(T=1000, N=64, num_atom_types=1, M=2, one=False, greedy=False, cell=[10.86] * 3,
              noise_kw=dict(sigma_min=1e-4, sigma_max=0.2, schedule_type="linear", corrector_step_epsilon=2.5e-8))
    # BASELINE configs[3] (SiGe 2x2x2: two atom types, greedy sampling and one transition per step ON --
    # experiments/.../SiGe_2x2x2 config, :97-98; cell 11.084): the same network shape with num_atom_types = 2
    kw4 = dict(kw, num_atom_types=2, one=True, greedy=True, cell=[11.084] * 3)
    net4 = _egnn_c3(2)
    B = 4
    for name, start, end, masked_fraction, run_kw, run_net, run_cell, nat in (
            ("traj_egnn_c3_top", 1000, 998, 1.0, kw, net, 10.86, 1), ("traj_egnn_c3_bottom", 2, 0, 0.1, kw, net, 10.86, 1),
            ("traj_egnn_c4_top", 1000, 998, 1.0, kw4, net4, 11.084, 2), ("traj_egnn_c4_mid", 500, 498, 0.5, kw4, net4, 11.084, 2)):
        # (configs[3]'s LAST indices, 2 -> 0, are in golden_live(): with these scale-1 weights the logits of one structure's atoms
        # agree to 1e-7, and with greedy sampling on a partly unmasked structure the Gumbel term is zeroed -- which atom the
        # one-transition rule picks is then decided by the last bit of the logits, i.e. by rounding noise)
        if name not in only:
            continue
        cell = run_cell
        gen, npar, spar = make_generator(record=True, net=run_net, **run_kw)
        X0 = torch.rand(B, N, 3, generator=g)
        masked = torch.rand(B, N, generator=g) < masked_fraction               # MASK = num_atom_types
        A0 = torch.where(masked, torch.full((B, N), nat), torch.randint(0, nat, (B, N), generator=g))
        L0 = torch.tensor([cell, cell, cell, 0, 0, 0.0]).repeat(B, 1)
        torch.manual_seed(910 + start)
        with torch.no_grad(), DrawRecorder() as rec:
            axl = gen.sample_from_noisy_composition(AXL(A=A0, X=X0, L=L0), start, end)
        out = dict(final_A=_np(axl.A), final_X=_np(axl.X), final_L=_np(axl.L), batch=np.array(B),
                   start_A=_np(A0), start_X=_np(X0), start_L=_np(L0), start_index=np.array(start), end_index=np.array(end))
        out.update(rec.pack())
        out.update(_pack_records(gen))
        save(name + ".npz", **out)


def _fp64_forward(net64, composition, time, sigma):
    """The REFERENCE's module in binary64 on one batch (the graph from the binary32 search, as in the fp32 run: the reference's
    neighbour search builds float32 lattice vectors)."""
    from diffusion_for_multi_scale_molecular_dynamics.models.score_networks import egnn_score_network as _mod
    B = composition.X.shape[0]
    batch64 = {NOISY_AXL_COMPOSITION: AXL(A=composition.A, X=composition.X.double(), L=composition.L.double()),
               TIME: torch.full((B, 1), time, dtype=torch.float64), NOISE: torch.full((B, 1), sigma, dtype=torch.float64),
               CARTESIAN_FORCES: torch.zeros_like(composition.X).double()}
    search = _mod.get_edges_with_radial_cutoff
    _mod.get_edges_with_radial_cutoff = lambda x, cell, *a, **k: search(x.float(), cell.float(), *a, **k)
    try:
        with torch.no_grad():
            return net64(batch64, conditional=False)
    finally:
        _mod.get_edges_with_radial_cutoff = search


def _forward_fixture(name, net, net64, A, X, L, time, noise):
    batch = {NOISY_AXL_COMPOSITION: AXL(A=A, X=X, L=L), TIME: time, NOISE: noise, CARTESIAN_FORCES: torch.zeros_like(X)}
    from diffusion_for_multi_scale_molecular_dynamics.models.score_networks import egnn_score_network as _mod
    with torch.no_grad():
        o = net(batch, conditional=False)
        batch64 = {NOISY_AXL_COMPOSITION: AXL(A=A, X=X.double(), L=L.double()), TIME: time.double(), NOISE: noise.double(),
                   CARTESIAN_FORCES: torch.zeros_like(X).double()}
        search = _mod.get_edges_with_radial_cutoff
        _mod.get_edges_with_radial_cutoff = lambda x, cell, *a, **k: search(x.float(), cell.float(), *a, **k)
        try:
            o64 = net64(batch64, conditional=False)
        finally:
            _mod.get_edges_with_radial_cutoff = search
    with torch.no_grad():
        o_perm = _reordered_copy(net)(batch, conditional=False)
    save(name, A=_np(A), X=_np(X), L=_np(L), time=_np(time), noise=_np(noise), out_A=_np(o.A), out_X=_np(o.X), out_L=_np(o.L),
         out_X_fp64=_np(o64.X), out_A_fp64=_np(o64.A), out_X_reordered=_np(o_perm.X), out_A_reordered=_np(o_perm.A))


def golden_c3_wide():
    """More forwards of the production network (models/score_networks/egnn_score_network.py:226-303), the inputs a sampler really
    meets included -- net_egnn_c3.npz holds eight uniform-random structures at sigma in [0.036, 0.176]:
      net_egnn_c3_wide.npz  32 structures of N = 64 (cell 10.86, one atom type): 8 uniform-random at sigma = 1e-4, 1e-3, 1e-2 and
                            0.2; 16 = the diamond sites of Si 2x2x2 displaced by sigma z at sigma = 1e-4 (x4), 1e-3 (x4),
                            1e-2 (x4), 5e-2 (x4) (what the end of a trajectory looks like), unmasked; 8 random with half the
                            atoms MASKed, sigma uniform in [1e-4, 0.2]
      net_egnn_c4.npz       the two-atom-type network of BASELINE configs[3] (cell 11.084): 8 structures, A in {0, 1, MASK},
                            four uniform-random and four displaced diamond sites
    Both with the module's binary64 output on the same inputs (out_X_fp64)."""
    _c3_wide_fixtures("net_egnn_c3_wide.npz", "net_egnn_c4.npz", 1.0, 1913)


def _c3_wide_fixtures(name_c3, name_c4, scale, seed):
    g = torch.Generator().manual_seed(seed)
    N = 64
    sites = _diamond_sites(2)
    net, net64 = _egnn_c3(1, scale), _egnn_c3(1, scale).double()
    X, A, sig = [], [], []
    for s in (1e-4, 1e-3, 1e-2, 0.2):
        for _ in range(2):
            X.append(torch.rand(N, 3, generator=g))
            A.append(torch.randint(0, 2, (N,), generator=g))
            sig.append(s)
    for s in (1e-4, 1e-3, 1e-2, 5e-2):
        for _ in range(4):
            X.append(torch.remainder(sites + s * torch.randn(N, 3, generator=g), 1.0))
            A.append(torch.zeros(N, dtype=torch.long))
            sig.append(s)
    for _ in range(8):
        X.append(torch.rand(N, 3, generator=g))
        A.append((torch.rand(N, generator=g) < 0.5).long())
        sig.append(float(1e-4 + (0.2 - 1e-4) * torch.rand(1, generator=g)))
    X, A = torch.stack(X), torch.stack(A)
    noise = torch.tensor(sig, dtype=torch.float32).reshape(-1, 1)
    B = X.shape[0]
    L = torch.tensor([10.86, 10.86, 10.86, 0, 0, 0.0]).repeat(B, 1)
    _forward_fixture(name_c3, net, net64, A, X, L, torch.rand(B, 1, generator=g), noise)

    net4, net4_64 = _egnn_c3(2, scale), _egnn_c3(2, scale).double()
    X = torch.stack([torch.rand(N, 3, generator=g) for _ in range(4)] +
                    [torch.remainder(sites + s * torch.randn(N, 3, generator=g), 1.0) for s in (1e-4, 1e-3, 1e-2, 5e-2)])
    A = torch.cat([torch.randint(0, 3, (4, N), generator=g), torch.randint(0, 2, (4, N), generator=g)])
    noise = torch.tensor([0.2, 0.1, 1e-2, 1e-3, 1e-4, 1e-3, 1e-2, 5e-2]).reshape(-1, 1)
    L = torch.tensor([11.084, 11.084, 11.084, 0, 0, 0.0]).repeat(8, 1)
    _forward_fixture(name_c4, net4, net4_64, A, X, L, torch.rand(8, 1, generator=g), noise)


LIVE_SCALE = 2.0


def golden_live():
    """The production network shape with formula weights at LIVE_SCALE x nn.Linear's default range.

    Why: at the default range a 4 x 256 x 4 stack of SiLU layers attenuates its signal -- measured on net_egnn_c3's inputs, a
    0.1 % change of a whole 256 x 256 matrix of the message MLP moves the scores by 3e-6 (rel-L2), below the 1e-5 bar, and the
    logits of the atoms of one structure agree to 1e-7: the scale-1 fixtures hold the graph, the coordinate path and the
    biases to the reference, but barely see the hidden layers (and the one-transition argmax over nearly equal logits is
    decided by rounding noise).  At 2 x the range the same perturbation moves the scores by 1.4e-4 and the logits differ between
    atoms by 1e-3: what a trained network looks like to the arithmetic.  (2.5 x is past the edge: |score| ~ 2.5, chaotic.)
      net_egnn_c3_live.npz, net_egnn_c4_live.npz   the inputs of golden_c3_wide (own seed) through the live networks
      traj_egnn_c3_live.npz       C3 settings, 500 -> 498, half the atoms MASKed
      traj_egnn_c4_live_bottom.npz  C4 settings (greedy + one transition per step), 2 -> 0: the one-transition rule is off in
                                  the LAST predictor step, which asserts that no MASK remains
                                  (generators/langevin_generator.py:601-604,616-620)
    Trajectories: every draw, every step's input / output composition and network output, and the module's binary64 output
    on every step's input."""
    _c3_wide_fixtures("net_egnn_c3_live.npz", "net_egnn_c4_live.npz", LIVE_SCALE, 2913)
    g = torch.Generator().manual_seed(2914)
    B, N = 4, 64
    base = dict(T=1000, N=64, M=2, noise_kw=dict(sigma_min=1e-4, sigma_max=0.2, schedule_type="linear",
                                                  corrector_step_epsilon=2.5e-8))
    for name, nat, start, end, masked_fraction, cell, extra in (
            ("traj_egnn_c3_live", 1, 500, 498, 0.5, 10.86, dict(one=False, greedy=False)),
            ("traj_egnn_c4_live_bottom", 2, 2, 0, 0.1, 11.084, dict(one=True, greedy=True))):
        net, net64 = _egnn_c3(nat, LIVE_SCALE), _egnn_c3(nat, LIVE_SCALE).double()
        gen, npar, spar = make_generator(record=True, net=net, num_atom_types=nat, cell=[cell] * 3, **base, **extra)
        X0 = torch.rand(B, N, 3, generator=g)
        masked = torch.rand(B, N, generator=g) < masked_fraction
        A0 = torch.where(masked, torch.full((B, N), nat), torch.randint(0, nat, (B, N), generator=g))
        L0 = torch.tensor([cell, cell, cell, 0, 0, 0.0]).repeat(B, 1)
        forwards = []
        predictions0 = gen._get_model_predictions

        def get_model_predictions(composition, time, sigma_noise, cartesian_forces):
            out = predictions0(composition, time, sigma_noise, cartesian_forces)
            forwards.append((AXL(A=composition.A.clone(), X=composition.X.clone(), L=composition.L.clone()),
                             float(time), float(sigma_noise)))
            return out

        gen._get_model_predictions = get_model_predictions
        torch.manual_seed(2900 + start)
        with torch.no_grad(), DrawRecorder() as rec:
            axl = gen.sample_from_noisy_composition(AXL(A=A0, X=X0, L=L0), start, end)
        out = dict(final_A=_np(axl.A), final_X=_np(axl.X), final_L=_np(axl.L), batch=np.array(B),
                   start_A=_np(A0), start_X=_np(X0), start_L=_np(L0), start_index=np.array(start), end_index=np.array(end),
                   formula_scale=np.array(LIVE_SCALE))
        out.update(rec.pack())
        out.update(_pack_records(gen))
        M = base["M"]
        order = [("pred" if k % (M + 1) == 0 else "corr") for k in range(len(forwards))]
        for kind in ("pred", "corr"):
            mine = [f for f, o in zip(forwards, order) if o == kind]
            assert len(mine) == len(out[f"{kind}_index"])
            for k, f in enumerate(mine):
                assert np.array_equal(_np(f[0].X), out[f"{kind}_composition_i_X"][k])
            out[f"{kind}_model_predictions_i_X_fp64"] = np.stack([_np(_fp64_forward(net64, *f).X) for f in mine])
        save(name + ".npz", **out)


def _diamond_sites(n_cells):
    """The 8 n^3 sites of the diamond structure in an n x n x n supercell, in relative coordinates (cell-major order)."""
    base = torch.tensor([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                         [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]])
    cells = torch.cartesian_prod(*[torch.arange(n_cells)] * 3).float()
    return ((cells[:, None, :] + base[None]) / n_cells).reshape(-1, 3)


def golden_c5_shape():
    """BASELINE configs[4] at the production network: Si 3x3x3 (N = 216, cell 16.29: experiments/.../Si_3x3x3/
    config_diffusion_egnn.yaml:46-60,93-103), the 4 x 256 x 4 EGNN at rc 7.5 (~85 edges per atom in the 16.5 A clipped graph
    cell), ConstrainedLangevinGenerator (generators/constrained_langevin_generator.py:94-163) with K = 108 diamond sites pinned
    (constrained_indices = arange), T = 2000, sigma 1e-4 .. 0.2 linear, eps 2.5e-8, M = 2, B = 2, formula weights:
      net_egnn_c5.npz          the network forward on two structures
      traj_egnn_c5_top.npz     sample_from_noisy_composition(2000 -> 1998): predictor, repaint (noised known rows), 2 correctors
      traj_egnn_c5_bottom.npz  the same 2 -> 0: the second predictor's repaint takes the i-1 == 0 branch (no noising, :120-123)
    Every draw and every step's input / output composition recorded (the outputs as explicit copies taken when the step
    returns, so the in-place repaint of the predictor's output is what is stored)."""
    net = _egnn_c3(1)
    g = torch.Generator().manual_seed(1216)
    B, N, cell, K = 2, 216, 16.29, 108
    batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.randint(0, 2, (B, N), generator=g), X=torch.rand(B, N, 3, generator=g),
                                        L=torch.tensor([cell, cell, cell, 0, 0, 0.0]).repeat(B, 1)),
             TIME: torch.rand(B, 1, generator=g), NOISE: torch.rand(B, 1, generator=g) * 0.2,
             CARTESIAN_FORCES: torch.zeros(B, N, 3)}
    with torch.no_grad():
        o = net(batch, conditional=False)
        # the REFERENCE's module evaluated in binary64 on the same inputs and weights: at this structure size its binary32
        # output sits 2.3e-5 (rel-L2) from it -- the noise floor any binary32 evaluation of this network shares (the
        # coordinate update x + trans rounds at the magnitude of x, 16 A here, while the score is the small difference)
        net64 = _egnn_c3(1).double()
        b = batch[NOISY_AXL_COMPOSITION]
        batch64 = {NOISY_AXL_COMPOSITION: AXL(A=b.A, X=b.X.double(), L=b.L.double()), TIME: batch[TIME].double(),
                   NOISE: batch[NOISE].double(), CARTESIAN_FORCES: batch[CARTESIAN_FORCES].double()}
        # (the reference's neighbour search is binary32-only -- it builds float32 lattice vectors: the fp64 run takes the edge
        # list from the same binary32 search, i.e. the graph of the fp32 run)
        from diffusion_for_multi_scale_molecular_dynamics.models.score_networks import egnn_score_network as _mod
        search = _mod.get_edges_with_radial_cutoff
        _mod.get_edges_with_radial_cutoff = lambda x, cell, *a, **k: search(x.float(), cell.float(), *a, **k)
        try:
            o64 = net64(batch64, conditional=False)
        finally:
            _mod.get_edges_with_radial_cutoff = search
    save("net_egnn_c5.npz", A=_np(batch[NOISY_AXL_COMPOSITION].A), X=_np(batch[NOISY_AXL_COMPOSITION].X),
         L=_np(batch[NOISY_AXL_COMPOSITION].L), time=_np(batch[TIME]), noise=_np(batch[NOISE]),
         out_A=_np(o.A), out_X=_np(o.X), out_L=_np(o.L), out_X_fp64=_np(o64.X), out_A_fp64=_np(o64.A))

    sites = _diamond_sites(3)[:K].clone()
    kw = dict(T=2000, N=N, num_atom_types=1, M=2, one=False, greedy=False, cell=[cell] * 3,
              noise_kw=dict(sigma_min=1e-4, sigma_max=0.2, schedule_type="linear", corrector_step_epsilon=2.5e-8))
    for name, start, end, masked_fraction, spread in (("traj_egnn_c5_top", 2000, 1998, 1.0, None),
                                                      ("traj_egnn_c5_bottom", 2, 0, 0.1, 2e-4)):
        constraint = SamplingConstraint(elements=["Si"], constrained_relative_coordinates=sites.clone(),
                                        constrained_atom_types=torch.zeros(K, dtype=torch.long))
        gen, npar, spar = make_generator(record=False, net=net, constraint=constraint, **kw)
        assert torch.equal(gen.constraint_indices, torch.arange(K))
        X0 = torch.rand(B, N, 3, generator=g)
        masked = torch.rand(B, N, generator=g) < masked_fraction
        A0 = torch.where(masked, torch.ones(B, N, dtype=torch.long), torch.zeros(B, N, dtype=torch.long))
        if spread is not None:      # near the end of a run the pinned atoms sit at their (slightly noised) sites, unmasked
            X0[:, :K] = torch.remainder(sites[None] + spread * torch.randn(B, K, 3, generator=g), 1.0)
            A0[:, :K] = 0
        L0 = torch.tensor([cell, cell, cell, 0, 0, 0.0]).repeat(B, 1)
        steps = dict(pred=[], corr=[])
        pred0, corr0 = gen.predictor_step, gen.corrector_step

        def copy(axl):
            return AXL(A=axl.A.clone(), X=axl.X.clone(), L=axl.L.clone())

        def predictor_step(composition_i, index_i, cartesian_forces):
            before = copy(composition_i)
            out = pred0(composition_i, index_i, cartesian_forces)
            steps["pred"].append((index_i, before, copy(out)))
            return out

        def corrector_step(composition_i, index_i, cartesian_forces):
            before = copy(composition_i)
            out = corr0(composition_i, index_i, cartesian_forces)
            steps["corr"].append((index_i, before, copy(out)))
            return out

        gen.predictor_step, gen.corrector_step = predictor_step, corrector_step
        # what the network returned in every step (generators/langevin_generator.py:113-153: one call per predictor / corrector
        # step, in order), and the same module in binary64 on the same inputs (the floor any binary32 evaluation shares)
        forwards = []
        predictions0 = gen._get_model_predictions

        def get_model_predictions(composition, time, sigma_noise, cartesian_forces):
            out = predictions0(composition, time, sigma_noise, cartesian_forces)
            forwards.append((copy(composition), float(time), float(sigma_noise), copy(out)))
            return out

        gen._get_model_predictions = get_model_predictions
        torch.manual_seed(1500 + start)
        with torch.no_grad(), DrawRecorder() as rec:
            axl = gen.sample_from_noisy_composition(AXL(A=A0.clone(), X=X0.clone(), L=L0.clone()), start, end)
        out = dict(final_A=_np(axl.A), final_X=_np(axl.X), final_L=_np(axl.L), batch=np.array(B),
                   start_A=_np(A0), start_X=_np(X0), start_L=_np(L0), start_index=np.array(start), end_index=np.array(end),
                   constrained_relative_coordinates=_np(sites), constrained_atom_types=np.zeros(K, dtype=np.int64),
                   constrained_indices=_np(gen.constraint_indices))
        out.update(rec.pack())
        for kind, key_in, key_out in (("pred", "pred_composition_i", "pred_composition_im1"),
                                      ("corr", "corr_composition_i", "corr_corrected_composition_i")):
            out[f"{kind}_index"] = np.array([s[0] for s in steps[kind]], dtype=np.int64)
            for field in ("A", "X", "L"):
                out[f"{key_in}_{field}"] = np.stack([_np(getattr(s[1], field)) for s in steps[kind]])
                out[f"{key_out}_{field}"] = np.stack([_np(getattr(s[2], field)) for s in steps[kind]])
        # the forwards in call order: predictor(i+1), corrector(i) x M per time index -> split like the step records
        M = kw["M"]
        order = [("pred" if k % (M + 1) == 0 else "corr") for k in range(len(forwards))]
        for kind in ("pred", "corr"):
            mine = [f for f, o in zip(forwards, order) if o == kind]
            assert len(mine) == len(steps[kind])
            for (index_i, before, _), (comp, _, _, _) in zip(steps[kind], mine):
                assert torch.equal(comp.X, before.X) and torch.equal(comp.A, before.A)
            for field in ("A", "X", "L"):
                out[f"{kind}_model_predictions_i_{field}"] = np.stack([_np(getattr(f[3], field)) for f in mine])
            out[f"{kind}_time"] = np.array([f[1] for f in mine], dtype=np.float64)
            out[f"{kind}_sigma"] = np.array([f[2] for f in mine], dtype=np.float64)
            out[f"{kind}_model_predictions_i_X_fp64"] = np.stack(
                [_np(_fp64_forward(net64, f[0], f[1], f[2]).X) for f in mine])
        save(name + ".npz", **out)


def golden_egnn_variants():
    """E_GCL options the BASELINE configurations do not use but the module accepts (models/egnn.py:36-66,128-131,157,234-264;
    models/egnn_utils.py:111-140): attention, normalize, tanh, sum aggregations, no residual, drop_duplicate_edges=False.
    Small networks (hidden 32), the reference's forward on B = 3 structures of N = 64 atoms; state_dict stored."""
    g = torch.Generator().manual_seed(808)
    variants = {
        "attention": dict(attention=True),
        "normalize_tanh": dict(normalize=True, tanh=True),
        "sum_noresidual": dict(coords_agg="sum", message_agg="sum", residual=False),
        "all_duplicates_kept": dict(attention=True, normalize=True, tanh=True, drop_duplicate_edges=False),
    }
    out = {"names": np.array(list(variants))}
    B, N, cell = 3, 64, 10.86
    for k, (name, kw) in enumerate(variants.items()):
        torch.manual_seed(900 + k)
        p = EGNNScoreNetworkParameters(num_atom_types=2, n_layers=2, coordinate_hidden_dimensions_size=32,
                                       coordinate_n_hidden_dimensions=2, message_hidden_dimensions_size=32,
                                       message_n_hidden_dimensions=2, node_hidden_dimensions_size=32,
                                       node_n_hidden_dimensions=2, edges="radial_cutoff", radial_cutoff=7.5, **kw)
        net = EGNNScoreNetwork(p).eval()
        batch = {NOISY_AXL_COMPOSITION: AXL(A=torch.randint(0, 3, (B, N), generator=g), X=torch.rand(B, N, 3, generator=g),
                                            L=torch.tensor([cell, cell, cell, 0, 0, 0.0]).repeat(B, 1)),
                 TIME: torch.rand(B, 1, generator=g), NOISE: torch.rand(B, 1, generator=g) * 0.2,
                 CARTESIAN_FORCES: torch.zeros(B, N, 3)}
        with torch.no_grad():
            o = net(batch, conditional=False)
        for key, val in (("A", batch[NOISY_AXL_COMPOSITION].A), ("X", batch[NOISY_AXL_COMPOSITION].X),
                         ("L", batch[NOISY_AXL_COMPOSITION].L), ("time", batch[TIME]), ("noise", batch[NOISE]),
                         ("out_A", o.A), ("out_X", o.X), ("out_L", o.L)):
            out[f"{name}/{key}"] = _np(val)
        for key, val in net.state_dict().items():
            out[f"{name}/net/{key}"] = _np(val)
    save("net_egnn_variants.npz", **out)


def golden_low_dimensions():
    """The reference's neighbour search and EGNN in ONE and TWO dimensions (utils/neighbors.py:36-224 takes spatial_dimension in
    {1, 2, 3}; the reference's own tests run both there: tests/utils/test_neighbors.py:239-260,
    tests/models/score_network/test_score_network_general_tests.py:335-371):
      low_dimensions.npz   d1 / d2: get_periodic_adjacency_information on random positions in a 1-D cell and in slightly sheared
                           2-D cells (adjacency, shifts, edge counts, canonically sorted), get_edges_with_radial_cutoff's unique
                           edge list, the smallest cell-crossing distance (for the cutoff-too-large check), and
                           compute_distances_in_batch's bag of distances (sorted);
                           egnn_d1 / egnn_d2: EGNNScoreNetwork (hidden 32, 2 layers, radial cutoff 3.0, formula weights at scale
                           1.5) forward on 4 structures of 6 / 12 atoms in cells of 7 - 10."""
    sys.path.insert(0, os.path.dirname(HERE))
    from formula_weights import fill_with_formula
    from diffusion_for_multi_scale_molecular_dynamics.utils.neighbors import _get_shortest_distance_that_crosses_unit_cell
    compute_distances_in_batch = _reference_compute_distances_in_batch()
    g = torch.Generator().manual_seed(1212)
    out = {}
    B = 4
    for d, N, rc in ((1, 6, 2.5), (2, 12, 3.0)):
        name = f"d{d}"
        X = torch.rand(B, N, d, generator=g)
        cell = torch.diag_embed(7.0 + 3.0 * torch.rand(B, d, generator=g)) + (0.3 * (torch.rand(B, d, d, generator=g) - 0.5) if d > 1 else 0.0)
        cart = torch.matmul(X, cell)
        info = get_periodic_adjacency_information(cart, cell, rc, spatial_dimension=d)
        adj, eb, shifts = _np(info.adjacency_matrix), _np(info.edge_batch_indices), _np(info.shifts)
        key = np.lexsort(tuple(shifts[:, k] for k in reversed(range(d))) + (adj[1], adj[0], eb))
        out[f"{name}/X"], out[f"{name}/cell"], out[f"{name}/cart"], out[f"{name}/rc"] = _np(X), _np(cell), _np(cart), np.array(rc)
        out[f"{name}/adj_sorted"] = adj[:, key].astype(np.int32)
        out[f"{name}/edge_batch_sorted"] = eb[key].astype(np.int32)
        out[f"{name}/shifts_sorted"] = shifts[key]
        out[f"{name}/number_of_edges"] = _np(info.number_of_edges)
        out[f"{name}/unique_edges"] = _np(get_edges_with_radial_cutoff(X, cell, rc, drop_duplicate_edges=True,
                                                                       spatial_dimension=d)).astype(np.int32)
        out[f"{name}/shortest_crossing"] = _np(_get_shortest_distance_that_crosses_unit_cell(cell, spatial_dimension=d))
        out[f"{name}/distances_sorted"] = np.sort(_np(compute_distances_in_batch(cart, cell, rc)))      # structure_utils.py:41-121
        # the network on the same kind of structures (orthogonal cell: lattice parameters = the lengths, angles zero)
        p = EGNNScoreNetworkParameters(spatial_dimension=d, num_atom_types=1, n_layers=2, coordinate_hidden_dimensions_size=32,
                                       coordinate_n_hidden_dimensions=2, message_hidden_dimensions_size=32,
                                       message_n_hidden_dimensions=2, node_hidden_dimensions_size=32, node_n_hidden_dimensions=2,
                                       edges="radial_cutoff", radial_cutoff=3.0)
        net = fill_with_formula(EGNNScoreNetwork(p).eval(), scale=1.5)
        lengths = 7.0 + 3.0 * torch.rand(B, d, generator=g)
        L = torch.cat([lengths, torch.zeros(B, d * (d + 1) // 2 - d)], dim=1)
        Xn, A = torch.rand(B, N, d, generator=g), torch.randint(0, 2, (B, N), generator=g)
        batch = {NOISY_AXL_COMPOSITION: AXL(A=A, X=Xn, L=L), TIME: torch.rand(B, 1, generator=g),
                 NOISE: torch.rand(B, 1, generator=g) * 0.2, CARTESIAN_FORCES: torch.zeros_like(Xn)}
        with torch.no_grad():
            o = net(batch, conditional=False)
        for key_, val in (("A", A), ("X", Xn), ("L", L), ("time", batch[TIME]), ("noise", batch[NOISE]), ("out_A", o.A), ("out_X", o.X)):
            out[f"egnn_{name}/{key_}"] = _np(val)
    save("low_dimensions.npz", **out)


def golden_egnn_options_wide():
    """E_GCL's options at the widths the hand-written kernels are instantiated for beyond 32 (formula weights, so no state_dict is
    stored; tests/formula_weights.py):
      template_1d     the reference's shipped configuration with an option switched on -- configuration_templates/
                      diffusion_config_files/config_diffusion_egnn_2_atoms_in_1D.yaml:52-67: spatial_dimension 1, two atoms,
                      4 graph layers x 128 wide x 4 hidden layers, normalize=True, fully connected -- on 16 structures
      attention_256   attention + tanh at the production width: 2 graph layers x 256 wide x 2 hidden layers, radial cutoff 7.5,
                      two atom types, N = 64, 3 structures (formula scale 2: the gate's logit then varies between edges)
      normalize_128   normalize + attention, 2 x 128 x 3, sum aggregations, N = 64, 3 structures
      default_widths, unequal_48_96   narrow / unequal message and coordinate widths (see below)
    Each with the module's binary64 output on the same inputs."""
    sys.path.insert(0, os.path.dirname(HERE))
    from formula_weights import fill_with_formula
    g = torch.Generator().manual_seed(4242)
    out = {}

    def record(name, p, scale, A, X, L, sigma):
        net = fill_with_formula(EGNNScoreNetwork(p).eval(), scale=scale)
        net64 = fill_with_formula(EGNNScoreNetwork(p).eval(), scale=scale).double()
        B = X.shape[0]
        batch = {NOISY_AXL_COMPOSITION: AXL(A=A, X=X, L=L), TIME: torch.rand(B, 1, generator=g), NOISE: sigma,
                 CARTESIAN_FORCES: torch.zeros_like(X)}
        with torch.no_grad():
            o = net(batch, conditional=False)
            if p.edges == "radial_cutoff":
                o64 = _fp64_forward_batch(net64, batch)
            else:
                o64 = net64({NOISY_AXL_COMPOSITION: AXL(A=A, X=X.double(), L=L.double()), TIME: batch[TIME].double(),
                             NOISE: sigma.double(), CARTESIAN_FORCES: torch.zeros_like(X).double()}, conditional=False)
        for key, val in (("A", A), ("X", X), ("L", L), ("time", batch[TIME]), ("noise", sigma), ("out_A", o.A), ("out_X", o.X),
                         ("out_X_fp64", o64.X), ("out_A_fp64", o64.A)):
            out[f"{name}/{key}"] = _np(val)
        out[f"{name}/formula_scale"] = np.array(scale)

    B = 16
    p = EGNNScoreNetworkParameters(spatial_dimension=1, num_atom_types=1, n_layers=4, coordinate_hidden_dimensions_size=128,
                                   coordinate_n_hidden_dimensions=4, coords_agg="mean", message_hidden_dimensions_size=128,
                                   message_n_hidden_dimensions=4, node_hidden_dimensions_size=128, node_n_hidden_dimensions=4,
                                   attention=False, normalize=True, residual=True, tanh=False, edges="fully_connected")
    record("template_1d", p, 2.0, torch.randint(0, 2, (B, 2), generator=g), torch.rand(B, 2, 1, generator=g),
           torch.ones(B, 1), torch.rand(B, 1, generator=g) * 0.2)
    B, N, cell = 3, 64, 11.084
    L = torch.tensor([cell, cell, cell, 0, 0, 0.0]).repeat(B, 1)
    p = EGNNScoreNetworkParameters(num_atom_types=2, n_layers=2, coordinate_hidden_dimensions_size=256,
                                   coordinate_n_hidden_dimensions=2, message_hidden_dimensions_size=256,
                                   message_n_hidden_dimensions=2, node_hidden_dimensions_size=256, node_n_hidden_dimensions=2,
                                   attention=True, tanh=True, edges="radial_cutoff", radial_cutoff=7.5)
    record("attention_256", p, 2.0, torch.randint(0, 3, (B, N), generator=g), torch.rand(B, N, 3, generator=g), L,
           torch.rand(B, 1, generator=g) * 0.2)
    p = EGNNScoreNetworkParameters(num_atom_types=2, n_layers=2, coordinate_hidden_dimensions_size=128,
                                   coordinate_n_hidden_dimensions=3, message_hidden_dimensions_size=128,
                                   message_n_hidden_dimensions=3, node_hidden_dimensions_size=128, node_n_hidden_dimensions=3,
                                   attention=True, normalize=True, coords_agg="sum", message_agg="sum",
                                   edges="radial_cutoff", radial_cutoff=7.5)
    record("normalize_128", p, 1.5, torch.randint(0, 3, (B, N), generator=g), torch.rand(B, N, 3, generator=g), L,
           torch.rand(B, 1, generator=g) * 0.2)
    # narrow and unequal widths (run zero-padded on the chain: kernels.EdgeChainPack): the reference's DEFAULT hyper-parameters
    # (egnn_score_network.py:23-45: message 16 x 1, node 32 x 1, coordinate 32 x 1, 4 graph layers), and message 48 / coordinate
    # 96 / node 64 with attention + tanh
    p = EGNNScoreNetworkParameters(num_atom_types=2, edges="radial_cutoff", radial_cutoff=7.5)
    record("default_widths", p, 2.0, torch.randint(0, 3, (B, N), generator=g), torch.rand(B, N, 3, generator=g), L,
           torch.rand(B, 1, generator=g) * 0.2)
    p = EGNNScoreNetworkParameters(num_atom_types=2, n_layers=2, coordinate_hidden_dimensions_size=96,
                                   coordinate_n_hidden_dimensions=2, message_hidden_dimensions_size=48,
                                   message_n_hidden_dimensions=2, node_hidden_dimensions_size=64, node_n_hidden_dimensions=2,
                                   attention=True, tanh=True, edges="radial_cutoff", radial_cutoff=7.5)
    record("unequal_48_96", p, 2.0, torch.randint(0, 3, (B, N), generator=g), torch.rand(B, N, 3, generator=g), L,
           torch.rand(B, 1, generator=g) * 0.2)
    out["names"] = np.array(["template_1d", "attention_256", "normalize_128", "default_widths", "unequal_48_96"])
    save("net_egnn_options_wide.npz", **out)


def _fp64_forward_batch(net64, batch):
    from diffusion_for_multi_scale_molecular_dynamics.models.score_networks import egnn_score_network as _mod
    b = batch[NOISY_AXL_COMPOSITION]
    batch64 = {NOISY_AXL_COMPOSITION: AXL(A=b.A, X=b.X.double(), L=b.L.double()), TIME: batch[TIME].double(),
               NOISE: batch[NOISE].double(), CARTESIAN_FORCES: batch[CARTESIAN_FORCES].double()}
    search = _mod.get_edges_with_radial_cutoff
    _mod.get_edges_with_radial_cutoff = lambda x, cell, *a, **k: search(x.float(), cell.float(), *a, **k)
    try:
        return net64(batch64, conditional=False)
    finally:
        _mod.get_edges_with_radial_cutoff = search


if __name__ == "__main__":
    torch.set_num_threads(1)
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all",):
        golden_schedules()
        golden_p1_p3()
        golden_p2()
        golden_noisers()
        golden_neighbors()
        golden_trajectories()
        golden_networks()
    if which in ("noisers",):
        golden_noisers()
    if which in ("all", "d3pm"):
        golden_d3pm_utils()
    if which in ("all", "next"):
        golden_next()
    if which in ("all", "distances"):
        golden_distances()
    if which in ("all", "c1"):
        golden_c1_exact()
    if which in ("all", "c3"):
        golden_c3_shape()
    if which in ("all", "c4"):
        golden_c3_shape(only=("traj_egnn_c4_top", "traj_egnn_c4_mid"))
    if which in ("all", "c3wide"):
        golden_c3_wide()
    if which in ("all", "live"):
        golden_live()
    if which in ("all", "c5"):
        golden_c5_shape()
    if which in ("all", "variants"):
        golden_egnn_variants()
    if which in ("all", "options_wide"):
        golden_egnn_options_wide()
    if which in ("all", "low_dimensions"):
        golden_low_dimensions()
