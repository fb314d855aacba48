"""CPU restatement of the reference's predictor-corrector sampling loop.  TEST INFRASTRUCTURE ONLY.

Whole-loop counterpart of mdx_oracle.c: numpy state, the C oracle for every per-step formula, and a torch-CPU
score network for the forward pass.  Used by tests/ (parity of the GPU generator), __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.

Follows (paths under /root/reference/src/diffusion_for_multi_scale_molecular_dynamics/):
  generators/predictor_corrector_axl_generator.py:81-161   sample / sample_from_noisy_composition
  generators/langevin_generator.py:536-645, 693-805        predictor_step / corrector_step
  generators/trajectory_initializer.py:101-123             FullRandomTrajectoryInitializer.initialize
  generators/constrained_langevin_generator.py:74-182      repaint
  sampling/diffusion_sampling.py:16-73                     create_batch_of_samples
"""
from collections import namedtuple

import numpy as np

from . import mdx_oracle as O

AXL = namedtuple("AXL", ["A", "X", "L"])
PREDICTOR, CORRECTOR = 0, 1


# ----------------------------------------------------------------------------------------------------------------
# noise sources
# ----------------------------------------------------------------------------------------------------------------
class TorchCpuNoise:
    """Draws from torch's CPU default generator in exactly the order the reference consumes it
    (SURVEY.md section 8a, row RNG), including the draws the reference throws away."""

    reference_order = True

    def rand(self, *shape):
        import torch
        return torch.rand(*shape).numpy()

    def randn(self, *shape):
        import torch
        return torch.randn(*shape).numpy()


class ReplayNoise:
    """Replays the draws recorded in a golden fixture (tests/golden/traj_*.npz)."""

    reference_order = True

    def __init__(self, fixture):
        self.kinds = fixture["draw_kinds"]
        self.shapes = fixture["draw_shapes"]
        self.offsets = fixture["draw_offsets"]
        self.values = fixture["draw_values"]
        self.cursor = 0

    def _next(self, kind, shape):
        i = self.cursor
        assert i < len(self.kinds), "fixture ran out of recorded draws"
        rec_shape = tuple(int(s) for s in self.shapes[i] if s >= 0)
        assert int(self.kinds[i]) == kind and rec_shape == tuple(shape), \
            f"draw {i}: fixture has kind {self.kinds[i]} shape {rec_shape}, sampler asked kind {kind} shape {shape}"
        self.cursor += 1
        return self.values[self.offsets[i]:self.offsets[i + 1]].reshape(shape).copy()

    def rand(self, *shape):
        return self._next(0, shape)

    def randn(self, *shape):
        return self._next(1, shape)

    def exhausted(self):
        return self.cursor == len(self.kinds)


class PhiloxNoise:
    """The device-RNG specification: draws are pure functions of (seed, call, draw, tag, item)."""

    reference_order = False

    def __init__(self, seed, call=0):
        self.seed, self.call = int(seed), int(call)

    def normal(self, draw, tag, n_items, width):
        return O.rng_normal(self.seed, self.call, draw, tag, n_items, width)

    def uniform(self, draw, tag, n_items, width):
        return O.rng_uniform(self.seed, self.call, draw, tag, n_items, width)

    def gumbel(self, draw, tag, n_items, width):
        return O.rng_gumbel(self.seed, self.call, draw, tag, n_items, width)


# ----------------------------------------------------------------------------------------------------------------
# per-step scalars
# ----------------------------------------------------------------------------------------------------------------
def step_scalars(tables, sigma_min, mode, index, number_of_atoms, spatial_dimension):
    """predictor_step scalars (langevin_generator.py:559-569) or corrector_step scalars (:719-733, :678, :749)."""
    f32 = np.float32
    atoms_pow = float(number_of_atoms) ** (1.0 / spatial_dimension)
    if mode == PREDICTOR:
        idx = index - 1
        sigma = tables["sigma"][idx]
        return dict(idx=idx, time=tables["time"][idx], sigma=sigma, w=tables["g_squared"][idx], n=tables["g"][idx],
                    sigma_n=f32(sigma / f32(atoms_pow)))
    if index == 0:
        idx, time, sigma = 0, f32(0.0), f32(sigma_min)
        sigma_n = f32(float(sigma_min) / atoms_pow)
    else:
        idx = index - 1
        time, sigma = tables["time"][idx], tables["sigma"][idx]
        sigma_n = f32(sigma / f32(atoms_pow))
    w = tables["epsilon"][index]
    return dict(idx=idx, time=time, sigma=sigma, w=w, n=np.sqrt(f32(2.0) * w), sigma_n=sigma_n)


# ----------------------------------------------------------------------------------------------------------------
# the generator
# ----------------------------------------------------------------------------------------------------------------
class OracleLangevinGenerator:
    """numpy/C restatement of LangevinGenerator (+ ConstrainedLangevinGenerator when a constraint is given)."""

    def __init__(self, noise_parameters, sampling_parameters, axl_network, constraint=None, noise=None):
        """noise_parameters / sampling_parameters: any objects with the reference's field names.
        axl_network: callable(batch_dict, conditional=False) -> AXL of torch CPU tensors.
        constraint: None or dict(constrained_relative_coordinates [K,d], constrained_atom_types [K],
                                 constrained_indices [K] or None)."""
        npar, spar = noise_parameters, sampling_parameters
        self.T = npar.total_time_steps
        self.M = spar.number_of_corrector_steps
        self.N = spar.number_of_atoms
        self.d = spar.spatial_dimension
        self.C = spar.num_atom_types + 1
        self.nl = self.d * (self.d + 1) // 2
        self.sigma_min = npar.sigma_min
        self.small_epsilon = spar.small_epsilon
        self.greedy = spar.atom_type_greedy_sampling
        self.one = spar.one_atom_type_transition_per_step
        self.in_corrector = spar.atom_type_transition_in_corrector
        self.fixed = spar.use_fixed_lattice_parameters
        self.fixed_lattice_parameters = None
        if self.fixed:
            cell = np.asarray(spar.cell_dimensions, dtype=np.float32)
            lat = np.zeros(self.nl, dtype=np.float32)
            lat[: self.d] = cell if cell.ndim == 1 else np.diag(cell)
            self.fixed_lattice_parameters = lat
        self.tables = O.noise_schedule(self.T, npar.schedule_type, npar.time_delta, npar.sigma_min, npar.sigma_max,
                                       npar.corrector_step_epsilon, self.C)
        self.net = axl_network
        self.noise = noise if noise is not None else TorchCpuNoise()
        self.constraint = constraint
        if constraint is not None:
            K = len(constraint["constrained_atom_types"])
            idx = constraint.get("constrained_indices")
            self.cidx = np.arange(K) if idx is None else np.asarray(idx)
        # build-only RePaint resampling (no reference counterpart; 0 = the reference's loop)
        self.resampling = int(getattr(spar, "repaint_resampling_steps", 0) or 0) if constraint is not None else 0
        self.visit = 0
        self.records = []          # filled when record=True
        self.record = False
        self.mask_left_at_last_step = False

    # -- draws -------------------------------------------------------------------------------------------------
    def _draw_id(self, index, offset):
        return index * (self.M + 1) * (self.resampling + 1) + self.visit * (self.M + 1) + offset

    def _normal_coords(self, B, index, offset, tag=O.TAG_COORD):
        if self.noise.reference_order:
            return self.noise.randn(B, self.N, self.d)
        return self.noise.normal(self._draw_id(index, offset), tag, B * self.N, self.d).reshape(B, self.N, self.d)

    def _normal_lattice(self, B, index, offset, needed):
        if self.noise.reference_order:
            return self.noise.randn(B, self.nl)
        if not needed:
            return None
        return self.noise.normal(self._draw_id(index, offset), O.TAG_LATTICE, B, self.nl)

    def _gumbel(self, B, index, offset):
        if self.noise.reference_order:
            import torch
            u = torch.from_numpy(self.noise.rand(B, self.N, self.C))
            # langevin_generator.py:100-107, evaluated by torch on the CPU exactly as the reference does
            return (-torch.log(-torch.log(u.clip(min=self.small_epsilon)))).numpy()
        return self.noise.gumbel(self._draw_id(index, offset), O.TAG_GUMBEL, B * self.N, self.C).reshape(B, self.N, self.C)

    def _binary(self, B, index, offset):
        if self.noise.reference_order:
            return self.noise.rand(B, self.N)
        return self.noise.uniform(self._draw_id(index, offset), O.TAG_BINARY, B * self.N, 1).reshape(B, self.N)

    # -- network -----------------------------------------------------------------------------------------------
    def _predict(self, comp, time, sigma):
        import torch
        B = comp.X.shape[0]
        batch = {
            "noisy_axl": AXL(A=torch.from_numpy(comp.A.copy()), X=torch.from_numpy(comp.X.copy()),
                             L=torch.from_numpy(comp.L.copy())),
            "time": torch.full((B, 1), float(time), dtype=torch.float32),
            "noise_parameter": torch.full((B, 1), float(sigma), dtype=torch.float32),
            "cartesian_forces": torch.zeros(B, self.N, self.d),
        }
        with torch.no_grad():
            out = self.net(batch, conditional=False)
        return AXL(A=out.A.numpy(), X=out.X.numpy(), L=out.L.numpy())

    # -- steps -------------------------------------------------------------------------------------------------
    def _atom_types(self, logits, A, idx, one, index, offset):
        B = A.shape[0]
        gumbel = self._gumbel(B, index, offset)
        u = self._binary(B, index, offset) if self.greedy else None
        t = self.tables
        return O.atom_types_update(logits, A, t["q_matrix"][idx], t["q_bar_matrix"][idx], t["q_bar_tm1_matrix"][idx],
                                   gumbel, u, self.small_epsilon, self.greedy, one)

    def predictor_step(self, comp, index):
        B = comp.X.shape[0]
        sc = step_scalars(self.tables, self.sigma_min, PREDICTOR, index, self.N, self.d)
        pred = self._predict(comp, sc["time"], sc["sigma"])
        last = sc["idx"] == 0
        A = self._atom_types(pred.A, comp.A, sc["idx"], self.one and not last, index, 0)
        if last and (A == self.C - 1).any():
            self.mask_left_at_last_step = True
        z = self._normal_coords(B, index, 0)
        X = O.coordinates_update(comp.X, pred.X, z, sc["w"], sc["n"], sc["sigma"])
        zl = self._normal_lattice(B, index, 0, needed=not self.fixed)
        L = comp.L if self.fixed else O.lattice_update(comp.L, pred.L, zl, sc["w"], sc["n"], sc["sigma_n"])
        out = AXL(A=A, X=X, L=L)
        if self.record:
            self.records.append(("predictor", index, comp, out, pred))
        if self.constraint is not None:
            out = self._repaint(out, index - 1, index)
        return out

    def corrector_step(self, comp, index, m=0):
        B = comp.X.shape[0]
        sc = step_scalars(self.tables, self.sigma_min, CORRECTOR, index, self.N, self.d)
        pred = self._predict(comp, sc["time"], sc["sigma"])
        z = self._normal_coords(B, index, 1 + m)
        X = O.coordinates_update(comp.X, pred.X, z, sc["w"], sc["n"], sc["sigma"])
        if self.noise.reference_order:
            self.noise.randn(B, self.nl)            # langevin_generator.py:761-763: drawn, never used
        if self.fixed:
            L = comp.L
        else:
            zl = self._normal_lattice(B, index, 1 + m, needed=True)   # :480-483, the draw that is used
            L = O.lattice_update(comp.L, pred.L, zl, sc["w"], sc["n"], sc["sigma_n"])
        A = comp.A
        if self.in_corrector:
            A = self._atom_types(pred.A, comp.A, sc["idx"], self.one, index, 1 + m)
        out = AXL(A=A, X=X, L=L)
        if self.record:
            self.records.append(("corrector", index, comp, out, pred))
        return out

    # -- repaint -----------------------------------------------------------------------------------------------
    def _repaint(self, comp, index, draw_index):
        """constrained_langevin_generator.py:136-163 with _noise_composition (:118-134)."""
        B = comp.X.shape[0]
        c = self.constraint
        cx = np.asarray(c["constrained_relative_coordinates"], dtype=np.float32)
        ca = np.asarray(c["constrained_atom_types"], dtype=np.int64)
        X, A = comp.X.copy(), comp.A.copy()
        if self.noise.reference_order:
            self.initialize(B)                       # composition_0_known: only its constrained rows survive
        if index == 0:
            X[:, self.cidx] = cx
            A[:, self.cidx] = ca
            return AXL(A=A, X=X, L=comp.L)
        idx = index - 1
        sigma = self.tables["sigma"][idx]
        qbar = self.tables["q_bar_matrix"][idx]
        if self.noise.reference_order:
            z = self.noise.randn(B, self.N, self.d)
            u = self.noise.rand(B, self.N, self.C)
        else:
            dr = self._draw_id(draw_index, 0)
            z = self.noise.normal(dr, O.TAG_REPAINT_Z, B * self.N, self.d).reshape(B, self.N, self.d)
            u = self.noise.uniform(dr, O.TAG_REPAINT_U, B * self.N, self.C).reshape(B, self.N, self.C)
        x0 = np.broadcast_to(cx, (B,) + cx.shape)
        a0 = np.broadcast_to(ca, (B,) + ca.shape)
        X[:, self.cidx] = O.noise_coordinates(x0, z[:, self.cidx], sigma)
        A[:, self.cidx] = O.noise_atom_types(a0, qbar, np.ascontiguousarray(u[:, self.cidx]))
        return AXL(A=A, X=X, L=comp.L)

    def forward_step(self, comp, index):
        """Resampling: one step of the forward process, time index i -> i+1, on the whole composition
        (X += g z wrapped -- the F1 arithmetic with g in place of sigma; A ~ one-step kernel Q -- the F2 arithmetic
        with Q in place of Q-bar), using the table row the predictor step i+1 -> i reads."""
        B = comp.X.shape[0]
        g = self.tables["g"][index]
        q = self.tables["q_matrix"][index]
        if self.noise.reference_order:
            z = self.noise.randn(B, self.N, self.d)
            u = self.noise.rand(B, self.N, self.C)
        else:
            dr = self._draw_id(index, 0)
            z = self.noise.normal(dr, O.TAG_RESAMPLE_Z, B * self.N, self.d).reshape(B, self.N, self.d)
            u = self.noise.uniform(dr, O.TAG_RESAMPLE_U, B * self.N, self.C).reshape(B, self.N, self.C)
        X = O.noise_coordinates(comp.X, z, g)
        A = O.noise_atom_types(comp.A, q, np.ascontiguousarray(u))
        return AXL(A=A, X=X, L=comp.L)

    # -- loop --------------------------------------------------------------------------------------------------
    def initialize(self, B):
        """trajectory_initializer.py:101-123"""
        A = np.full((B, self.N), self.C - 1, dtype=np.int64)
        if self.noise.reference_order:
            X = self.noise.rand(B, self.N, self.d)
        else:
            X = self.noise.uniform(0, O.TAG_INIT, B * self.N, self.d).reshape(B, self.N, self.d)
        if self.fixed:
            L = np.tile(self.fixed_lattice_parameters, (B, 1))
        elif self.noise.reference_order:
            L = self.noise.randn(B, self.nl)
        else:
            L = self.noise.normal(0, O.TAG_INIT_LATTICE, B, self.nl)
        return AXL(A=A, X=np.ascontiguousarray(X, dtype=np.float32), L=np.ascontiguousarray(L, dtype=np.float32))

    def sample_from_noisy_composition(self, comp, starting_step_index, ending_step_index=0):
        for i in range(starting_step_index - 1, max(ending_step_index, 0) - 1, -1):
            visits = 1 + self.resampling if i > 0 else 1
            for self.visit in range(visits):
                comp = self.predictor_step(comp, i + 1)
                for m in range(self.M):
                    comp = self.corrector_step(comp, i, m)
                if self.visit < visits - 1:
                    comp = self.forward_step(comp, i)
            self.visit = 0
        return comp

    def sample(self, number_of_samples):
        comp = self.initialize(number_of_samples)
        comp = self.sample_from_noisy_composition(comp, self.T, 0)
        if self.constraint is not None:              # constrained_langevin_generator.py:179-182
            X, A = comp.X.copy(), comp.A.copy()
            X[:, self.cidx] = np.asarray(self.constraint["constrained_relative_coordinates"], dtype=np.float32)
            A[:, self.cidx] = np.asarray(self.constraint["constrained_atom_types"], dtype=np.int64)
            comp = AXL(A=A, X=X, L=comp.L)
        return comp


def create_batch_of_samples(generator, number_of_samples, sample_batchsize=None):
    """sampling/diffusion_sampling.py:16-73"""
    bs = number_of_samples if sample_batchsize is None else sample_batchsize
    parts = []
    start = 0
    while start < number_of_samples:
        n = min(bs, number_of_samples - start)
        parts.append(generator.sample(n))
        start += n
    A = np.concatenate([p.A for p in parts])
    X = np.concatenate([p.X for p in parts])
    L = np.concatenate([p.L for p in parts]).copy()
    d = X.shape[-1]
    L[..., d:] = 0
    cart = (X * L[:, None, :d]).astype(np.float32)   # X @ diag(L[:d])
    return {"cartesian_positions": cart, "original_axl": AXL(A=A, X=X, L=L)}


class OracleAdaptiveCorrectorGenerator(OracleLangevinGenerator):
    """generators/adaptive_corrector.py:17-148: predictor touches the atom types only; corrector step size
    eps = 2 (r mean|z| / (mean|sigma s| / sigma))^2 from batch means."""

    def __init__(self, noise_parameters, sampling_parameters, axl_network, noise=None):
        super().__init__(noise_parameters, sampling_parameters, axl_network, noise=noise)
        self.corrector_r = noise_parameters.corrector_r

    def predictor_step(self, comp, index):
        B = comp.X.shape[0]
        sc = step_scalars(self.tables, self.sigma_min, PREDICTOR, index, self.N, self.d)
        pred = self._predict(comp, sc["time"], sc["sigma"])
        last = sc["idx"] == 0
        A = self._atom_types(pred.A, comp.A, sc["idx"], self.one and not last, index, 0)
        if self.noise.reference_order:
            self.noise.randn(B, self.N, self.d)          # drawn by the reference, unused
            self.noise.randn(B, self.nl)
        out = AXL(A=A, X=comp.X, L=comp.L)
        if self.record:
            self.records.append(("predictor", index, comp, out, pred))
        return out

    def _eps(self, sigma, score, z, coordinates):
        f32 = np.float32
        flat = score.reshape(score.shape[0], -1) if coordinates else score
        score_norm = f32(np.linalg.norm(flat.astype(np.float32), axis=-1).mean(dtype=np.float32)) / f32(sigma)
        z_norm = f32(np.linalg.norm(z.astype(np.float32), axis=-1).mean(dtype=np.float32))
        ratio = f32(self.corrector_r) * z_norm / max(score_norm, f32(self.small_epsilon))
        return f32(2.0) * f32(ratio) * f32(ratio)

    def corrector_step(self, comp, index, m=0):
        B = comp.X.shape[0]
        sc = step_scalars(self.tables, self.sigma_min, CORRECTOR, index, self.N, self.d)
        pred = self._predict(comp, sc["time"], sc["sigma"])
        z = self._normal_coords(B, index, 1 + m)
        eps = self._eps(sc["sigma"], pred.X, z, True)
        X = O.coordinates_update(comp.X, pred.X, z, eps, np.sqrt(np.float32(2.0) * eps), sc["sigma"])
        L = comp.L
        if self.noise.reference_order:
            z_lat = self.noise.randn(B, self.nl)
        if not self.fixed:
            if self.noise.reference_order:
                z_used = self.noise.randn(B, self.nl)
            else:
                z_lat = self._normal_lattice(B, index, 1 + m, needed=True)
                z_used = z_lat
            eps_l = self._eps(sc["sigma_n"], pred.L, z_lat, False)
            L = O.lattice_update(comp.L, pred.L, z_used, eps_l, np.sqrt(np.float32(2.0) * eps_l), sc["sigma_n"])
        out = AXL(A=comp.A, X=X, L=L)
        if self.record:
            self.records.append(("corrector", index, comp, out, pred))
        return out
