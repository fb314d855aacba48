"""ctypes front-end of the CPU oracle (oracle/mdx_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing under
diffusion_for_multi_scale_molecular_dynamics_amd/ does.  Every wrapper takes and returns numpy arrays; the
reference lines each C function restates are cited in mdx_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmdx_oracle.so")

TAG_COORD, TAG_GUMBEL, TAG_LATTICE, TAG_INIT, TAG_REPAINT_X0, TAG_BINARY, TAG_REPAINT_Z, TAG_REPAINT_U, \
    TAG_INIT_LATTICE, TAG_RESAMPLE_Z, TAG_RESAMPLE_U = range(11)


def build(force=False):
    """Compile the oracle with gcc (see Makefile for the flags that are part of the arithmetic contract)."""
    src = os.path.join(_HERE, "mdx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def _declare(L):
    L.mdxo_logf.restype = C.c_float
    L.mdxo_logf.argtypes = [C.c_float]
    L.mdxo_expf.restype = C.c_float
    L.mdxo_expf.argtypes = [C.c_float]
    L.mdxo_log.restype = C.c_double
    L.mdxo_log.argtypes = [C.c_double]
    L.mdxo_exp.restype = C.c_double
    L.mdxo_exp.argtypes = [C.c_double]
    L.mdxo_sincospif.restype = None
    L.mdxo_sincospif.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.mdxo_philox4x32_10.restype = None
    L.mdxo_philox4x32_10.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
    for name in ("mdxo_rng_normal", "mdxo_rng_uniform", "mdxo_rng_gumbel"):
        f = getattr(L, name)
        f.restype = None
        f.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int64, C.c_int, _f32p]
    L.mdxo_noise_schedule.restype = C.c_int
    L.mdxo_noise_schedule.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int] + \
        [_f32p] * 12
    L.mdxo_wrap.restype = None
    L.mdxo_wrap.argtypes = [_f32p, C.c_int64, _f32p]
    L.mdxo_coordinates_update.restype = None
    L.mdxo_coordinates_update.argtypes = [_f32p, _f32p, _f32p, C.c_float, C.c_float, C.c_float, C.c_int64, _f32p]
    L.mdxo_lattice_update.restype = None
    L.mdxo_lattice_update.argtypes = [_f32p, _f32p, _f32p, C.c_float, C.c_float, C.c_float, C.c_int64, _f32p]
    L.mdxo_atom_types_update.restype = C.c_int
    L.mdxo_atom_types_update.argtypes = [_f32p, _i64p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int, C.c_int,
                                         C.c_float, C.c_int, C.c_int, _i64p, C.c_void_p, C.c_void_p]
    L.mdxo_noise_coordinates.restype = None
    L.mdxo_noise_coordinates.argtypes = [_f32p, _f32p, C.c_float, C.c_int64, _f32p]
    L.mdxo_noise_atom_types.restype = None
    L.mdxo_noise_atom_types.argtypes = [_i64p, _f32p, _f32p, C.c_int64, C.c_int, _i64p]
    L.mdxo_shortest_crossing_distance.restype = C.c_float
    L.mdxo_shortest_crossing_distance.argtypes = [_f32p]
    L.mdxo_radius_graph.restype = C.c_int64
    L.mdxo_radius_graph.argtypes = [_f32p, _f32p, C.c_float, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]
    L.mdxo_image_vectors_out.restype = None
    L.mdxo_image_vectors_out.argtypes = [_f32p, _f32p]


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


# ----------------------------------------------------------------------------------------------------------------
# scalar math (vectorised in python for tests)
# ----------------------------------------------------------------------------------------------------------------
def logf(x):
    L = lib()
    return np.array([L.mdxo_logf(float(v)) for v in np.ravel(_f32(x))], dtype=np.float32).reshape(np.shape(x))


def expf(x):
    L = lib()
    return np.array([L.mdxo_expf(float(v)) for v in np.ravel(_f32(x))], dtype=np.float32).reshape(np.shape(x))


def sincospif(v):
    L = lib()
    s, c = C.c_float(), C.c_float()
    out = np.zeros((np.size(v), 2), dtype=np.float32)
    for i, x in enumerate(np.ravel(_f32(v))):
        L.mdxo_sincospif(float(x), C.byref(s), C.byref(c))
        out[i] = (s.value, c.value)
    return out


def philox(c0, c1, c2, c3, k0, k1):
    out = (C.c_uint32 * 4)()
    lib().mdxo_philox4x32_10(c0, c1, c2, c3, k0, k1, out)
    return np.array(list(out), dtype=np.uint32)


def rng_normal(seed, call, draw, tag, n_items, width):
    out = np.empty((n_items, width), dtype=np.float32)
    lib().mdxo_rng_normal(seed, call, draw, tag, n_items, width, out)
    return out


def rng_uniform(seed, call, draw, tag, n_items, width):
    out = np.empty((n_items, width), dtype=np.float32)
    lib().mdxo_rng_uniform(seed, call, draw, tag, n_items, width, out)
    return out


def rng_gumbel(seed, call, draw, tag, n_items, width):
    out = np.empty((n_items, width), dtype=np.float32)
    lib().mdxo_rng_gumbel(seed, call, draw, tag, n_items, width, out)
    return out


# ----------------------------------------------------------------------------------------------------------------
# S1
# ----------------------------------------------------------------------------------------------------------------
SCHEDULE_KEYS = ("time", "sigma", "sigma_squared", "g", "g_squared", "epsilon", "sqrt_2_epsilon", "beta",
                 "alpha_bar", "q_matrix", "q_bar_matrix", "q_bar_tm1_matrix")


def noise_schedule(total_time_steps, schedule_type="exponential", time_delta=1e-5, sigma_min=0.005, sigma_max=0.5,
                   corrector_step_epsilon=2e-5, num_classes=2):
    T, Cn = int(total_time_steps), int(num_classes)
    vec = [np.empty(T, dtype=np.float32) for _ in range(9)]
    mats = [np.empty((T, Cn, Cn), dtype=np.float32) for _ in range(3)]
    st = {"exponential": 0, "linear": 1}[schedule_type]
    rc = lib().mdxo_noise_schedule(T, st, time_delta, sigma_min, sigma_max, corrector_step_epsilon, Cn, *vec, *mats)
    if rc != 0:
        raise ValueError(f"mdxo_noise_schedule failed with status {rc}")
    return dict(zip(SCHEDULE_KEYS, vec + mats))


# ----------------------------------------------------------------------------------------------------------------
# P1 / P2 / P3 / F1 / F2
# ----------------------------------------------------------------------------------------------------------------
def wrap(y):
    y = _f32(y)
    out = np.empty_like(y)
    lib().mdxo_wrap(y.ravel(), y.size, out.reshape(-1))
    return out


def coordinates_update(x, s, z, w, n, sigma):
    x, s, z = _f32(x), _f32(s), _f32(z)
    out = np.empty_like(x)
    lib().mdxo_coordinates_update(x.ravel(), s.ravel(), z.ravel(), np.float32(w), np.float32(n), np.float32(sigma),
                                  x.size, out.reshape(-1))
    return out


def lattice_update(l, s, z, w, n, sigma_n):
    l, s, z = _f32(l), _f32(s), _f32(z)
    out = np.empty_like(l)
    lib().mdxo_lattice_update(l.ravel(), s.ravel(), z.ravel(), np.float32(w), np.float32(n), np.float32(sigma_n),
                              l.size, out.reshape(-1))
    return out


def atom_types_update(logits, a, q, qbar, qbar_tm1, gumbel, u, small_epsilon, greedy, one_transition,
                      return_details=False):
    logits, gumbel = _f32(logits), _f32(gumbel)
    B, N, Cn = logits.shape
    a = _i64(a)
    u = _f32(u) if u is not None else np.zeros((B, N), dtype=np.float32)
    out = np.empty((B, N), dtype=np.int64)
    p = np.empty((B, N, Cn), dtype=np.float32)
    g = np.empty((B, N, Cn), dtype=np.float32)
    rc = lib().mdxo_atom_types_update(logits.ravel(), a.ravel(), _f32(q).ravel(), _f32(qbar).ravel(),
                                      _f32(qbar_tm1).ravel(), gumbel.ravel(), u.ravel(), B, N, Cn,
                                      np.float32(small_epsilon), int(greedy), int(one_transition), out.reshape(-1),
                                      p.ctypes.data, g.ctypes.data)
    if rc != 0:
        raise ValueError(f"mdxo_atom_types_update failed with status {rc}")
    return (out, p, g) if return_details else out


def noise_coordinates(x0, z, sigma):
    x0, z = _f32(x0), _f32(z)
    out = np.empty_like(x0)
    lib().mdxo_noise_coordinates(x0.ravel(), z.ravel(), np.float32(sigma), x0.size, out.reshape(-1))
    return out


def noise_atom_types(a0, qbar, u):
    a0, u = _i64(a0), _f32(u)
    Cn = u.shape[-1]
    out = np.empty(a0.shape, dtype=np.int64)
    lib().mdxo_noise_atom_types(a0.ravel(), _f32(qbar).ravel(), u.ravel(), a0.size, Cn, out.reshape(-1))
    return out


# ----------------------------------------------------------------------------------------------------------------
# N1
# ----------------------------------------------------------------------------------------------------------------
class CutoffTooLarge(ValueError):
    pass


def radius_graph(cart, cell, rc, unique):
    """Returns dict(counts[B,N], src[E], dst[E], image[E] (full mode only))."""
    cart, cell = _f32(cart), _f32(cell)
    B, N, _ = cart.shape
    mode = 1 if unique else 0
    counts = np.empty((B, N), dtype=np.int64)
    E = lib().mdxo_radius_graph(cart.ravel(), cell.ravel(), np.float32(rc), B, N, mode, counts.ctypes.data, None,
                                None, None)
    if E == -2:
        raise CutoffTooLarge("radial cutoff reaches beyond the first shell of periodic images")
    src = np.empty(E, dtype=np.int64)
    dst = np.empty(E, dtype=np.int64)
    image = np.empty(E if mode == 0 else 0, dtype=np.int32)
    lib().mdxo_radius_graph(cart.ravel(), cell.ravel(), np.float32(rc), B, N, mode, counts.ctypes.data,
                            src.ctypes.data, dst.ctypes.data, image.ctypes.data if mode == 0 else None)
    return dict(counts=counts, src=src, dst=dst, image=image)


def image_vectors(cell3x3):
    out = np.empty((27, 3), dtype=np.float32)
    lib().mdxo_image_vectors_out(_f32(cell3x3).ravel(), out.reshape(-1))
    return out


def shortest_crossing_distance(cell3x3):
    return lib().mdxo_shortest_crossing_distance(_f32(cell3x3).ravel())
