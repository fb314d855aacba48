"""Summary statistics of a batch of final structures and Kolmogorov-Smirnov distances between them (test infrastructure).

Whole trajectories cannot be compared bit for bit between implementations (the MLP template's map is
chaotic, the radial cutoff makes the EGNN discontinuous), so the device-Philox fast mode is held to the reference
DISTRIBUTIONALLY: the reference's own measure for "do these samples come from the same distribution" is the two-sample
KS distance (src/.../metrics/kolmogorov_smirnov_metrics.py:7-75, scipy.stats.ks_2samp on a scalar per sample).  The scalars
used here, all functions of the final relative coordinates X [B, N, 3] of an orthorhombic cell:

  nn      every atom's minimum-image distance to its nearest neighbour (relative units)          B N values
  pair    every minimum-image pair distance i < j                                                 B N (N-1) / 2 values
  x, y, z every atom's coordinate along one axis                                                  B N values each
  atom<k><axis>  (optional, networks that are not permutation equivariant) the coordinate of atom k along an axis: B values
  to_pinned  (repaint runs) every free atom's minimum-image distance to the nearest pinned site      B (N - K) values
  pair_same, pair_diff, nn_same  (several atom types) pair distances by equal / different type, nearest atom of the own type

The reference side is stored as QUANTILE TABLES (tests/golden/dist_*.npz, made by tests/golden/make_distributions.py from the
reference's own runs): `table[k]` = the k / (len - 1) quantile of the pooled reference sample.  ks_to_table evaluates
sup |F_sample - F_table| with the table's CDF interpolated linearly between its knots (resolution 1 / (len - 1)).
"""
import numpy as np


def _minimum_image_distances(X):
    d = X[:, :, None, :].astype(np.float64) - X[:, None, :, :].astype(np.float64)
    d -= np.round(d)
    return np.sqrt((d * d).sum(-1))                    # [B, N, N]


def statistics(X, per_atom=False, sites=None, pinned=None, types=None):
    """dict name -> 1-D float64 array of the scalars listed in the module docstring; with `sites` [N, 3] also `disp`: the
    displacement of every coordinate from its site, wrapped to [-1/2, 1/2).  With `pinned` [K, 3] (a repaint run: the first K
    atoms are held at these sites) the scalars are those of the FREE atoms X[:, K:] alone, plus `to_pinned`: every free atom's
    minimum-image distance to the nearest pinned site.  With `types` [B, N] (several atom types) also `pair_same` /
    `pair_diff` (the pair distances between atoms of equal / of different type) and `nn_same` (every atom's distance to the
    nearest atom of its own type): continuous scalars that carry the joint distribution of types and positions."""
    X = np.asarray(X)
    if pinned is not None:
        pinned = np.asarray(pinned, np.float64)
        free = X[:, pinned.shape[0]:]
        out = statistics(free, per_atom=per_atom)
        d = free[:, :, None, :].astype(np.float64) - pinned[None, None]
        d -= np.round(d)
        out["to_pinned"] = np.sqrt((d * d).sum(-1)).min(-1).ravel()
        return out
    B, N, _ = X.shape
    r = _minimum_image_distances(X)
    iu = np.triu_indices(N, k=1)
    out = {"pair": r[:, iu[0], iu[1]].ravel()}
    if types is not None:
        same = np.asarray(types)[:, :, None] == np.asarray(types)[:, None, :]
        pairs, same_pairs = r[:, iu[0], iu[1]], same[:, iu[0], iu[1]]
        out["pair_same"], out["pair_diff"] = pairs[same_pairs], pairs[~same_pairs]
    r[:, np.arange(N), np.arange(N)] = np.inf
    out["nn"] = r.min(-1).ravel()
    if types is not None:
        nn_same = np.where(same, r, np.inf).min(-1).ravel()
        out["nn_same"] = nn_same[np.isfinite(nn_same)]           # (an atom alone of its type in a structure has none)
    for k, axis in enumerate("xyz"):
        out[axis] = X[..., k].astype(np.float64).ravel()
    if sites is not None:
        u = X.astype(np.float64) - np.asarray(sites, np.float64)[None]
        out["disp"] = (u - np.round(u)).ravel()
    if per_atom:
        for n in range(N):
            for k, axis in enumerate("xyz"):
                out[f"atom{n}{axis}"] = X[:, n, k].astype(np.float64)
    return out


def quantile_table(values, knots=2049):
    return np.quantile(np.asarray(values, np.float64), np.linspace(0.0, 1.0, knots)).astype(np.float64)


def ks_two_sample(a, b):
    """sup |F_a - F_b| of two empirical distributions (the statistic of scipy.stats.ks_2samp)."""
    a, b = np.sort(np.asarray(a, np.float64)), np.sort(np.asarray(b, np.float64))
    grid = np.concatenate([a, b])
    fa = np.searchsorted(a, grid, side="right") / a.size
    fb = np.searchsorted(b, grid, side="right") / b.size
    return float(np.abs(fa - fb).max())


def ks_to_table(values, table):
    """sup |F_values - F_table|, F_table = the piecewise-linear CDF through the quantile table's knots."""
    v = np.sort(np.asarray(values, np.float64))
    table = np.asarray(table, np.float64)
    p = np.linspace(0.0, 1.0, table.size)
    # strictly increasing knots for the interpolation (ties in the table: keep the last probability at a repeated value)
    keep = np.concatenate([table[1:] > table[:-1], [True]])
    tq, tp = table[keep], p[keep]
    f_table = np.interp(v, tq, tp, left=0.0, right=1.0)
    hi = np.arange(1, v.size + 1) / v.size
    lo = np.arange(0, v.size) / v.size
    d = max(np.abs(hi - f_table).max(), np.abs(lo - f_table).max())
    # and at the table's knots
    f_sample = np.searchsorted(v, tq, side="right") / v.size
    return float(max(d, np.abs(f_sample - tp).max()))
