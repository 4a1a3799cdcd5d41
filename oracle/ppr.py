"""Oracle: Personalized PageRank as the reference obtains it from igraph/PRPACK.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: igraph is
not installable here; the semantics below are a restatement of the published
PRPACK algorithm (python_igraph==0.11.8 -> igraph C core 0.10.x ->
vendor/prpack), anchored on the reference's single call site
``src/hipporag/HippoRAG.py:1736-1743``:

    graph.personalized_pagerank(vertices=range(V), damping=damping,
        directed=False, weights='weight', reset=reset_prob,
        implementation='prpack')

Semantics restated:
  * the graph is undirected (``utils/config_utils.py:176``); every igraph edge
    (u, v, w) contributes A[u,v] += w and A[v,u] += w, parallel edges add up
    (``HippoRAG.py:1189-1223`` creates one igraph edge per dict key, so a fact
    pair (s,o)/(o,s) becomes two parallel edges, ``:906-910``); self pairs are
    dropped before they reach igraph (``:1201``);
  * weights are normalised per SOURCE vertex: P[i,j] = A[i,j] / sum_i A[i,j]
    (column-stochastic); vertices without edges are dangling;
  * ``reset`` must be non-negative with a positive sum; it is divided by its
    sum and used both as teleport vector v and as the distribution dangling
    mass is sent to (u == v), so the solution satisfies
        x = alpha P x + (alpha d^T x + 1 - alpha) v ,   sum(x) = 1
    i.e. x = normalise_1((I - alpha P)^-1 v);
  * PRPACK solves this to tolerance 1e-10 (Gauss-Seidel, or dense Gaussian
    elimination below 128 vertices); the exact fixed point is what the oracle
    returns.
"""

from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def build_symmetric_csr(num_vertices: int, src, dst, weight) -> sp.csr_matrix:
    """Adjacency A (float64 CSR) of the undirected multigraph.

    Follows ``HippoRAG.py:1189-1223`` (add_new_edges: self pairs skipped, one
    igraph edge per ``node_to_node_stats`` key) + igraph's undirected handling
    (each edge seen from both endpoints; parallel edges sum).
    """
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    w = np.asarray(weight, dtype=np.float64)
    keep = src != dst
    src, dst, w = src[keep], dst[keep], w[keep]
    rows = np.concatenate([src, dst])
    cols = np.concatenate([dst, src])
    vals = np.concatenate([w, w])
    a = sp.coo_matrix((vals, (rows, cols)), shape=(num_vertices, num_vertices))
    a = a.tocsr()  # sums duplicates
    a.sort_indices()
    return a


def column_normalize(a: sp.csr_matrix) -> sp.csr_matrix:
    """P[i,j] = A[i,j] / colsum_j (fp64); zero columns (dangling) stay zero."""
    a = a.tocsr().astype(np.float64)
    colsum = np.asarray(a.sum(axis=0)).ravel()
    inv = np.zeros_like(colsum)
    nz = colsum > 0
    inv[nz] = 1.0 / colsum[nz]
    # scale column j by inv[j]:  P = A @ diag(inv); do it on the data array so that
    # the rounding is a single fp64 division-equivalent per entry.
    p = a.copy()
    p.data = a.data / colsum[a.indices]
    p.sort_indices()
    return p


def _sanitise_reset(reset) -> np.ndarray:
    """``HippoRAG.py:1735``: NaN / negative entries -> 0."""
    r = np.asarray(reset, dtype=np.float64)
    return np.where(np.isnan(r) | (r < 0), 0.0, r)


def ppr_exact(p: sp.csr_matrix, reset, alpha: float = 0.5, method: str = "auto") -> np.ndarray:
    """Exact PPR vector x = normalise_1((I - alpha P)^-1 v), fp64.

    method "solve": sparse LU; "power": fp64 leaky power iteration run until the
    update is below 1e-15 of the mass (alpha^k decay => <= ~60 sweeps at 0.5).
    """
    r = _sanitise_reset(reset)
    s = r.sum()
    if not s > 0:
        # igraph raises on an all-zero reset; the reference asserts before
        # (HippoRAG.py:1644).  Mirror as ValueError.
        raise ValueError("reset vector has no positive entry")
    v = r / s
    n = p.shape[0]
    if method == "auto":
        # sparse LU fill-in explodes on random graphs (11 s at n = 6000); the fp64 power iteration
        # agrees with it to 2e-15 (tests/test_oracle_ppr.py) and takes milliseconds
        method = "solve" if n <= 1500 else "power"
    if method == "solve":
        m = sp.identity(n, format="csc", dtype=np.float64) - alpha * p.tocsc()
        x = spla.spsolve(m, v)
    elif method == "power":
        x = v.copy()
        for _ in range(200):
            xn = alpha * (p @ x) + (1.0 - alpha) * v
            d = np.abs(xn - x).sum()
            x = xn
            if d <= 1e-16 * np.abs(x).sum():
                break
    else:
        raise ValueError(method)
    x = np.asarray(x, dtype=np.float64)
    return x / x.sum()


def ppr_power(p: sp.csr_matrix, reset, alpha: float = 0.5, iters: int = 20,
              dtype=np.float64) -> np.ndarray:
    """Fixed-count leaky power iteration (what the device kernel runs):

        x_0 = v ;  x_{k+1} = alpha P x_k + (1 - alpha) v ;  return x_K / sum(x_K)

    Because dangling mass in PRPACK goes to v as well (u == v), the leaky
    iteration converges to a multiple of the PRPACK solution; one final
    normalisation recovers it (SURVEY.md section 7 "hard parts").
    ``reset`` may be un-normalised: the result is scale-invariant.
    """
    r = _sanitise_reset(reset).astype(dtype)
    pd = p.astype(dtype)
    a = dtype(alpha)
    b = dtype(1.0) - a
    x = r.copy()
    for _ in range(iters):
        x = a * (pd @ x) + b * r
    x64 = x.astype(np.float64)
    return x64 / x64.sum()


def ppr_prpack_form(p: sp.csr_matrix, reset, alpha: float = 0.5, tol: float = 1e-14) -> np.ndarray:
    """The PRPACK formulation written out literally (explicit dangling term,
    probability vector preserved every sweep).  Only used by tests to show that
    ``ppr_exact`` / ``ppr_power`` compute the same fixed point."""
    r = _sanitise_reset(reset)
    v = r / r.sum()
    colsum = np.asarray(p.sum(axis=0)).ravel()
    dangling = colsum == 0
    x = v.copy()
    for _ in range(500):
        xn = alpha * (p @ x) + (alpha * x[dangling].sum() + (1.0 - alpha)) * v
        d = np.abs(xn - x).sum()
        x = xn
        if d < tol:
            break
    return x / x.sum()
