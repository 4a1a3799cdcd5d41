"""Pin the oracle's PPR against independent formulations (the reference pins nothing here:
SURVEY.md 8c "parity unpinned"): networkx.pagerank (networkx 3.4.2 is the version the reference
lists in requirements.txt:7), a sparse direct solve, the literal PRPACK formulation and the C port."""

import networkx as nx
import numpy as np
import pytest

import oracle
from oracle import ppr as oppr
from oracle.prpack_port import PrpackCSR


def random_multigraph(n, m, seed, isolated=0):
    rng = np.random.default_rng(seed)
    live = n - isolated
    src = rng.integers(0, live, m)
    dst = rng.integers(0, live, m)
    w = rng.choice([1.0, 2.0, 0.85, 3.0], m)
    # reversed duplicates (the reference's (s,o)/(o,s) parallel edges) + a few self loops
    src = np.concatenate([src, dst[: m // 4], np.arange(3)])
    dst = np.concatenate([dst, src[: m // 4], np.arange(3)])
    w = np.concatenate([w, w[: m // 4], np.ones(3)])
    return src, dst, w


def nx_pagerank(n, src, dst, w, reset, alpha):
    g = nx.MultiGraph()
    g.add_nodes_from(range(n))
    for u, v, ww in zip(src.tolist(), dst.tolist(), w.tolist()):
        if u != v:
            g.add_edge(u, v, weight=ww)
    pers = {i: float(reset[i]) for i in range(n)}
    pr = nx.pagerank(g, alpha=alpha, personalization=pers, dangling=pers, weight="weight",
                     tol=1e-15, max_iter=1000)
    return np.array([pr[i] for i in range(n)])


@pytest.mark.parametrize("n,m,isolated", [(100, 400, 0), (200, 900, 5), (300, 1200, 5)])
def test_exact_matches_networkx_and_prpack_form(n, m, isolated):
    src, dst, w = random_multigraph(n, m, seed=n, isolated=isolated)
    p = oracle.column_normalize(oracle.build_symmetric_csr(n, src, dst, w))
    rng = np.random.default_rng(7)
    reset = np.zeros(n)
    reset[rng.integers(0, n - isolated, 5)] = rng.random(5) + 0.1
    reset[n - isolated:] += 0.03          # seeded dangling vertices
    reset[::7] += 0.01 * rng.random(len(reset[::7]))
    for alpha in (0.5, 0.85):
        x = oracle.ppr_exact(p, reset, alpha, method="solve")
        assert abs(x.sum() - 1) < 1e-12
        np.testing.assert_allclose(x, nx_pagerank(n, src, dst, w, reset, alpha), rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(x, oppr.ppr_prpack_form(p, reset, alpha), rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(x, oracle.ppr_exact(p, reset, alpha, method="power"), rtol=1e-10, atol=1e-15)


@pytest.mark.parametrize("n", [60, 127, 128, 400])
def test_c_port_matches_exact(n):
    # straddles PRPACK's 128-vertex switch between Gaussian elimination and Gauss-Seidel
    src, dst, w = random_multigraph(n, 5 * n, seed=3 * n, isolated=4)
    p = oracle.column_normalize(oracle.build_symmetric_csr(n, src, dst, w))
    rng = np.random.default_rng(n)
    reset = rng.random(n) * (rng.random(n) < 0.2)
    reset[0] = 1.0
    reset[n - 1] = 0.5                    # a seeded isolated vertex
    x = oracle.ppr_exact(p, reset, 0.5, method="solve")
    c = PrpackCSR(p)
    xc, sweeps = c.solve(reset, 0.5, "prpack")
    assert (sweeps == 0) == (n < 128)
    np.testing.assert_allclose(xc, x, rtol=1e-8, atol=1e-11)
    xg, _ = c.solve(reset, 0.5, "gs")
    xe, _ = c.solve(reset, 0.5, "ge")
    np.testing.assert_allclose(xg, x, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(xe, x, rtol=1e-10, atol=1e-14)


def test_power_20_iterations_is_within_budget():
    # the device kernel runs 20 leaky sweeps in fp32: check the truncation + fp32 error budget
    from tests.helpers import make_case
    kg, _, _, index = make_case(4000, 40000, 32, seed=11)
    rng = np.random.default_rng(0)
    reset = np.zeros(kg.num_vertices)
    reset[kg.passage_vertex] = 0.05 * rng.random(kg.n_passages).astype(np.float32)
    reset[rng.integers(0, kg.n_entities, 5)] = rng.random(5)
    x = oracle.ppr_exact(index.p, reset, 0.5)
    x20 = oracle.ppr_power(index.p, reset, 0.5, iters=20)
    x20_32 = oracle.ppr_power(index.p, reset, 0.5, iters=20, dtype=np.float32)
    pv = kg.passage_vertex
    nz = x[pv] > 0
    assert np.max(np.abs(x20[pv][nz] - x[pv][nz]) / x[pv][nz]) < 2e-6
    assert np.max(np.abs(x20_32[pv][nz] - x[pv][nz]) / x[pv][nz]) < 1e-5


def test_reset_sanitised_and_zero_mass():
    n = 50
    src, dst, w = random_multigraph(n, 200, seed=1)
    p = oracle.column_normalize(oracle.build_symmetric_csr(n, src, dst, w))
    reset = np.zeros(n)
    reset[3] = 1.0
    dirty = reset.copy()
    dirty[5] = np.nan
    dirty[6] = -2.0                        # HippoRAG.py:1735
    np.testing.assert_allclose(oracle.ppr_exact(p, dirty), oracle.ppr_exact(p, reset))
    with pytest.raises(ValueError):
        oracle.ppr_exact(p, np.zeros(n))
    with pytest.raises(ValueError):
        PrpackCSR(p).solve(np.zeros(n))
