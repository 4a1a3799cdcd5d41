"""The numerics of the staged fp8 PPR (hipporag_amd/csrc/ppr8.hip), emulated on the CPU.

tools/exp_fp8_final.ppr8 replays the kernel's arithmetic with numpy + torch's e4m3 conversions
(degree-scaled variable, quantised start, static power-of-two scales, fp32 true residual, final
flush).  This pins the claims DESIGN.md section 4 makes about the scheme without a GPU: at 20 sweeps
the result is within a few 1e-7 of the exact PPR vector on the benchmark-like graph and stays
inside the 1e-5 parity bar on a hub-heavy bipartite graph whose own truncation error is 2e-6.
The GPU tests (tests/test_gpu_parity.py) check the real kernels against the same oracle."""

import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from hipporag_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

torch = pytest.importorskip("torch")
if not hasattr(torch, "float8_e4m3fn"):
    pytest.skip("torch without float8_e4m3fn", allow_module_level=True)

from exp_fp8_final import plan_for, ppr8  # noqa: E402


def _problem(a, pv, batch, seed):
    rng = np.random.default_rng(seed)
    a = a.tocsr().astype(np.float64)
    n = a.shape[0]
    d = np.asarray(a.sum(axis=0)).ravel()
    d1 = np.where(d > 0, d, 1.0)
    p = oracle.column_normalize(a)
    at32 = (sp.diags(1.0 / d1) @ a).tocsr().astype(np.float32)
    v = np.zeros((n, batch))
    for q in range(batch):
        pr = rng.standard_normal(len(pv)).astype(np.float32)
        pr = (pr - pr.min()) / (pr.max() - pr.min())             # min-max prior: one exact 0, one exact 1
        v[pv, q] = pr * np.float32(0.05)
        seeds = rng.choice(n, 5, replace=False)
        v[seeds, q] += rng.random(5) * (1.0 if q % 2 == 0 else 1e-3)
    exact = np.stack([oracle.ppr_exact(p, v[:, q], 0.5) for q in range(batch)], 1)
    return at32, d1, v, exact


def test_stage_plan_matches_the_engine():
    """hipporag_amd.engine.fp8_stage_plan mirrors ppr8_plan (csrc/shard.hip): round 5 -- one boundary fewer at the
    benchmark's 20 sweeps (4-sweep stages late, where the residual travels in 3 bytes; the last stage stays 2 sweeps),
    the round-1 plan below 19 sweeps and at small damping."""
    assert plan_for(20) == [1, 2, 3, 4, 4, 4, 2]
    assert plan_for(16) == [1, 2, 3, 3, 3, 3, 1] and plan_for(18) == [1, 2, 3, 3, 3, 3, 3]
    assert plan_for(19) == [1, 2, 3, 3, 4, 4, 2] and plan_for(24) == [1, 2, 3, 4, 4, 4, 4, 2]
    assert plan_for(20, 0.3) == [1, 2, 3, 3, 3, 3, 3, 2]
    from hipporag_amd.engine import fp8_stage_plan
    assert fp8_stage_plan(20, 0.5, tol=1.5e-6) == [1, 2, 3, 3, 4, 4, 2, 1]      # under a tolerance: ends on 2 + 1 (lower final measure)
    assert all(sum(fp8_stage_plan(k, 0.5, tol=1e-6)) == k and len(fp8_stage_plan(k, 0.5, tol=1e-6)) <= 11 for k in range(16, 31))
    for al in (0.3, 0.5, 0.6):
        assert all(sum(plan_for(k, al)) == k and len(plan_for(k, al)) <= 12 for k in range(16, 31))   # kP8MaxStages


def test_the_round5_plan_is_as_accurate_as_the_plan_it_replaces_and_reports_the_same_residual():
    """20 sweeps, emulation of the device arithmetic: 1+2+3+4+4+4+2 (five boundaries) against 1+2+3+3+3+3+3+2 (six) --
    the true error AND the contract's measure (what a final sweep reports as the residual)."""
    kg = synth.make_kg(20_000, 200_000, 1236)
    a = oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight)
    at32, d1, v, exact = _problem(a, kg.passage_vertex, 8, seed=5)
    pv = kg.passage_vertex
    xn, mn = ppr8(at32, d1, v, 0.5, plan_for(20), measure_rows=pv)
    xo, mo = ppr8(at32, d1, v, 0.5, [1, 2, 3, 3, 3, 3, 3, 2], measure_rows=pv)
    new, old = np.abs(xn[pv] / exact[pv] - 1).max(), np.abs(xo[pv] / exact[pv] - 1).max()
    assert new < 1.3 * old and new < 1.5e-6, (new, old)
    assert mn.max() < 1.5 * mo.max(), (mn.max(), mo.max())
    # the variant that ends on a 3-sweep stage: same error, but its last right-hand side is rounded one sweep earlier
    # and the measure reads that (the reason the plan ends on 2)
    x3, m3 = ppr8(at32, d1, v, 0.5, [1, 2, 3, 3, 4, 4, 3], measure_rows=pv)
    assert np.abs(x3[pv] / exact[pv] - 1).max() < 1.3 * old and m3.max() > 1.5 * mo.max(), (m3.max(), mo.max())


def test_fp8_scheme_reaches_fp32_level_accuracy_on_the_benchmark_graph():
    kg = synth.make_kg(20_000, 200_000, 1236)
    a = oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight)
    at32, d1, v, exact = _problem(a, kg.passage_vertex, 8, seed=5)
    x = ppr8(at32, d1, v, 0.5, plan_for(20))
    pv = kg.passage_vertex
    rel = np.abs(x[pv] / exact[pv] - 1)
    assert rel.max() < 1.5e-6, rel.max()
    # and it is the refinement that does it: a plain fp8 state (one stage, no residual) is useless
    x1 = ppr8(at32, d1, v, 0.5, [1, 19])
    assert np.abs(x1[pv] / exact[pv] - 1).max() > 1e-3


def test_fp8_scheme_on_a_hub_heavy_bipartite_graph_stays_inside_the_parity_bar():
    n = 5000
    hubs, leaves = np.arange(10), np.arange(10, n)
    src = np.concatenate([leaves, hubs[:-1]])
    dst = np.concatenate([hubs[(leaves - 10) % 10], hubs[1:]])
    a = oracle.build_symmetric_csr(n, src, dst, np.ones(len(src)))
    pv = leaves[::8]
    at32, d1, v, exact = _problem(a, pv, 8, seed=6)
    x = ppr8(at32, d1, v, 0.5, plan_for(20))
    p = oracle.column_normalize(a)
    x20 = np.stack([oracle.ppr_power(p, v[:, q], 0.5, 20) for q in range(8)], 1)
    err8 = np.abs(x[pv] / exact[pv] - 1).max()
    err20 = np.abs(x20[pv] / exact[pv] - 1).max()
    assert err8 < 1e-5 and err8 < 6 * err20, (err8, err20)
