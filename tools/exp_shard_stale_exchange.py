"""Numerics experiment (CPU, not product code): the row-sharded fp8 PPR with the state exchanged ONLY at the stage
boundaries -- inside a stage a shard iterates its own rows against the off-shard entries frozen at the stage's first
iterate (block-Jacobi inside the stage; the boundary's true residual keeps the result exact).  How many more sweeps does
the 1e-5 bar cost, against the exchanges saved?  (the answer decides for the hybrid mode instead.)

    python tools/exp_shard_stale_exchange.py
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import oracle  # noqa: E402
from exp_fp8_zspace import q8  # noqa: E402
from hipporag_amd import synth  # noqa: E402


def plan_for(iters):
    left = iters - 3
    return [1, 2] + [3] * (left // 3) + ([left % 3] if left % 3 else [])


def ppr8(at_diag, at_off, d1, v, alpha, plan, stale):
    """tools/exp_fp8_final.py's emulation; stale=True: the C sweeps of a stage see the off-shard columns at rt."""
    at = (at_diag + at_off).tocsr()
    al, be = np.float32(alpha), np.float32(1 - alpha)
    zv = v / d1[:, None]
    qs = np.exp2(-np.ceil(np.log2(np.maximum(zv.max(axis=0), 1e-300))))
    zv = (zv * qs).astype(np.float32)
    R = be * zv
    c = q8(zv * np.float32(128.0)); inv = np.float32(1 / 128.0)
    X = np.zeros_like(zv, dtype=np.float64)
    bound = max(alpha, 1 - alpha) + 0.07
    scale_for = lambda m: np.float32(2.0 ** math.floor(math.log2(224.0 / (bound * max((1 - alpha ** m) / (1 - alpha), 1.0)))))
    rt, exchanges = None, 0
    cs_next = scale_for(plan[1])
    for si, m in enumerate(plan):
        if si > 0:
            cs = cs_next; inv = np.float32(1.0) / cs; c = rt
            off = at_off @ rt if stale else None
            for _ in range(m - 1):
                c = q8(al * ((at_diag @ c + off) if stale else (at @ c)) + rt)
                exchanges += 0 if stale else 1
            if stale and m > 1:
                exchanges += 1                      # the stage's last iterate, before the boundary reads it
            bound *= alpha ** m
            cs_next = scale_for(plan[si + 1]) if si + 1 < len(plan) else np.float32(1)
        R = (R + (al * (at @ c) - c) * inv).astype(np.float32)
        X = X + c.astype(np.float64) * inv
        if si + 1 < len(plan):
            rt = q8((R * cs_next).astype(np.float32))
            exchanges += 1                          # the new right-hand side / first iterate
    x = (X + R) * d1[:, None]
    return x / x.sum(0), exchanges


def main():
    rng = np.random.default_rng(5)
    B, world, alpha = 16, 8, 0.5
    kg = synth.make_kg(100_000, 1_000_000, 1236)
    a = oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight).astype(np.float64)
    n, pv = a.shape[0], kg.passage_vertex
    d = np.asarray(a.sum(axis=0)).ravel(); d1 = np.where(d > 0, d, 1.0)
    p = oracle.column_normalize(a)
    at = (sp.diags(1.0 / d1) @ a).tocsr().astype(np.float32)
    shard = rng.integers(0, world, n)               # the benchmark generator has no locality: a random partition
    coo = at.tocoo()
    same = shard[coo.row] == shard[coo.col]
    at_diag = sp.csr_matrix((coo.data[same], (coo.row[same], coo.col[same])), shape=at.shape)
    at_off = sp.csr_matrix((coo.data[~same], (coo.row[~same], coo.col[~same])), shape=at.shape)
    v = np.zeros((n, B))
    for q in range(B):
        pr = rng.standard_normal(len(pv)).astype(np.float32)
        v[pv, q] = (pr - pr.min()) / (pr.max() - pr.min()) * np.float32(0.05)
        v[rng.choice(n, 5, replace=False), q] += rng.random(5)
    xe = np.stack([oracle.ppr_exact(p, v[:, q], alpha) for q in range(B)], 1)
    print(f"synth cfg2 graph, {world} random shards: {same.mean():.3f} of the entries are shard-local")
    for iters in (20, 23, 26, 29, 32, 38, 44):
        for stale in (False, True):
            x, ex = ppr8(at_diag, at_off, d1, v, alpha, plan_for(iters), stale)
            err = np.abs(x[pv] / xe[pv] - 1).max()
            print(f"  sweeps {iters:2d}  {'boundary-only exchange' if stale else 'exchange every sweep  '}: "
                  f"max rel err {err:.2e}, state exchanges {ex}", flush=True)


if __name__ == "__main__":
    main()
