"""Numerics experiment (CPU, not product code): fp8 (e4m3) PPR state in degree-scaled space.

z = D^-1 x turns the column-stochastic sweep x <- a P x + b v into z <- a At z + b D^-1 v with
At = D^-1 A ROW-stochastic, so the max-norm of the residual contracts by a per sweep and the
per-stage fp8 scales are static:  cs_{s+1} = cs_s * 2^{m_s}.

  X = 0, R = b D^-1 v                       (fp32)
  stage s:  rt = Q(R cs); c = rt; (m_s - 1) x [c <- Q(a At c + rt)];
            boundary sweep: R <- R + (a At c - c)/cs  (fp32),  X += c/cs
  result    z = X + R   (one more exact sweep for free),  x = D z

    python tools/exp_fp8_zspace.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402


def q8(a):
    t = torch.from_numpy(np.clip(a, -448.0, 448.0).astype(np.float32))
    return t.to(torch.float8_e4m3fn).to(torch.float32).numpy()


def staged_z(at32, vz, alpha, plan, top=256.0):
    al, be = np.float32(alpha), np.float32(1 - alpha)
    R = (be * vz).astype(np.float32)
    X = np.zeros_like(R, dtype=np.float64)
    mx = np.abs(R).max(axis=0)
    cs = np.exp2(np.floor(np.log2(top / np.maximum(mx, 1e-30)))).astype(np.float32)
    for m in plan:
        rt = q8(R * cs)
        c = rt.copy()
        for _ in range(m - 1):
            c = q8(al * (at32 @ c) + rt)
        R = (R + (al * (at32 @ c) - c) / cs).astype(np.float32)
        X = X + c.astype(np.float64) / cs
        cs = cs * np.float32(2.0 ** m)
    return X + R


def graphs():
    rng = np.random.default_rng(7)
    out = {}
    # (1) the benchmark generator at cfg2 scale
    from hipporag_amd import synth
    kg = synth.make_kg(100_000, 1_000_000, 1236)
    out["synth cfg2"] = (oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight), kg.passage_vertex)
    kg = synth.make_kg(50_000, 500_000, 99, power_law=True)
    out["synth power-law"] = (oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight), kg.passage_vertex)
    # (2) ring of 4000 (slow mixing, eigenvalues near +-1)
    n = 4000
    i = np.arange(n)
    out["ring"] = (oracle.build_symmetric_csr(n, i, (i + 1) % n, np.ones(n)), i[::8])
    # (3) star forest + chain between hubs (hub rows, bipartite => eigenvalue -1)
    n = 5000
    hubs = np.arange(10)
    leaves = np.arange(10, n)
    src = np.concatenate([leaves, hubs[:-1]])
    dst = np.concatenate([hubs[(leaves - 10) % 10], hubs[1:]])
    out["stars"] = (oracle.build_symmetric_csr(n, src, dst, np.ones(len(src))), leaves[::8])
    # (4) two dense clusters joined by one weak edge + wildly varying weights
    n = 2000
    s1 = rng.integers(0, 1000, 20000); d1 = rng.integers(0, 1000, 20000)
    s2 = rng.integers(1000, 2000, 20000); d2 = rng.integers(1000, 2000, 20000)
    src = np.concatenate([s1, s2, [0]]); dst = np.concatenate([d1, d2, [1999]])
    w = np.concatenate([10.0 ** rng.uniform(-3, 3, 40000), [1e-3]])
    keep = src != dst
    out["barbell wild weights"] = (oracle.build_symmetric_csr(n, src[keep], dst[keep], w[keep]), np.arange(0, n, 8))
    return out


def main():
    rng = np.random.default_rng(11)
    B = 8
    plans = {"4x5": [4] * 5, "5x4": [5] * 4, "2x10": [2] * 10, "3,3,3,3,4,4": [3, 3, 3, 3, 4, 4], "10,10": [10, 10],
             "7,7,6": [7, 7, 6], "4,4,4,4,2,2": [4, 4, 4, 4, 2, 2]}
    for name, (a, pv) in graphs().items():
        a = a.tocsr().astype(np.float64)
        n = a.shape[0]
        d = np.asarray(a.sum(axis=0)).ravel()
        d1 = np.where(d > 0, d, 1.0)
        p = oracle.column_normalize(a)
        at32 = (sp.diags(1.0 / d1) @ a).tocsr().astype(np.float32)
        # reset: passage prior (min-max like: one exact zero, values in [0, 0.05]) + up to 5 seeds
        v = np.zeros((n, B))
        for q in range(B):
            pr = rng.random(len(pv)).astype(np.float32)
            pr = (pr - pr.min()) / (pr.max() - pr.min())
            v[pv, q] = pr * np.float32(0.05)
            seeds = rng.choice(n, 5, replace=False)
            v[seeds, q] += rng.random(5) * (1.0 if q % 2 == 0 else 1e-3)   # odd queries: tiny seeds
        xe = np.stack([oracle.ppr_exact(p, v[:, q], 0.5) for q in range(B)], 1)
        x20 = np.stack([oracle.ppr_power(p, v[:, q], 0.5, 20) for q in range(B)], 1)
        base = np.abs(x20[pv] / xe[pv] - 1).max()
        vz = (v / d1[:, None]).astype(np.float32)
        line = f"{name:22s} fp64 power x20: {base:.2e} |"
        for pn, plan in plans.items():
            z = staged_z(at32, vz, 0.5, plan)
            x = z * d1[:, None]
            x = x / x.sum(0)
            rel = np.abs(x[pv] / xe[pv] - 1).max()
            line += f" {pn}: {rel:.2e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
