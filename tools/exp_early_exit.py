"""Numerics experiment (CPU, not product code): where could a tolerance-driven solve STOP?

The staged fp8 PPR of csrc/ppr8.hip / csrc/shard.hip in the emulation of tools/exp_fp8_final.py, with the passage-row
final sweep (mode F) evaluated after EVERY stage: the result it would return, its true error against the exact fp64
solution, and the contract's measure g * max_p |R_p| / z_p it would report.  Question: on which graphs does the measure
pass a tolerance well before `iters` sweeps, and is it still an upper bound of the error there?

    python tools/exp_early_exit.py [--stages 2]
"""
from __future__ import annotations

import argparse
import math
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import oracle  # noqa: E402
from exp_fp8_zspace import graphs, q8  # noqa: E402


def run(at32, d1, v, pv, xe, alpha, plan):
    """Yields (sweeps if the solve stopped after this stage, true error, measure) for every stage >= 1."""
    al, be = np.float32(alpha), np.float32(1 - alpha)
    g = alpha / (1 - alpha)
    zv = v / d1[:, None]
    s0 = zv.max(axis=0)
    qs = np.exp2(-np.ceil(np.log2(np.maximum(s0, 1e-300))))
    zv = (zv * qs).astype(np.float32)
    R = be * zv
    c = q8(zv * np.float32(128.0))
    inv = np.float32(1 / 128.0)
    X = np.zeros_like(zv, dtype=np.float64)
    bound = max(alpha, 1 - alpha) + 0.07

    def scale_for(m):
        growth = (1 - alpha ** m) / (1 - alpha) if alpha < 1 else m
        return np.float32(2.0 ** math.floor(math.log2(224.0 / (bound * max(growth, 1.0)))))

    k_done, r16, rho, rt = 0, False, None, None
    cs_next = scale_for(plan[1])
    out = []
    for si, m in enumerate(plan):
        if si > 0:
            cs = cs_next
            inv = np.float32(1.0) / cs
            c = rt
            for _ in range(m - 1):
                c = q8(al * (at32 @ c) + rt)
            bound *= alpha ** m
            cs_next = scale_for(plan[si + 1]) if si + 1 < len(plan) else np.float32(1)
        k_done += m
        r_in = ((rt + rho) * inv).astype(np.float32) if r16 else R
        R = (r_in + (al * (at32 @ c) - c) * inv).astype(np.float32)
        X = X + c.astype(np.float64) * inv
        # what mode F after this stage's (m - 1) sweeps returns: z = X + R at the passage rows
        z = (X + R)[pv]
        x = z * d1[pv, None]
        xfull = (X + R) * d1[:, None]
        x = x / xfull.sum(0)
        err = np.abs(x / xe[pv] - 1).max(axis=0)
        meas = g * (np.abs(R[pv]) / np.maximum(z, 1e-300)).max(axis=0)
        out.append((k_done, err, meas))
        if si + 1 < len(plan):
            q = (R * cs_next).astype(np.float32)
            rt = q8(q)
            r16 = si > 0 and alpha ** k_done <= 1.0 / 64.0
            if r16:
                rho = (q - rt).astype(np.float16).astype(np.float32)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tol", type=float, default=1.5e-6)
    args = ap.parse_args()
    rng = np.random.default_rng(5)
    B = 16
    plans = {"1,2,3,3,3,3,3,2": [1, 2, 3, 3, 3, 3, 3, 2], "1,2,2,2,2,2,2,2,2,2,1": [1, 2] + [2] * 8 + [1],
             "1,2,3,2,2,2,2,2,2,2": [1, 2, 3] + [2] * 7}
    for name, (a, pv) in graphs().items():
        a = a.tocsr().astype(np.float64)
        n = a.shape[0]
        d = np.asarray(a.sum(axis=0)).ravel()
        d1 = np.where(d > 0, d, 1.0)
        p = oracle.column_normalize(a)
        at32 = (sp.diags(1.0 / d1) @ a).tocsr().astype(np.float32)
        v = np.zeros((n, B))
        for q in range(B):
            pr = rng.standard_normal(len(pv)).astype(np.float32)
            pr = (pr - pr.min()) / (pr.max() - pr.min())
            v[pv, q] = pr * np.float32(0.05)
            seeds = rng.choice(n, 5, replace=False)
            v[seeds, q] += rng.random(5) * (1.0 if q % 2 == 0 else 1e-3)
        xe = np.stack([oracle.ppr_exact(p, v[:, q], 0.5) for q in range(B)], 1)
        print(f"== {name}")
        for pname, plan in plans.items():
            res = run(at32, d1, v, np.asarray(pv), xe, 0.5, plan)
            stop = next((k for k, e, m in res if m.max() <= args.tol), None)
            line = " ".join(f"{k}:{e.max():.1e}/{m.max():.1e}({(m / np.maximum(e, 1e-30)).min():.2f})" for k, e, m in res)
            print(f"  plan {pname}: stop at {stop}\n    sweeps:err/measure(min measure/err)  {line}", flush=True)


if __name__ == "__main__":
    main()
