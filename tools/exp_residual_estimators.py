"""Numerics experiment (CPU, not product code): which a-posteriori quantity the staged fp8 PPR can report
per query predicts the true relative error of the passage scores -- the basis of the tol / max_iters
contract of hrag_retrieve (reference: PRPACK iterates to an L1 residual of 1e-10, HippoRAG.py:1736-1743).

    python tools/exp_residual_estimators.py
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
from exp_fp8_zspace import graphs, q8  # noqa: E402


def plan_for(iters):
    left = iters - 3
    return [1, 2] + [3] * (left // 3) + ([left % 3] if left % 3 else [])


def ppr8_trace(at32, d1, v, alpha, plan):
    """the emulation of tools/exp_fp8_final.py, returning per-boundary (R all rows, z estimate)"""
    al, be = np.float32(alpha), np.float32(1 - alpha)
    zv = (v / d1[:, None])
    s0 = zv.max(axis=0)
    qs = np.exp2(-np.ceil(np.log2(np.maximum(s0, 1e-300))))
    zv = (zv * qs).astype(np.float32)
    R = be * zv
    c = q8(zv * np.float32(128.0)); inv = np.float32(1 / 128.0)
    X = np.zeros_like(zv, dtype=np.float64)
    bound = max(alpha, 1 - alpha) + 0.07

    def scale_for(m):
        growth = (1 - alpha ** m) / (1 - alpha) if alpha < 1 else m
        return np.float32(2.0 ** math.floor(math.log2(224.0 / (bound * max(growth, 1.0)))))

    k_done, r16, rho, rt = 0, False, None, None
    cs_next = scale_for(plan[1]) if len(plan) > 1 else np.float32(1)
    trace = []
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
        trace.append((k_done, R.copy(), X.copy()))
        if si + 1 < len(plan):
            q = (R * cs_next).astype(np.float32)
            rt = q8(q)
            r16 = si > 0 and alpha ** k_done <= 1.0 / 64.0
            if r16:
                rho = (q - rt).astype(np.float16).astype(np.float32)
    return trace, qs


def main():
    rng = np.random.default_rng(5)
    B = 16
    alpha = 0.5
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
        xe = np.stack([oracle.ppr_exact(p, v[:, q], alpha) for q in range(B)], 1)
        print(f"== {name}  (V={n}, Np={len(pv)})")
        for iters in (20, 23, 26, 29):
            trace, qs = ppr8_trace(at32, d1, v, alpha, plan_for(iters))
            kf, Rf, Xf = trace[-1]
            kb, Rb, Xb = trace[-2]
            z = Xf + Rf
            x = z * d1[:, None]
            xn = x / x.sum(0)
            err = np.abs(xn[pv] / xe[pv] - 1).max(axis=0)          # per query, all passages
            g = alpha / (1 - alpha)
            est_a = g * (np.abs(Rf[pv]) / z[pv]).max(axis=0)
            est_b = g * np.abs(Rf).max(axis=0) / z[pv].min(axis=0)
            est_c = (d1[:, None] * np.abs(Rf)).sum(0) * g / x.sum(0)
            est_d = g * (np.abs(Rf) / np.maximum(z, 1e-300)).max(axis=0)
            rb = np.abs(Rb).max(axis=0)
            rf = np.abs(Rf).max(axis=0)
            print(f"  K={iters}: err max {err.max():.1e} med {np.median(err):.1e} | A(pass dz/z) {est_a.max():.1e} "
                  f"min ratio est/err {np.min(est_a / err):.2f} | B(rigorous) {est_b.max():.1e} | C(L1) {est_c.max():.1e} "
                  f"| D(all dz/z) {est_d.max():.1e} min ratio {np.min(est_d / err):.2f} | |Rb|inf {rb.max():.1e} |Rf|inf {rf.max():.1e} "
                  f"ratio {np.max(rf / rb):.3f} | min z_p {z[pv].min():.1e}", flush=True)


if __name__ == "__main__":
    main()
