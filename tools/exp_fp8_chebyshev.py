"""Numerics experiment (CPU, not product code): Chebyshev semi-iteration INSIDE the stages of the staged fp8 PPR.

A stage solves (I - G) c = R', G = a At (spectrum in [-a, a]: At = D^-1 A is similar to a symmetric matrix on an
undirected graph), starting from c_0 = 0, c_1 = R'.  Plain: c_{k+1} = G c_k + R' (error polynomial lambda^m, max a^m);
Chebyshev: c_{k+1} = w_{k+1} (G c_k + R' - c_{k-1}) + c_{k-1}, w_1 = 1, w_2 = 1 / (1 - a^2 / 2),
w_{k+1} = 1 / (1 - a^2 w_k / 4) (max error 1 / T_m(1 / a): a = 0.5 -> 1/2, 1/7, 1/26, 1/97).  With c_0 = 0 and
c_1 = R' the first two steps need no history: c_2 = w_2 (G c_1 + R'), c_3 = w_3 G c_2 + R'.

    python tools/exp_fp8_chebyshev.py
Prints, per graph, the worst relative error at the passage vertices against the exact solution for the product's plan
(20 sweeps) and for Chebyshev plans with fewer sweeps.
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
from exp_fp8_final import plan_for  # noqa: E402


def omegas(alpha, m):
    w = [1.0]
    if m >= 2:
        w.append(1.0 / (1.0 - alpha * alpha / 2.0))
    while len(w) < m:
        w.append(1.0 / (1.0 - alpha * alpha * w[-1] / 4.0))
    return w            # w[k-1] = w_k


def ppr8(at32, d1, v, alpha, plan, cheb, rho_form=True, state_q=q8):
    al, be = np.float32(alpha), np.float32(1 - alpha)
    zv = (v / d1[:, None])
    s0 = zv.max(axis=0)
    qs = np.exp2(-np.ceil(np.log2(np.maximum(s0, 1e-300))))
    zv = (zv * qs).astype(np.float32)
    R = be * zv
    c = state_q(zv * np.float32(128.0)); inv = np.float32(1 / 128.0)
    X = np.zeros_like(zv, dtype=np.float64)
    bound = max(alpha, 1 - alpha) + 0.07

    def contraction(m):
        if not cheb:
            return alpha ** m
        x = 1.0 / alpha                      # 1 / T_m(1 / alpha)
        t0, t1 = 1.0, x
        for _ in range(m - 1):
            t0, t1 = t1, 2 * x * t1 - t0
        return 1.0 / t1 if m >= 1 else 1.0

    def scale_for(m):
        growth = (1 - alpha ** m) / (1 - alpha) if alpha < 1 else m
        if cheb:
            growth *= 1.15
        return np.float32(2.0 ** math.floor(math.log2(224.0 / (bound * max(growth, 1.0)))))

    k_done, r16, rho, rt = 0, False, None, None
    cs_next = scale_for(plan[1]) if len(plan) > 1 else np.float32(1)
    sat = 0
    for si, m in enumerate(plan):
        if si > 0:
            cs = cs_next
            inv = np.float32(1.0) / cs
            w = omegas(alpha, m) if cheb else [1.0] * m
            c_prev = np.zeros_like(rt)
            c = rt
            for k in range(2, m + 1):
                wk = np.float32(w[k - 1])
                new = wk * (al * (at32 @ c) + rt - c_prev) + c_prev
                sat = max(sat, float(np.abs(new).max()))
                c_prev, c = c, state_q(new.astype(np.float32))
            # the true residual contracts by ~ max(contraction, rounding): keep the product's bound (a^m) when the
            # stage is plain, the Chebyshev figure + the rounding floor otherwise
            bound *= max(contraction(m), 2.0 ** -4) if cheb else alpha ** m
            cs_next = scale_for(plan[si + 1]) if si + 1 < len(plan) else np.float32(1)
        k_done += m
        r_in = ((rt + rho) * inv).astype(np.float32) if r16 else R
        R = (r_in + (al * (at32 @ c) - c) * inv).astype(np.float32)
        X = X + c.astype(np.float64) * inv
        if si + 1 < len(plan):
            q = (R * cs_next).astype(np.float32)
            sat = max(sat, float(np.abs(q).max()))
            rt = state_q(q)
            r16 = rho_form and si > 0 and alpha ** k_done <= 1.0 / 64.0
            if r16:
                rho = (q - rt).astype(np.float16).astype(np.float32)
    z = X + R
    x = z * d1[:, None]
    return x / x.sum(0), sat


def main():
    rng = np.random.default_rng(5)
    B = 16
    plans = {
        "product 20": (plan_for(20), False),
        "plain 17": (plan_for(17), False),
        "cheb 1,2x8 (17)": ([1] + [2] * 8, True),
        "cheb 1,2x7 (15)": ([1] + [2] * 7, True),
        "cheb 1,2,3x5 (18)": ([1, 2] + [3] * 5, True),
        "cheb 1,2,3x4 (15)": ([1, 2] + [3] * 4, True),
        "cheb 1,2,3x4,2 (17)": ([1, 2] + [3] * 4 + [2], True),
        "cheb 1,3x5 (16)": ([1] + [3] * 5, True),
        "cheb 1,4x4 (17)": ([1] + [4] * 4, True),
    }
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
        print(name, flush=True)
        for pname, (plan, cheb) in plans.items():
            x8, sat = ppr8(at32, d1, v, 0.5, plan, cheb)
            err = np.abs(x8[pv] / xe[pv] - 1)
            print(f"   {pname:22s} sweeps {sum(plan):2d} boundaries {len(plan) - 1:2d}  max rel err {err.max():.2e}  median {np.median(err.max(0)):.2e}  peak |value| {sat:.0f}", flush=True)


if __name__ == "__main__":
    main()
