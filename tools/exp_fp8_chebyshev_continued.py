"""Numerics experiment (CPU, not product code), round 5: carry the Chebyshev recurrence ACROSS the stage boundaries of
the staged fp8 PPR instead of restarting it at c_0 = 0 in every stage (round-4 review, next #1: "an un-restarted degree-15
polynomial would give 1/T_15(2) = 5e-9 where five restarted 3-sweep stages give 1/26^5 = 8e-8").

Global semi-iteration for (I - G) z = b, spectrum of G inside [-a, a]:
    z_{k+1} = z_k + D_{k+1},   D_{k+1} = w_{k+1} r_k + (w_{k+1} - 1) D_k,   r_k = b - (I - G) z_k,
    w_1 = 1, w_2 = 1 / (1 - a^2 / 2), w_{k+1} = 1 / (1 - a^2 w_k / 4)      (-> 2 / (1 + sqrt(1 - a^2)) = 1.072 at a = 0.5).
In the staged scheme z_k = X + c_k / cs with the stage's right-hand side R' = Q(true residual at the stage's start):
inside a stage c_{k+1} = c_k + w (G c_k + R' - c_k) + (w - 1)(c_k - c_{k-1}); at a boundary the history term is the LAST
update of the previous stage, an own-row quantity (c_m - c_{m-1}, one more byte per row to read), rescaled to the new
stage's scale: c_1 = Q(w R' + (w - 1) D_prev).  Every boundary still forms the TRUE residual, so refinement stays exact.

Scales: the ideal power of two from the measured max |R| of the boundary itself (the device predicts one stage ahead; this
is the favourable case for both variants).  Printed per graph: worst relative error at the passage vertices against the
exact solution for the plain product plan (20), the restarted accelerated plan the product ships (1,3,3,3,3,3 = 16) and
restarted / continued plans of 13 .. 16 sweeps.

    python tools/exp_fp8_chebyshev_continued.py
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
from exp_fp8_final import plan_for, ppr8 as ppr8_plain  # noqa: E402


def omega_seq(alpha, n):
    w = [1.0, 1.0 / (1.0 - alpha * alpha / 2.0)]
    while len(w) < n + 2:
        w.append(1.0 / (1.0 - alpha * alpha * w[-1] / 4.0))
    return w


def ppr8_cheb(at32, d1, v, alpha, plan, continued, measure_rows=None):
    """Stage 0 = the quantised start (plain, as the product); stages >= 1 run Chebyshev steps -- restarted per stage
    (the product's HRAG_OPT_ACCEL) or continued across the boundaries (this experiment)."""
    al, be = np.float32(alpha), np.float32(1 - alpha)
    zv = v / d1[:, None]
    qs = np.exp2(-np.ceil(np.log2(np.maximum(zv.max(axis=0), 1e-300))))
    zv = (zv * qs).astype(np.float32)
    R = be * zv
    c = q8(zv * np.float32(128.0)); inv = np.float32(1 / 128.0)
    X = np.zeros_like(zv, dtype=np.float64)
    R = (R + (al * (at32 @ c) - c) * inv).astype(np.float32)      # boundary closing stage 0
    X = X + c.astype(np.float64) * inv
    w_glob = omega_seq(alpha, sum(plan) + 4)
    k_glob = 1                                   # global Chebyshev step counter (continued variant)
    d_prev = None                                # last update of the previous stage, in units of ITS scale, and that scale
    cs_prev = None
    for si, m in enumerate(plan[1:], start=1):
        # ideal measured scale: the stage's iterate grows to <= ~2.2 x the right-hand side (w2 (1 + a) + history)
        mx = float(np.abs(R).max())
        cs = np.float32(2.0 ** math.floor(math.log2(224.0 / max(mx * 2.4, 1e-30))))
        inv = np.float32(1.0) / cs
        rt = q8((R * cs).astype(np.float32))
        w_loc = omega_seq(alpha, m + 2)
        c_prev = np.zeros_like(rt)
        if continued and d_prev is not None:
            wk = np.float32(w_glob[k_glob])
            hist = (d_prev * (cs / cs_prev)).astype(np.float32)
            c = q8((wk * rt + (wk - np.float32(1)) * hist).astype(np.float32))
        else:
            c = rt                               # c_1 = R' (w_1 = 1)
        k_glob += 1
        for k in range(2, m + 1):
            wk = np.float32(w_glob[k_glob] if continued else w_loc[k - 1])
            new = wk * (al * (at32 @ c) + rt - c_prev) + c_prev if not continued else \
                c + wk * (al * (at32 @ c) + rt - c) + (wk - np.float32(1)) * (c - c_prev)
            c_prev, c = c, q8(new.astype(np.float32))
            k_glob += 1
        d_prev, cs_prev = (c - c_prev).astype(np.float32), cs
        R = ((rt * inv) + (al * (at32 @ c) - c) * inv + (R - rt * inv)).astype(np.float32)   # true residual: R + (G c - c) / cs
        X = X + c.astype(np.float64) * inv
    z = X + R
    x = z * d1[:, None]
    if measure_rows is not None:
        zz = z[measure_rows]
        meas = np.where(zz > 0, np.abs(R[measure_rows]) / np.where(zz > 0, zz, 1), 0.0).max(axis=0) * alpha / (1 - alpha)
        return x / x.sum(0), meas
    return x / x.sum(0)


def main():
    rng = np.random.default_rng(5)
    B = 16
    plans = [("restarted 1,3x5 (16, product)", [1] + [3] * 5, False), ("restarted 1,3x4 (13)", [1] + [3] * 4, False),
             ("restarted 1,3x4,2 (15)", [1] + [3] * 4 + [2], False),
             ("continued 1,3x5 (16)", [1] + [3] * 5, True), ("continued 1,3x4,2 (15)", [1] + [3] * 4 + [2], True),
             ("continued 1,3x4,1 (14)", [1] + [3] * 4 + [1], True), ("continued 1,3x4 (13)", [1] + [3] * 4, True),
             ("continued 1,4x3 (13)", [1] + [4] * 3, True), ("continued 1,2x6 (13)", [1] + [2] * 6, True),
             ("continued 1,4,4,5 (14)", [1, 4, 4, 5], True)]
    for name, (a, pv) in graphs().items():
        if name == "ring":
            continue
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
        x20 = ppr8_plain(at32, d1, v, 0.5, plan_for(20))
        print(f"{name}: plain product plan (20 sweeps) max rel err {np.abs(x20[pv] / xe[pv] - 1).max():.2e}", flush=True)
        for pname, plan, cont in plans:
            x = ppr8_cheb(at32, d1, v, 0.5, plan, cont)
            err = np.abs(x[pv] / xe[pv] - 1)
            print(f"   {pname:32s} sweeps {sum(plan):2d}  max rel err {err.max():.2e}  median {np.median(err.max(0)):.2e}", flush=True)


if __name__ == "__main__":
    main()
