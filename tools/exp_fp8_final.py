"""Numerics experiment (CPU, not product code): the staged fp8 (e4m3) PPR exactly as csrc/ppr8.hip
runs it -- degree-scaled space, stage 0 = quantised v/d (mass-matched start), static power-of-two
scales, fp32 true residual, final flush -- against the exact fp64 solution.

    python tools/exp_fp8_final.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import oracle  # noqa: E402
from exp_fp8_zspace import graphs, q8  # noqa: E402


def plan_for(iters, damping=0.5):
    """stage lengths -- ppr8_plan in csrc/shard.hip (mirrored by hipporag_amd.engine.fp8_stage_plan)"""
    from hipporag_amd.engine import fp8_stage_plan
    return fp8_stage_plan(iters, damping)


def ppr8(at32, d1, v, alpha, plan, rho_form=True, measure_rows=None):
    """Mirror of ppr8_begin / ppr8_sweep (csrc/shard.hip) + the kernels of csrc/ppr8.hip, fp32 arithmetic.
    rho_form: the true residual travels as (rt + fp16 remainder) once damping^k <= 2^-6, like the device.
    measure_rows (passage rows): also return the contract's measure per query, damping / (1 - damping) *
    max_p |R_p| / z_p of the final sweep (include/hrag.h, hrag_retrieve)."""
    import math
    al, be = np.float32(alpha), np.float32(1 - alpha)
    zv = (v / d1[:, None])
    s0 = zv.max(axis=0)
    qs = np.exp2(-np.ceil(np.log2(np.maximum(s0, 1e-300))))
    zv = (zv * qs).astype(np.float32)                      # max in (0.5, 1]
    R = be * zv
    c = q8(zv * np.float32(128.0)); inv = np.float32(1 / 128.0)
    X = np.zeros_like(zv, dtype=np.float64)
    bound = max(alpha, 1 - alpha) + 0.07                   # |R_0| <= bound * max(v/d)

    def scale_for(m):                                      # static power-of-two scale of a stage of m sweeps
        growth = (1 - alpha ** m) / (1 - alpha) if alpha < 1 else m
        return np.float32(2.0 ** math.floor(math.log2(224.0 / (bound * max(growth, 1.0)))))

    k_done, r16, rho, rt = 0, False, None, None
    cs_next = scale_for(plan[1]) if len(plan) > 1 else np.float32(1)
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
        if si + 1 < len(plan):
            q = (R * cs_next).astype(np.float32)
            rt = q8(q)
            r16 = rho_form and si > 0 and alpha ** k_done <= 1.0 / 64.0
            if r16:
                rho = (q - rt).astype(np.float16).astype(np.float32)
    z = X + R
    x = z * d1[:, None]
    if measure_rows is not None:
        zz = z[measure_rows]
        m = np.where(zz > 0, np.abs(R[measure_rows]) / np.where(zz > 0, zz, 1), 0.0).max(axis=0) * alpha / (1 - alpha)
        return x / x.sum(0), m
    return x / x.sum(0)


def main():
    rng = np.random.default_rng(5)
    B = 16
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
        line = f"{name:22s}"
        for iters in (16, 20, 24):
            xp = np.stack([oracle.ppr_power(p, v[:, q], 0.5, iters) for q in range(B)], 1)
            x8 = ppr8(at32, d1, v, 0.5, plan_for(iters))
            line += f" | K={iters} power {np.abs(xp[pv] / xe[pv] - 1).max():.1e} fp8 {np.abs(x8[pv] / xe[pv] - 1).max():.1e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
