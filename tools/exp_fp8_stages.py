"""Numerics experiment (CPU, not product code): can the PPR state go below fp16?

Emulates the staged residual-correction iteration of csrc/ppr16.hip with a chain of low-precision
stages and reports the max relative error over ALL passage vertices against the exact fp64 solution.

  stage 0      h  in fp16, K0 sweeps                                 (mode H)
  boundary     R  = (alpha P xhat + beta v) - xhat   (fp32, true residual; xhat = h + sum c_s)
               rt = quant(R * cs)                                    (mode R)
  stage s      c <- quant(alpha P c + rt), c_1 = rt, m_s sweeps incl. the boundary sweep  (mode C)
  result       x  = h + sum_s c_s / cs_s

quant = fp16 | e4m3 | e5m2 (torch CPU conversions, round-to-nearest-even, saturating).

    python tools/exp_fp8_stages.py [V E B]
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from tests.helpers import make_case  # noqa: E402
from hipporag_amd import synth  # noqa: E402
from hipporag_amd.graph import bf16_bits_to_float  # noqa: E402

FMT = {
    "f16": (torch.float16, 65504.0),
    "e4m3": (torch.float8_e4m3fn, 448.0),
    "e5m2": (torch.float8_e5m2, 57344.0),
}


def quant(a: np.ndarray, fmt: str) -> np.ndarray:
    dt, mx = FMT[fmt]
    t = torch.from_numpy(np.clip(a, -mx, mx).astype(np.float32))
    return t.to(dt).to(torch.float32).numpy()


def staged(p32, v, alpha, plan, verbose=False):
    """p32: scipy csr fp32 [V,V]; v: fp32 [V,B] teleport (scaled so sum in (2^14, 2^15]).
    plan: list of (fmt, n_sweeps); the first entry is the h stage."""
    beta = np.float32(1.0 - alpha)
    al = np.float32(alpha)
    fmt0, k0 = plan[0]
    h = quant(v, fmt0)
    for _ in range(k0):
        h = quant(al * (p32 @ h) + beta * v, fmt0)
    xhat = h.astype(np.float64)          # exact sum of exactly representable parts (for the check only)
    # true residual of h, fp32 arithmetic like the kernel
    R = (al * (p32 @ h) + beta * v) - h
    for fmt, m in plan[1:]:
        mx = FMT[fmt][1]
        # per-query power-of-two scale: max|R| * cs in (mx/4, mx/2]
        amax = np.abs(R).max(axis=0)
        cs = np.exp2(np.floor(np.log2(mx / 2 / np.maximum(amax, 1e-30)))).astype(np.float32)
        rt = quant(R * cs, fmt)
        c = rt.copy()
        for _ in range(m - 1):
            c = quant(al * (p32 @ c) + rt, fmt)
        # new true residual: R' = R + (alpha P c - c) / cs     (fp32)
        R = R + (al * (p32 @ c) - c) / cs
        xhat = xhat + c.astype(np.float64) / cs.astype(np.float64)
    return xhat


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    kg, pass_bits, fact_bits, index = make_case(V, E, 64, seed=1236)
    qf_bits, _ = synth.make_queries_np(fact_bits, B, seed=1)
    qp_bits, _ = synth.make_queries_np(pass_bits, B, seed=2)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    resets, exact = [], []
    for q in range(B):
        r = oracle.retrieve_one(index, qf[q], qp[q])
        resets.append(r.reset)
        exact.append(r.x)
    v = np.stack(resets, 1)
    # per-query power-of-two scale so that sum(v) in (2^14, 2^15]
    s = np.exp2(np.floor(np.log2(32768.0 / v.sum(0))))
    v32 = (v * s).astype(np.float32)
    xe = np.stack(exact, 1)
    p32 = index.p.astype(np.float32).tocsr()
    pv = kg.passage_vertex
    plans = {
        "f16x10+f16x10 (current)": [("f16", 10), ("f16", 10)],
        "f16x11+e4m3x3x3": [("f16", 11), ("e4m3", 3), ("e4m3", 3), ("e4m3", 3)],
        "f16x12+e4m3x4x2": [("f16", 12), ("e4m3", 4), ("e4m3", 4)],
        "f16x10+e4m3x5x2": [("f16", 10), ("e4m3", 5), ("e4m3", 5)],
        "f16x11+e5m2x3x3": [("f16", 11), ("e5m2", 3), ("e5m2", 3), ("e5m2", 3)],
        "f16x10+e5m2x2x5": [("f16", 10)] + [("e5m2", 2)] * 5,
        "f16x8+e4m3x3x4": [("f16", 8)] + [("e4m3", 3)] * 4,
        "f16x8+e4m3x2x6": [("f16", 8)] + [("e4m3", 2)] * 6,
        "e4m3x2x10": [("e4m3", 2)] * 10,
        "f16x12+f16x4+e4m3x4": [("f16", 12), ("f16", 4), ("e4m3", 4)],
    }
    # reference: plain fp32 20 sweeps
    x = v32.copy()
    for _ in range(20):
        x = np.float32(0.5) * (p32 @ x) + np.float32(0.5) * v32
    x = x.astype(np.float64)
    x /= x.sum(0)
    print(f"V={V} E={E} B={B}  fp32 x20: max rel err passages {np.abs(x[pv] / xe[pv] - 1).max():.3e}")
    for name, plan in plans.items():
        assert sum(m for _, m in plan) == 20, name
        xh = staged(p32, v32, 0.5, plan)
        xh = xh / xh.sum(0)
        rel = np.abs(xh[pv] / xe[pv] - 1)
        relall = np.abs(xh / np.maximum(xe, 1e-300) - 1)[xe > 0]
        print(f"{name:28s} passages max {rel.max():.3e}  p99.9 {np.quantile(rel, 0.999):.3e}   all-vertex max {relall.max():.3e}")


if __name__ == "__main__":
    main()
