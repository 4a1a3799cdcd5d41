#!/usr/bin/env python
"""Randomised soak of the index-time entity KNN (GPU): hipporag_amd.knn.retrieve_knn on random shapes (queries
1 .. 3000, keys 5 .. 40000, dims 8 .. 1024 incl. odd ones, k 1 .. 2047, query batch sizes, row norms over two decades,
duplicate clusters, an optional score threshold) against an fp64 restatement of the reference's normalize -> mm -> topk
(utils/embed_utils.py:6-94): ids tie-class aware, scores to 3e-6; thresholded lists = the prefixes of the full lists.

    python tools/soak_knn.py [--seconds 100] [--seed 1]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.helpers import tie_aware_equal  # noqa: E402


def ref_scores(q, keys):
    qn = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
    kn = keys / np.maximum(np.linalg.norm(keys, axis=1, keepdims=True), 1e-12)
    return qn.astype(np.float64) @ kn.astype(np.float64).T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=100.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from hipporag_amd.knn import retrieve_knn
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    n = bad = 0
    while time.time() < t_end:
        nq = int(rng.choice([1, 2, 7, 37, 130, 700, 3000]))
        nk = int(rng.choice([5, 17, 300, 2500, 9000, 40000]))
        dim = int(rng.choice([8, 24, 64, 96, 128, 200, 384, 768, 1024]))       # multiples of 8 (what retrieve_knn accepts)
        k = int(rng.choice([1, 5, 16, 17, 100, 2047]))
        qb = int(rng.choice([1, 16, 64, 1000, 4096]))
        thr = float(rng.choice([-2.0, -2.0, 0.5, 0.8]))             # -2: no threshold
        dup = int(rng.choice([0, 0, 3, 30]))
        seed = int(rng.integers(1, 1 << 30))
        par = dict(nq=nq, nk=nk, dim=dim, k=k, qb=qb, min_score=thr, dup=dup, seed=seed)
        r = np.random.default_rng(seed)
        keys = r.standard_normal((nk, dim)).astype(np.float32) * r.uniform(0.1, 10, (nk, 1)).astype(np.float32)
        if dup and nk > dup + 2:                                     # clusters of near-duplicates: many near-ties at the top
            keys[1: 1 + dup] = keys[0] * r.uniform(0.5, 2.0, (dup, 1)).astype(np.float32) + \
                1e-3 * r.standard_normal((dup, dim)).astype(np.float32)
        q = (keys[r.integers(0, nk, nq)] + 0.3 * r.standard_normal((nq, dim))).astype(np.float32)
        try:
            kw = dict(k=k, query_batch_size=qb, return_arrays=True)
            if thr > -1.5:
                kw["min_score"] = thr
            idx, sc = retrieve_knn(None, None, q, keys, **kw)
            s = ref_scores(q, keys)
            kk = min(k, nk)
            why = ""
            for i in sorted(set(np.linspace(0, nq - 1, min(nq, 6)).astype(int).tolist())):
                order = np.argsort(s[i], kind="stable")[::-1][:kk]
                want_sc = s[i][order]
                got_i = np.asarray(idx[i])
                got_s = np.asarray(sc[i], dtype=np.float64)
                if thr > -1.5:                                       # the prefix above the threshold (ids -1 / scores beyond it dropped)
                    valid = got_i >= 0
                    got_i, got_s = got_i[valid], got_s[valid]
                    m = int((want_sc >= thr).sum())
                    # a score within 3e-6 of the threshold may fall on either side
                    lo, hi = int((want_sc >= thr + 3e-6).sum()), int((want_sc >= thr - 3e-6).sum())
                    if not (lo <= len(got_i) <= hi):
                        why = f"query {i}: {len(got_i)} neighbours above {thr}, expected {m}"
                        break
                    order, want_sc = order[:len(got_i)], want_sc[:len(got_i)]
                if len(got_i) != len(order) or not tie_aware_equal(got_i, order, want_sc, abs_gap=4e-6):
                    why = f"query {i}: ids differ"
                    break
                if len(got_i) and np.abs(got_s - s[i][got_i]).max() > 3e-6:
                    why = f"query {i}: score dev {np.abs(got_s - s[i][got_i]).max():.2e}"
                    break
            ok = not why
        except Exception as exc:  # noqa: BLE001
            ok, why = False, f"{type(exc).__name__}: {str(exc)[:300]}"
        n += 1
        par.update(ok=ok)
        if not ok:
            bad += 1
            par["why"] = why
        print("ok  " if ok else "FAIL", json.dumps(par), flush=True)
    print(f"{n} cases;", "SOAK OK" if bad == 0 else f"SOAK FAILED ({bad})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
