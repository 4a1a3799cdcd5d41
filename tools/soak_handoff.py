#!/usr/bin/env python
"""Soak test of the in-kernel hand-offs (long-row combine, fused top-k selection): the same batch many times at full
size (cfg 3) through every kernel family -- every call must reproduce the first one bit for bit.
    python tools/soak_handoff.py [--calls 200]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from bench import CONFIGS
from hipporag_amd import synth
from hipporag_amd.engine import HippoRAGEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=200)
    ap.add_argument("--config", default="cfg3")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    V, E, D, seed = cfg["V"], cfg["E"], cfg["D"], cfg["seed"]
    dev = torch.device("cuda", 0)
    kg = synth.make_kg(V, E, seed)
    pemb = synth.make_embeddings_torch(kg.n_passages, D, seed + 1, dev)
    femb = synth.make_embeddings_torch(kg.n_facts, D, seed + 2, dev)
    bad = 0
    for B, calls in ((256, args.calls), (64, args.calls), (1, 2 * args.calls), (384, args.calls // 2)):
        eng = HippoRAGEngine(kg.csr, kg.passage_vertex, pemb, femb, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                             max_batch=B, max_topk=200)
        qf, _ = synth.make_queries_torch(femb, B, 7)
        qp, _ = synth.make_queries_torch(pemb, B, 8)
        cnt = torch.full((B,), 5, dtype=torch.int32, device=dev)
        ref = None
        for i in range(calls):
            idx, sc = eng.score_facts(qf, k=5)
            out = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=200)
            cur = (idx.clone(), sc.clone(), out.doc_idx.clone(), out.doc_score.clone())
            if ref is None:
                ref = cur
            elif not all(torch.equal(a, b) for a, b in zip(ref, cur)):
                bad += 1
                print(f"B={B}: call {i} differs from call 0", flush=True)
        torch.cuda.synchronize()
        print(f"B={B}: {calls} calls, slab width {eng.timings()['slab_width']}, mismatches so far {bad}", flush=True)
        eng.close()
    print("SOAK", "OK" if bad == 0 else f"FAILED ({bad})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
