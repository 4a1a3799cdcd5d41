#!/usr/bin/env python
"""A few retrieves of one configuration / batch, to be run under `rocprofv3 --kernel-trace`:
    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python tools/trace_target.py --config cfg2
tools/timeline.py turns the trace of the last retrieve into a start / duration / gap table."""
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
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--calls", type=int, default=4)
    ap.add_argument("--flags", type=int, default=0)
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    V, E, D, seed = cfg["V"], cfg["E"], cfg["D"], cfg["seed"]
    B = args.batch or cfg["B"]
    dev = torch.device("cuda", 0)
    kg = synth.make_kg(V, E, seed)
    pemb = synth.make_embeddings_torch(kg.n_passages, D, seed + 1, dev)
    femb = synth.make_embeddings_torch(kg.n_facts, D, seed + 2, dev)
    eng = HippoRAGEngine(kg.csr, kg.passage_vertex, pemb, femb, kg.subj_vertex, kg.obj_vertex,
                         kg.num_chunks, max_batch=B, max_topk=200, flags=args.flags)
    qf, _ = synth.make_queries_torch(femb, B, 7)
    qp, _ = synth.make_queries_torch(pemb, B, 8)
    cnt = torch.full((B,), 5, dtype=torch.int32, device=dev)
    for _ in range(args.calls):
        idx, sc = eng.score_facts(qf, k=5)
        eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=200)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
