#!/usr/bin/env python
"""Times phase A (fused fact top-k) and the passage GEMM at the cfg-3 shapes; HRAG_SIM_SMALL_TILES / HRAG_GEMM_DBG
select kernel variants (read once per process)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from hipporag_amd import synth
from hipporag_amd.engine import HippoRAGEngine
from hipporag_amd.graph import build_csr
dev = torch.device("cuda", 0)
F, NP, D, B = 875_000, 125_000, 768, int(os.environ.get("B", 256))
femb = synth.make_embeddings_torch(F, D, 3, dev)
pemb = synth.make_embeddings_torch(NP, D, 4, dev)
g = build_csr(4, [0, 1], [1, 2], [1.0, 1.0])
zeros = np.zeros(F, np.int32)
eng = HippoRAGEngine(g, np.full(NP, 3, np.int32), pemb, femb, zeros, zeros, np.zeros(4, np.int32), max_batch=B, max_topk=200)
qf, _ = synth.make_queries_torch(femb, B, 7)
qp, _ = synth.make_queries_torch(pemb, B, 8)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print(json.dumps({"env": {k: os.environ.get(k) for k in ("HRAG_SIM_SMALL_TILES", "HRAG_GEMM_DBG", "B")},
                  "score_facts_ms": t(lambda: eng.score_facts(qf, k=5)), "passage_gemm_ms": t(lambda: eng.sim_scores("passages", qp))}))
