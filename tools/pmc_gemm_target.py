#!/usr/bin/env python
"""Target for the rocprofv3 PMC passes on the similarity GEMM: phase A (fused fact top-k) and the
passage GEMM at the cfg-3 shapes (F = 875k, Np = 125k, D = 768, B = 256), a few launches each.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/pmc_gemm_MfmaUtil -- python tools/pmc_gemm_target.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from hipporag_amd import synth
from hipporag_amd.engine import HippoRAGEngine
from hipporag_amd.graph import build_csr

dev = torch.device("cuda", 0)
F, NP, D, B = 875_000, 125_000, 768, 256
femb = synth.make_embeddings_torch(F, D, 3, dev)
pemb = synth.make_embeddings_torch(NP, D, 4, dev)
g = build_csr(4, [0, 1], [1, 2], [1.0, 1.0])                       # the graph is irrelevant here
zeros = np.zeros(F, np.int32)
eng = HippoRAGEngine(g, np.full(NP, 3, np.int32), pemb, femb, zeros, zeros, np.zeros(4, np.int32),
                     max_batch=B, max_topk=200)
qf, _ = synth.make_queries_torch(femb, B, 7)
qp, _ = synth.make_queries_torch(pemb, B, 8)
for _ in range(4):
    eng.score_facts(qf, k=5)
    eng.sim_scores("passages", qp)
torch.cuda.synchronize()
eng.close()
print("pmc gemm target done")
