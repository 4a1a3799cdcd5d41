#!/usr/bin/env python
"""Target for rocprofv3 on the index-time KNN: the synonymy call of add_synonymy_edges
(reference src/hipporag/HippoRAG.py:959-1020 -> utils/embed_utils.py:6-94) at its real shape -- 875 k keys x 768
(fp32-faithful split layout: 2304 fp16 elements per row), query blocks of 4096 -- through hipporag_amd.knn.retrieve_knn.
The kernel under measurement is sim_gemm256_kernel<8, true, true> (tile maxima, fp16): an MFMA-bound shape.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/pmc_knn_target.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from hipporag_amd.knn import retrieve_knn

N, NQ, D = int(os.environ.get("HRAG_KNN_KEYS", 875_000)), int(os.environ.get("HRAG_KNN_QUERIES", 8192)), 768
g = torch.Generator(device="cuda").manual_seed(5)
keys = torch.randn((N, D), generator=g, device="cuda", dtype=torch.float32).cpu().numpy()
idx, sc = retrieve_knn(None, None, keys[:NQ], keys, k=103, return_arrays=True, min_score=0.8)
torch.cuda.synchronize()
assert (idx[:, 0] == range(NQ)).all() and abs(float(sc[:, 0].min()) - 1.0) < 1e-5     # every vector finds itself first
print(f"knn target done: {N} keys x {NQ} queries, MFMA work {2.0 * N * NQ * 3 * D / 1e12:.1f} TFLOP")
