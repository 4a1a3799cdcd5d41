"""The fp16-state PPR of small graphs as ONE cooperative launch (csrc/ppr16.hip ppr16_persist_kernel: grid barriers
between the sweeps) against the launch-per-sweep loop of the same arithmetic (HRAG_OPT_NO_PERSIST): every output must
be bit-identical -- a stale gather across the non-coherent per-XCD L2s, a missed barrier or a lost long-row segment
would show up here.  Replaces igraph's personalized_pagerank behind HippoRAG.run_ppr
(reference src/hipporag/HippoRAG.py:1736-1743); parity with the oracle is covered by tests/test_gpu_parity.py, which
runs on the single-launch path by default at these sizes."""
import numpy as np
import pytest

from helpers import make_case

pytestmark = pytest.mark.gpu


def _inputs(eng, kg, dim, b, seed, gpu_device):
    import torch
    from hipporag_amd import synth
    rng = np.random.default_rng(seed)
    qf = torch.from_numpy(synth.make_embeddings_np(b, dim, seed + 5).view(np.int16)).to(gpu_device).view(torch.bfloat16)
    qp = torch.from_numpy(synth.make_embeddings_np(b, dim, seed + 6).view(np.int16)).to(gpu_device).view(torch.bfloat16)
    idx, sc = eng.score_facts(qf, k=5)
    cnt = torch.from_numpy(rng.integers(0, 6, size=b).astype(np.int32)).to(gpu_device)   # some rows keep nothing
    return qp, idx, sc, cnt


@pytest.mark.parametrize("v,e,b,iters,power_law,no_fp8", [
    (9000, 90000, 64, 20, True, False),     # long rows cut into segments, one full slab
    (9000, 90000, 20, 16, True, False),     # ragged slab, shortest plan of the two-stage scheme
    (30000, 300000, 33, 21, False, False),  # odd sweep count: h ends in the other buffer
    (4000, 30000, 130, 20, True, True),     # three slabs (fp8 path off): one arrival counter per slab
])
def test_single_launch_ppr_is_bit_identical_to_one_launch_per_sweep(gpu_device, v, e, b, iters, power_law, no_fp8):
    import torch
    from hipporag_amd.engine import HippoRAGEngine
    from hipporag_amd._lib import OPT_NO_FP8, OPT_NO_PERSIST
    dim = 64
    kg, pass_bits, fact_bits, _ = make_case(v, e, dim, seed=77 + b, power_law=power_law)
    eng = HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex,
                         kg.num_chunks, max_batch=b, max_topk=200, flags=OPT_NO_FP8 if no_fp8 else 0)
    qp, idx, sc, cnt = _inputs(eng, kg, dim, b, 3, gpu_device)
    outs = []
    for persist in (True, False, True):          # the third run: buffers left behind by the other path
        eng.set_flags(OPT_NO_PERSIST, not persist)
        # want_residual: the last sweep also measures the relative update at the passages (convergence contract)
        o = eng.retrieve(qp, idx, sc, cnt, ppr_iters=iters, k=200, ppr_tol=1e-6, ppr_max_iters=iters)
        torch.cuda.synchronize()
        t = eng.timings()
        assert t["slab_width"] == 64 and t["ppr_single_launch"] == (1 if persist else 0), t
        outs.append((o.doc_idx.cpu().numpy(), o.doc_score.cpu().numpy(), o.flags.cpu().numpy(),
                     o.residual.cpu().numpy(), eng.last_doc_scores(b).cpu().numpy()))
    eng.close()
    for other in outs[1:]:
        for a, c in zip(outs[0], other):
            assert np.array_equal(a.view(np.int32) if a.dtype == np.float32 else a,
                                  c.view(np.int32) if c.dtype == np.float32 else c)
    assert np.isfinite(outs[0][1]).all() and (outs[0][1] > 0).any()


def test_single_launch_ppr_is_not_taken_on_large_graphs(gpu_device):
    """A graph whose sweep keeps every resident wavefront busy for many chunks gains nothing from the barrier form."""
    import torch
    from hipporag_amd.engine import HippoRAGEngine
    dim = 64
    kg, pass_bits, fact_bits, _ = make_case(400000, 1600000, dim, seed=5)
    eng = HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex,
                         kg.num_chunks, max_batch=64, max_topk=50)
    qp, idx, sc, cnt = _inputs(eng, kg, dim, 64, 9, gpu_device)
    eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=50)
    torch.cuda.synchronize()
    assert eng.timings()["ppr_single_launch"] == 0
    eng.close()
