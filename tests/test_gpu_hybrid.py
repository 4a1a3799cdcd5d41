"""The hybrid multi-GPU mode (hipporag_amd/dist.py HybridRetriever): embeddings row-sharded, one
all-to-all of passage-score rows, PPR query-parallel on a replicated graph through hrag_retrieve_scored -- with the
ranks emulated as threads on ONE device (dist.LocalComm).  Every score is the same MFMA chain whatever slice of the
matrix it is computed from, so every rank's result must be BIT-IDENTICAL to the single-GPU engine on its queries."""

import numpy as np
import pytest

import oracle
from hipporag_amd import synth
from hipporag_amd.graph import bf16_bits_to_float
from tests.helpers import make_case, tie_aware_equal

pytestmark = pytest.mark.gpu


def _bf16(bits, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).to(device).view(torch.bfloat16)


@pytest.mark.parametrize("world,b", [(4, 280), (2, 24), (8, 64)])
def test_hybrid_ranks_are_bit_identical_to_the_single_gpu_engine(gpu_device, world, b):
    import torch
    from hipporag_amd import dist as hd
    from hipporag_amd.engine import HippoRAGEngine
    kg, pass_bits, fact_bits, index = make_case(9000, 90000, 128, seed=700 + world)
    sidx = hd.shard_index(kg.csr, kg.passage_vertex, world, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=3)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=4)
    qf_t, qp_t = _bf16(qf_bits, gpu_device), _bf16(qp_bits, gpu_device)
    kw = dict(link_top_k=5, damping=0.5, passage_node_weight=0.05, ppr_iters=20, k=100)
    arrays = dict(csr=kg.csr, passage_vertex=kg.passage_vertex, subj_vertex=kg.subj_vertex, obj_vertex=kg.obj_vertex,
                  num_chunks=kg.num_chunks)
    f_idx, f_sc, d_idx, d_sc, flags = hd.run_local_hybrid(world, arrays, sidx, pass_bits, fact_bits, qf_t, qp_t, kw,
                                                          gpu_device, 100)
    assert np.all(flags == 0)
    bn = b // world
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex,
                        kg.num_chunks, max_batch=b, max_topk=100) as eng:
        idx, sc = eng.score_facts(qf_t, k=5)
        np.testing.assert_array_equal(f_idx, idx.cpu().numpy())
        np.testing.assert_array_equal(f_sc, sc.cpu().numpy())
        cnt = torch.full((b,), 5, dtype=torch.int32, device=gpu_device)
        for r in range(world):          # the single-GPU engine on rank r's queries: the same batch size, the same PPR path
            sl = slice(r * bn, (r + 1) * bn)
            one = eng.retrieve(qp_t[sl], idx[sl], sc[sl], cnt[sl], **kw)
            np.testing.assert_array_equal(d_idx[sl], one.doc_idx.cpu().numpy())
            np.testing.assert_array_equal(d_sc[sl], one.doc_score.cpu().numpy())
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    for q in range(0, b, max(1, b // 9)):
        ref = oracle.retrieve_one(index, qf[q], qp[q])
        assert tie_aware_equal(d_idx[q], ref.sorted_doc_ids[:100], ref.sorted_doc_scores[:100], rel_gap=2e-5), q
        want = ref.x[kg.passage_vertex][d_idx[q]]
        assert float((np.abs(d_sc[q] - want) / want).max()) < 3e-6, q


def test_retrieve_scored_equals_retrieve(gpu_device):
    """hrag_retrieve_scored on the scores hrag_sim_scores returns == hrag_retrieve, on an engine WITHOUT embeddings."""
    import torch
    from hipporag_amd import dist as hd
    from hipporag_amd.engine import HippoRAGEngine
    kg, pass_bits, fact_bits, _ = make_case(5000, 50000, 64, seed=41)
    b = 33
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=3)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=4)
    kw = dict(link_top_k=5, damping=0.5, passage_node_weight=0.05, ppr_iters=20, k=50)
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex,
                        kg.num_chunks, max_batch=b, max_topk=50) as eng, \
            hd.build_ppr_engine(kg.csr, kg.passage_vertex, kg.subj_vertex, kg.obj_vertex, kg.num_chunks, 64, b, 50) as ppr:
        idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
        cnt = torch.full((b,), 5, dtype=torch.int32, device=gpu_device)
        cnt[3] = 0                                                  # one DPR-fallback row
        want = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, **kw)
        scores = eng.sim_scores("passages", _bf16(qp_bits, gpu_device))
        got = ppr.retrieve_scored(scores, idx, sc, cnt, **kw)
        torch.cuda.synchronize()
        for a, w in ((got.doc_idx, want.doc_idx), (got.doc_score, want.doc_score), (got.flags, want.flags)):
            np.testing.assert_array_equal(a.cpu().numpy(), w.cpu().numpy())
        with pytest.raises(Exception):
            ppr.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, **kw)     # no embeddings: must refuse, not crash
