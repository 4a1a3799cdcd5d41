"""The graph compiler's locality numbering (graph.locality_order, HippoRAGEngine(locality=...)): the engine renumbers
the vertices by the first passage that links them and sweeps in SELL-C-sigma windows with an XCD-blocked launch
(hrag_opts.sell_sigma, HRAG_OPT_XCD_BLOCKED) -- the caller keeps ITS numbering everywhere (passage positions, fact
ids, hrag_ppr's vertex order), and the results stay within the parity bars of the oracle."""

import numpy as np
import pytest

import oracle
from hipporag_amd import synth
from hipporag_amd.graph import bf16_bits_to_float, locality_order, locality_score, relabel_csr
from tests.helpers import tie_aware_equal

pytestmark = pytest.mark.gpu


def _bf16(bits, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).to(device).view(torch.bfloat16)


def _case(seed):
    """a community-structured graph under the reference's hash-order entity numbering"""
    kg = synth.hash_order(synth.make_kg(16000, 160000, seed, community=256), seed + 1)
    pb, fb = synth.make_embeddings_np(kg.n_passages, 64, seed + 2), synth.make_embeddings_np(kg.n_facts, 64, seed + 3)
    a = oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight)
    index = oracle.RefIndex(fact_emb=bf16_bits_to_float(fb), passage_emb=bf16_bits_to_float(pb),
                            subj_vertex=kg.subj_vertex, obj_vertex=kg.obj_vertex, num_chunks=kg.num_chunks,
                            passage_vertex=kg.passage_vertex, p=oracle.column_normalize(a))
    return kg, pb, fb, index


def test_locality_order_recovers_what_the_hash_order_hides():
    kg = synth.make_kg(16000, 160000, 5, community=256)
    hashed = synth.hash_order(kg, 6)
    perm = locality_order(hashed.csr, hashed.passage_vertex)
    assert np.array_equal(np.sort(perm), np.arange(kg.num_vertices))
    assert np.array_equal(perm[hashed.passage_vertex], kg.n_entities + np.arange(kg.n_passages))   # passages keep their order
    again = relabel_csr(hashed.csr, perm)
    assert locality_score(hashed.csr, 512) < 0.1 < 0.5 < locality_score(again, 512)
    np.testing.assert_allclose(np.sort(again.val), np.sort(hashed.csr.val))


@pytest.mark.parametrize("b", [5, 40, 130])
def test_engine_with_locality_numbering_matches_the_oracle(gpu_device, b):
    import torch
    from hipporag_amd._lib import OPT_XCD_BLOCKED
    from hipporag_amd.engine import HippoRAGEngine
    kg, pb, fb, index = _case(31)
    qf_bits, _ = synth.make_queries_np(fb, b, seed=3)
    qp_bits, _ = synth.make_queries_np(pb, b, seed=4)
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pb, fb, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                        max_batch=b, max_topk=100, locality="auto") as eng:
        assert eng.locality_score > 0.3 and (eng.opt_flags & OPT_XCD_BLOCKED)     # the numbering found the locality
        idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
        out = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, torch.full((b,), 5, dtype=torch.int32, device=gpu_device),
                           ppr_iters=20, k=100)
        torch.cuda.synchronize()
        d_idx, d_sc = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
        # run_ppr seam: V-length input and output in the CALLER's vertex order
        q0 = oracle.retrieve_one(index, bf16_bits_to_float(qf_bits)[0], bf16_bits_to_float(qp_bits)[0])
        x, _ = eng.ppr(torch.from_numpy(q0.reset.astype(np.float32))[None], 0.5, 40)
        np.testing.assert_allclose(x[0].cpu().numpy(), q0.x, rtol=2e-5, atol=1e-12)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    for q in range(0, b, max(1, b // 8)):
        ref = oracle.retrieve_one(index, qf[q], qp[q])
        assert tie_aware_equal(d_idx[q], ref.sorted_doc_ids[:100], ref.sorted_doc_scores[:100], rel_gap=2e-5), q
        want = ref.x[kg.passage_vertex][d_idx[q]]
        assert float((np.abs(d_sc[q] - want) / want).max()) < 1e-5, q


def test_degree_order_puts_the_hubs_first_and_keeps_the_passages_in_order():
    from hipporag_amd.graph import degree_order
    kg = synth.make_kg(8000, 80000, 9, power_law=True)
    perm = degree_order(kg.csr, kg.passage_vertex)
    assert np.array_equal(np.sort(perm), np.arange(kg.num_vertices))
    assert np.array_equal(perm[kg.passage_vertex], kg.n_entities + np.arange(kg.n_passages))
    g = relabel_csr(kg.csr, perm)
    deg = np.diff(g.row_ptr)
    assert np.all(np.diff(deg[:kg.n_entities]) <= 0)                      # entities by falling entry count
    np.testing.assert_allclose(np.sort(g.val), np.sort(kg.csr.val))
    # the score of a candidate numbering without building the relabelled matrix
    assert locality_score(kg.csr, 512, perm=perm) == locality_score(g, 512)


@pytest.mark.gpu
@pytest.mark.parametrize("b", [3, 40, 130])
def test_hub_first_numbering_matches_the_oracle(gpu_device, b):
    """HippoRAGEngine(locality="degree"): vertices renumbered hub-first (graph.degree_order; an experiment switch), the
    window / XCD switches left alone, and everything the caller sees stays in the caller's numbering."""
    import torch
    from hipporag_amd._lib import OPT_XCD_BLOCKED
    from hipporag_amd.engine import HippoRAGEngine
    from tests.helpers import make_case
    kg, pb, fb, index = make_case(9000, 90000, 64, seed=17, power_law=(b == 40))
    qf_bits, _ = synth.make_queries_np(fb, b, seed=3)
    qp_bits, _ = synth.make_queries_np(pb, b, seed=4)
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pb, fb, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                        max_batch=b, max_topk=100, locality="degree") as eng:
        assert eng.numbering == "degree" and not (eng.opt_flags & OPT_XCD_BLOCKED)
        idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
        out = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, torch.full((b,), 5, dtype=torch.int32, device=gpu_device),
                           ppr_iters=20, k=100)
        torch.cuda.synchronize()
        d_idx, d_sc = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
        q0 = oracle.retrieve_one(index, bf16_bits_to_float(qf_bits)[0], bf16_bits_to_float(qp_bits)[0])
        x, _ = eng.ppr(torch.from_numpy(q0.reset.astype(np.float32))[None], 0.5, 40)
        np.testing.assert_allclose(x[0].cpu().numpy(), q0.x, rtol=2e-5, atol=1e-12)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    for q in range(0, b, max(1, b // 8)):
        ref = oracle.retrieve_one(index, qf[q], qp[q])
        assert tie_aware_equal(d_idx[q], ref.sorted_doc_ids[:100], ref.sorted_doc_scores[:100], rel_gap=2e-5), q
        want = ref.x[kg.passage_vertex][d_idx[q]]
        assert float((np.abs(d_sc[q] - want) / want).max()) < 1e-5, q
