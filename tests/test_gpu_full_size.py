"""BASELINE configs[3] and configs[4] in front of the driver (round-4 review, next #2): the two configurations whose
sizes the `-m gpu` suite had only met on 6 - 12 k-vertex graphs.

* configs[3]: the full 1M-node / 10M-edge index as 8 row shards with the GLOBAL batch of 1024 queries, all 8 shard
  engines emulated on the one device at hand (dist.LocalComm: the exchange is a barrier on shared state buffers, SURVEY
  8(e)'s emulated gather): bit-identity with the unsharded engine on every query + the fp64 oracle on 8 queries.
  It is `bench.py --config cfg4local` (bench.local_shards_parity) with assertions.
* configs[4]: one GPU's share at a size the suite can afford -- power-law KG, fp16 embeddings of dimension 1024, int32
  CSR, 512 queries (two exchange groups of the fp8 state, long-row segments of the hub rows) at 2M vertices / 20M
  edges instead of 10M / 100M -- against the oracle on 4 queries.

Reference call sites: the per-query loop HippoRAG.py:459 (no sharding upstream), run_ppr HippoRAG.py:1736-1749."""

import numpy as np
import pytest

import oracle
from tests.helpers import ranked_parity, write_test_report

pytestmark = pytest.mark.gpu


def test_configs3_full_index_as_8_emulated_row_shards_global_batch_1024(gpu_device):
    import bench
    cfg = bench.CONFIGS["cfg4local"]
    assert (cfg["V"], cfg["E"], cfg["B"], cfg["shard_of"]) == (1_000_000, 10_000_000, 1024, 8)
    res = bench.local_shards_parity(cfg, 0, 8, 2, gpu_device)
    write_test_report("cfg4_full_size_8_emulated_shards", {k: res[k] for k in (
        "parity_vs_oracle", "bit_identical_to_single_gpu_engine_on_relabelled_index", "ms_per_step")})
    bit = res["bit_identical_to_single_gpu_engine_on_relabelled_index"]
    assert bit["fact_ids"] and bit["fact_scores"] and bit["doc_ids"] and bit["doc_scores"], bit
    par = res["parity_vs_oracle"]
    assert par["queries_checked"] == 8 and par["flags_or"] == 0, par
    assert par["topk_ids_equal"] and par["max_rel_score_err"] < 1e-5, par
    assert res["config"]["global_batch"] == 1024 and res["config"]["shards"] == 8


def test_configs4_share_shape_power_law_fp16_dim1024_two_groups(gpu_device):
    import torch
    from hipporag_amd import synth
    from hipporag_amd.engine import HippoRAGEngine
    V, E, D, B, seed, K = 2_000_000, 20_000_000, 1024, 512, 1239, 200
    kg = synth.make_kg(V, E, seed, power_law=True)
    assert kg.csr.col_idx.dtype == np.int32 and kg.csr.row_ptr.dtype == np.int32
    deg = np.diff(kg.csr.row_ptr)
    assert deg.max() > 20_000                                   # hub rows: cut into segments, combined in the sweep
    pass_emb = synth.make_embeddings_torch(kg.n_passages, D, seed + 1, gpu_device, dtype=torch.float16)
    fact_emb = synth.make_embeddings_torch(kg.n_facts, D, seed + 2, gpu_device, dtype=torch.float16)
    qf = synth.make_queries_torch(fact_emb, B, seed + 100)[0]
    qp = synth.make_queries_torch(pass_emb, B, seed + 500)[0]
    cnt = torch.full((B,), 5, dtype=torch.int32, device=gpu_device)
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_emb, fact_emb, kg.subj_vertex, kg.obj_vertex,
                        kg.num_chunks, max_batch=B, max_topk=K) as eng:
        lay = eng.shard_layout(B, 0)
        assert lay.n_slabs == 4 and lay.n_groups == 2           # what hrag_retrieve lays the fp8 state out as
        idx, sc = eng.score_facts(qf, k=5)
        out = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=K)
        out2 = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=K)
        torch.cuda.synchronize()
        tm = eng.timings()
        assert tm["slab_width"] == 128 and tm["n_long_rows"] > 0
    ids, scores = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
    assert torch.equal(out.doc_idx, out2.doc_idx) and torch.equal(out.doc_score, out2.doc_score)
    assert np.all(out.flags.cpu().numpy() == 0)
    for q in range(B):
        d = np.diff(scores[q])
        assert len(np.unique(ids[q])) == K and np.all(d <= 0) and np.all(ids[q][1:][d == 0] < ids[q][:-1][d == 0])
    a = oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight)
    index = oracle.RefIndex(fact_emb=fact_emb.float().cpu().numpy(), passage_emb=pass_emb.float().cpu().numpy(),
                            subj_vertex=kg.subj_vertex, obj_vertex=kg.obj_vertex, num_chunks=kg.num_chunks,
                            passage_vertex=kg.passage_vertex, p=oracle.column_normalize(a))
    qf_h, qp_h = qf.float().cpu().numpy(), qp.float().cpu().numpy()
    fact_ids = idx.cpu().numpy()
    worst, gap, exact, npos = 0.0, 0.0, 0, 0
    for q in (0, 170, 341, 511):                                # both exchange groups, all four slabs
        ref = oracle.retrieve_one(index, qf_h[q], qp_h[q])
        np.testing.assert_array_equal(fact_ids[q], ref.fact_candidates)
        rep = ranked_parity(ids[q], scores[q], ref.sorted_doc_ids, ref.sorted_doc_scores, ref.x[kg.passage_vertex])
        assert rep["equal"], (q, rep)
        worst, gap = max(worst, rep["worst_rel_err"]), max(gap, rep["rel_gap"])
        exact += rep["exact_positions"]; npos += rep["n"]
    write_test_report("cfg5_share_shape_parity", {"V": V, "E": E, "dim": D, "batch": B, "queries": 4, "ranks_per_query": K,
                                                  "max_rel_score_err": worst, "exact_id_fraction": exact / npos,
                                                  "tie_window_rel": gap, "n_long_rows": int(tm["n_long_rows"])})
    assert worst < 1e-5, worst
