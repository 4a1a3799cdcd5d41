"""GPU parity tests: the HIP path (through the C ABI, via hipporag_amd.engine) against the CPU
oracle on the same seeded inputs.  Bars (BASELINE.json north_star): ranked ids identical (modulo
documented tie classes), PPR scores within 1e-5 relative."""

import numpy as np
import pytest

import oracle
from hipporag_amd import _lib
from hipporag_amd.graph import bf16_bits_to_float
from tests.helpers import make_case, ranked_parity, tie_aware_equal, write_test_report

pytestmark = pytest.mark.gpu


def _t(x, device, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x)).to(device)
    return t if dtype is None else t.to(dtype)


def _bf16(bits, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).to(device).view(torch.bfloat16)


@pytest.fixture(scope="module")
def case(gpu_device):
    from hipporag_amd.engine import HippoRAGEngine
    from hipporag_amd import synth
    kg, pass_bits, fact_bits, index = make_case(6000, 60000, 192, seed=21, power_law=True)
    eng = HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex,
                         kg.num_chunks, max_batch=80, max_topk=200, long_row_nnz=16, segment_nnz=64)
    qf_bits, _ = synth.make_queries_np(fact_bits, 70, seed=5)
    qp_bits, _ = synth.make_queries_np(pass_bits, 70, seed=6)
    yield dict(kg=kg, index=index, eng=eng, qf_bits=qf_bits, qp_bits=qp_bits,
               pass_bits=pass_bits, fact_bits=fact_bits)
    eng.close()


# ----------------------------------------------------------------------------- K1 similarity
@pytest.mark.parametrize("rows,dim,batch", [(1000, 768, 1), (333, 64, 5), (4100, 200, 64), (777, 768, 130),
                                            (129, 1024, 17),
                                            # batch <= 2: the GEMV kernel (csrc/sim_gemv.hip), every chunk count
                                            (1001, 4096, 2), (517, 1536, 2), (5, 8, 2), (2000, 768, 2),
                                            (900, 512, 1), (64, 2048, 1), (300, 3000, 1), (2000, 768, 8)])
def test_sim_scores_match_fp64_dot(gpu_device, rows, dim, batch):
    from hipporag_amd.engine import HippoRAGEngine
    from hipporag_amd import synth
    from hipporag_amd.graph import build_csr
    emb = synth.make_embeddings_np(rows, dim, seed=rows)
    q = synth.make_embeddings_np(batch, dim, seed=rows + 1)      # asymmetric, unrelated rows
    g = build_csr(4, [0, 1], [1, 2], [1.0, 1.0])
    with HippoRAGEngine(g, np.full(rows, 3, np.int32), emb, max_batch=batch) as eng:
        got = eng.sim_scores("passages", _bf16(q, gpu_device)).cpu().numpy()
    want = bf16_bits_to_float(q).astype(np.float64) @ bf16_bits_to_float(emb).astype(np.float64).T
    assert got.shape == (batch, rows)
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-6)


# ----------------------------------------------------------------------------- K4 top-k
def _check_topk(gpu_device, scores, k, **kw):
    from hipporag_amd.engine import topk_rows
    idx, val, mn, mx = topk_rows(_t(scores, gpu_device), k, want_minmax=True, **kw)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    n = kw.get("n", scores.shape[1])
    for r in range(scores.shape[0]):
        row = scores[r, :n] + np.float32(0.0)
        want = oracle.topk_desc(row, k)
        m = len(want)
        np.testing.assert_array_equal(idx[r, :m], want)
        np.testing.assert_array_equal(val[r, :m], row[want])
        assert np.all(idx[r, m:] == -1)
        assert mn.cpu().numpy()[r] == row.min() and mx.cpu().numpy()[r] == row.max()


@pytest.mark.parametrize("n,k", [(3, 5), (100, 5), (5000, 200), (300001, 200), (4099, 2048), (1, 1),
                                 (300001, 2047), (875000, 2047), (70000, 2048)])   # large k on long rows: the radix refinement
def test_topk_rows_exact(gpu_device, n, k):
    rng = np.random.default_rng(n + k)
    s = rng.standard_normal((4, n)).astype(np.float32)
    s[1] = np.round(s[1] * 4) / 4           # heavy ties
    s[2] = 0.25                             # all equal -> index-descending
    if n > 2:
        s[3, :2] = [0.0, -0.0]              # the two zeros tie
        s[3] = np.sort(s[3])                # ascending input
    _check_topk(gpu_device, s, k)


def test_topk_rows_adversarial_overflow_path(gpu_device):
    # every large value lives in the slice one thread scans => thread-local maxima give a useless
    # lower bound => candidate buffer overflows => the refinement path must still be exact
    n, k = 1 << 18, 200
    rng = np.random.default_rng(0)
    s = rng.random((2, n)).astype(np.float32)
    hot = np.arange(0, n, 4096)             # all inside thread 0's float4 stride
    s[0, hot] += 10.0
    s[0, hot + 1] += 10.0
    _check_topk(gpu_device, s, k)
    _check_topk(gpu_device, s[:, : n - 3], k, n=n - 7)      # unaligned row start / ragged n


def test_topk_normalize_matches_min_max(gpu_device):
    from hipporag_amd.engine import topk_rows
    rng = np.random.default_rng(1)
    s = rng.standard_normal((3, 9000)).astype(np.float32)
    s[2] = 1.5                                              # range 0 -> ones
    idx, val = topk_rows(_t(s, gpu_device), 5, normalize=True)
    for r in range(3):
        norm = oracle.min_max_normalize(s[r])
        want = oracle.topk_desc(norm, 5) if r < 2 else oracle.topk_desc(s[r], 5)
        np.testing.assert_array_equal(idx.cpu().numpy()[r], want)
        np.testing.assert_array_equal(val.cpu().numpy()[r], norm[want])


# ----------------------------------------------------------------------------- K3 PPR
@pytest.mark.parametrize("batch", [1, 3, 7, 33, 70])
def test_ppr_matches_exact_solution(case, gpu_device, batch):
    kg, index, eng = case["kg"], case["index"], case["eng"]
    rng = np.random.default_rng(batch)
    v = kg.num_vertices
    reset = np.zeros((batch, v), dtype=np.float32)
    for b in range(batch):
        reset[b, kg.passage_vertex] = 0.05 * rng.random(kg.n_passages, dtype=np.float32)
        seeds = rng.integers(0, kg.n_entities, 5)
        reset[b, seeds] += rng.random(5, dtype=np.float32)
    reset[0, 11] = np.nan
    reset[0, 12] = -3.0                                     # sanitised like HippoRAG.py:1735
    x, flags = eng.ppr(_t(reset, gpu_device), damping=0.5, iters=30)
    x = x.cpu().numpy()
    assert np.all(flags.cpu().numpy() == 0)
    for b in range(batch):
        want = oracle.ppr_exact(index.p, reset[b].astype(np.float64), 0.5)
        assert abs(x[b].sum(dtype=np.float64) - 1.0) < 1e-5
        big = want > 1e-9
        rel = np.abs(x[b][big] - want[big]) / want[big]
        assert rel.max() < 1e-5, (b, rel.max())
        assert np.all(x[b][~big] <= 2e-9)


def test_ppr_twenty_sweeps_and_zero_mass(case, gpu_device):
    kg, index, eng = case["kg"], case["index"], case["eng"]
    rng = np.random.default_rng(3)
    reset = np.zeros((2, kg.num_vertices), dtype=np.float32)
    reset[0, kg.passage_vertex] = 0.05 * rng.random(kg.n_passages, dtype=np.float32)
    reset[0, 17] = 1.0
    x, flags = eng.ppr(_t(reset, gpu_device), damping=0.5, iters=20)
    assert flags.cpu().numpy().tolist() == [0, 2]           # row 1 has no mass
    assert np.all(x.cpu().numpy()[1] == 0)
    want = oracle.ppr_exact(index.p, reset[0].astype(np.float64), 0.5)
    pv = kg.passage_vertex
    got = x.cpu().numpy()[0][pv]
    nz = want[pv] > 0
    assert (np.abs(got[nz] - want[pv][nz]) / want[pv][nz]).max() < 1e-5
    # same arithmetic as the oracle's fp32 leaky iteration up to summation order
    p32 = oracle.ppr_power(index.p, reset[0], 0.5, iters=20, dtype=np.float32)
    np.testing.assert_allclose(x.cpu().numpy()[0], p32, rtol=2e-5, atol=1e-12)


# ----------------------------------------------------------------------------- phase A / B
def test_score_facts_matches_oracle(case, gpu_device):
    index, eng = case["index"], case["eng"]
    qf = bf16_bits_to_float(case["qf_bits"])
    idx, sc = eng.score_facts(_bf16(case["qf_bits"], gpu_device), k=5)
    idx, sc = idx.cpu().numpy(), sc.cpu().numpy()
    for b in range(qf.shape[0]):
        s = oracle.fact_scores(index.fact_emb, qf[b])
        want = oracle.topk_desc(s, 5)
        assert tie_aware_equal(idx[b], want, s[want], abs_gap=2e-6)
        np.testing.assert_allclose(sc[b], s[idx[b]], rtol=0, atol=2e-6)


@pytest.mark.parametrize("rows,dim,b,k", [(6000, 192, 70, 5), (1000, 768, 130, 16), (129, 72, 65, 5), (100, 64, 200, 16),
                                          (40000, 256, 256, 5), (7, 64, 66, 5),
                                          (6000, 192, 40, 5), (1000, 768, 17, 16), (300, 64, 64, 5),   # 64-query tiles
                                          (1_100_000, 32, 66, 5),        # > 8192 tiles: the select rounds re-read memory
                                          (20000, 128, 1024, 16), (9000, 64, 300, 16)])   # pass 3 in tile order (>= 4 pairs per tile)
def test_score_facts_fused_is_bit_identical_to_gemm_plus_topk(gpu_device, rows, dim, b, k):
    """Batches > 16 take the fused path (tile maxima -> k tiles -> recomputed scores, csrc/sim_gemm.hip)
    without the [B, F] score matrix: ids and normalised scores must equal the two-step path
    (hrag_sim_scores + hrag_topk_rows) bit for bit, including duplicated rows (ties -> larger index),
    a last partial tile and fewer rows than k."""
    import torch
    from hipporag_amd.engine import HippoRAGEngine, topk_rows
    from hipporag_amd.graph import build_csr
    from hipporag_amd import synth
    emb = synth.make_embeddings_np(rows, dim, seed=rows + 3)
    if rows > 300:
        emb[200:260] = emb[100:160]                      # exact ties across two tiles
        emb[-1] = emb[5]
    q, _ = synth.make_queries_np(emb, b, seed=9)
    g = build_csr(4, [0, 1], [1, 2], [1.0, 1.0])
    zeros = np.zeros(rows, np.int32)
    with HippoRAGEngine(g, np.array([3], np.int32), emb[:1], emb, zeros, zeros, np.zeros(4, np.int32),
                        max_batch=b, max_topk=16) as eng:
        idx, sc = eng.score_facts(_bf16(q, gpu_device), k=k)
        raw = eng.sim_scores("facts", _bf16(q, gpu_device))
        idx2, sc2 = topk_rows(raw, k, normalize=True)
        torch.cuda.synchronize()
    assert torch.equal(idx, idx2)
    assert torch.equal(sc, sc2)
    if rows < k:
        assert (idx[:, rows:] == -1).all()


@pytest.mark.parametrize("b", [2, 12, 40, 70])
def test_fp16_embeddings_end_to_end(gpu_device, b):
    """BASELINE configs[4] names fp16 embeddings: the same kernels on IEEE binary16 elements
    (v_mfma_f32_16x16x32_f16 / v_dot2c_f32_f16).  Raw scores against an fp64 dot of the fp16-rounded
    inputs, then phase A + B against the oracle built from those inputs: B = 2 takes the GEMV,
    12 / 40 the MFMA GEMM tiles of 16 / 64 queries, 70 the fused top-k + the fp8-state PPR."""
    import torch
    from hipporag_amd.engine import HippoRAGEngine
    from hipporag_amd import synth
    kg, pass_bits, fact_bits, index = make_case(5000, 50000, 96, seed=61)
    pass16 = bf16_bits_to_float(pass_bits).astype(np.float16)
    fact16 = bf16_bits_to_float(fact_bits).astype(np.float16)
    index.passage_emb, index.fact_emb = pass16.astype(np.float32), fact16.astype(np.float32)
    rng = np.random.default_rng(b)
    qf = (fact16[rng.integers(0, len(fact16), b)].astype(np.float32) + 0.05 * rng.standard_normal((b, 96))).astype(np.float16)
    qp = (pass16[rng.integers(0, len(pass16), b)].astype(np.float32) + 0.05 * rng.standard_normal((b, 96))).astype(np.float16)
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pass16, fact16, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                        max_batch=b, max_topk=50) as eng:
        tq = torch.from_numpy(qf).to(gpu_device)
        raw = eng.sim_scores("facts", tq).cpu().numpy()
        idx, sc = eng.score_facts(tq, k=5)
        cnt = _t(np.full(b, 5, np.int32), gpu_device)
        out = eng.retrieve(torch.from_numpy(qp).to(gpu_device), idx, sc, cnt, ppr_iters=20, k=50)
        got_idx, got_sc = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
        idx_h = idx.cpu().numpy()
    want = qf.astype(np.float64) @ fact16.astype(np.float64).T
    np.testing.assert_allclose(raw, want, rtol=0, atol=3e-6)
    for q in range(b):
        ref = oracle.retrieve_one(index, qf[q].astype(np.float32), qp[q].astype(np.float32))
        assert tie_aware_equal(idx_h[q], np.array(ref.fact_candidates), ref.fact_candidate_scores, abs_gap=2e-6), q
        assert tie_aware_equal(got_idx[q], ref.sorted_doc_ids[:50], ref.sorted_doc_scores[:50], rel_gap=2e-5), q
        w = ref.x[kg.passage_vertex][got_idx[q]]
        assert (np.abs(got_sc[q] - w) / w).max() < 1e-5, q


def _oracle_batch(case, kept_lists):
    index = case["index"]
    qf = bf16_bits_to_float(case["qf_bits"])
    qp = bf16_bits_to_float(case["qp_bits"])
    out = []
    for b, kept in enumerate(kept_lists):
        flt = (lambda cand, kept=kept: kept) if kept is not None else None
        out.append(oracle.retrieve_one(index, qf[b], qp[b], filter_fn=flt))
    return out


def test_retrieve_end_to_end_identity_filter(case, gpu_device):
    import torch
    eng = case["eng"]
    b = case["qf_bits"].shape[0]
    idx, sc = eng.score_facts(_bf16(case["qf_bits"], gpu_device), k=5)
    cnt = torch.full((b,), 5, dtype=torch.int32, device=gpu_device)
    out = eng.retrieve(_bf16(case["qp_bits"], gpu_device), idx, sc, cnt, ppr_iters=25, k=200)
    refs = _oracle_batch(case, [None] * b)
    got_idx, got_sc = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
    assert np.all(out.flags.cpu().numpy() == 0)
    for q in range(b):
        ref = refs[q]
        np.testing.assert_array_equal(idx.cpu().numpy()[q], ref.fact_candidates)
        want_ids, want_sc = ref.sorted_doc_ids[:200], ref.sorted_doc_scores[:200]
        assert tie_aware_equal(got_idx[q], want_ids, want_sc, rel_gap=2e-5), q
        np.testing.assert_allclose(got_sc[q], ref.x[case["kg"].passage_vertex][got_idx[q]], rtol=1e-5, atol=0)


def test_retrieve_filter_subsets_and_dpr_fallback(case, gpu_device):
    import torch
    eng, index = case["eng"], case["index"]
    b = 12
    qf_bits, qp_bits = case["qf_bits"][:b], case["qp_bits"][:b]
    sub = dict(case, qf_bits=qf_bits, qp_bits=qp_bits)
    idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
    idx_h, sc_h = idx.cpu().numpy(), sc.cpu().numpy()
    # the "LLM filter": keep a subset, in its own order; query 3 and 7 keep nothing
    plans = [[0, 1, 2, 3, 4], [4, 0], [2], [], [1, 3], [0], [3, 2, 1], [], [0, 4], [4, 3, 2, 1, 0], [1], [0, 2]]
    kept_idx = np.full((b, 5), -1, np.int32)
    kept_sc = np.zeros((b, 5), np.float32)
    kept_cnt = np.zeros(b, np.int32)
    kept_lists = []
    for q, plan in enumerate(plans):
        kept_idx[q, :len(plan)] = idx_h[q, plan]
        kept_sc[q, :len(plan)] = sc_h[q, plan]
        kept_cnt[q] = len(plan)
        kept_lists.append(idx_h[q, plan].tolist())
    out = eng.retrieve(_bf16(qp_bits, gpu_device), _t(kept_idx, gpu_device), _t(kept_sc, gpu_device),
                       _t(kept_cnt, gpu_device), ppr_iters=25, k=150)
    refs = _oracle_batch(sub, kept_lists)
    flags = out.flags.cpu().numpy()
    for q in range(b):
        ref = refs[q]
        assert bool(flags[q] & 1) == ref.used_dpr == (len(plans[q]) == 0)
        want_ids, want_sc = ref.sorted_doc_ids[:150], ref.sorted_doc_scores[:150]
        got = out.doc_idx.cpu().numpy()[q]
        if ref.used_dpr:                     # normalised DPR scores, HippoRAG.py:467-469
            assert tie_aware_equal(got, want_ids, want_sc, abs_gap=3e-6), q
            np.testing.assert_allclose(out.doc_score.cpu().numpy()[q], want_sc, rtol=0, atol=3e-6)
        else:
            assert tie_aware_equal(got, want_ids, want_sc, rel_gap=2e-5), q
            np.testing.assert_allclose(out.doc_score.cpu().numpy()[q],
                                       ref.x[case["kg"].passage_vertex][got], rtol=1e-5, atol=0)


def test_dense_retrieve_matches_oracle(case, gpu_device):
    eng, index = case["eng"], case["index"]
    qp = bf16_bits_to_float(case["qp_bits"][:9])
    idx, sc = eng.dense_retrieve(_bf16(case["qp_bits"][:9], gpu_device), k=50)
    for q in range(9):
        ids, scores = oracle.retrieve_dpr_one(index, qp[q])
        assert tie_aware_equal(idx.cpu().numpy()[q], ids[:50], scores[:50], abs_gap=3e-6)
        np.testing.assert_allclose(sc.cpu().numpy()[q], scores[:50], rtol=0, atol=3e-6)


def test_seed_edge_cases(gpu_device):
    """Phrase shared by several kept facts (mean, HippoRAG.py:1608), num_chunks divisor (:1600),
    absent phrases (-1), subject == object, and the :1541 assert condition (flag bit 2)."""
    from hipporag_amd.engine import HippoRAGEngine
    from hipporag_amd.graph import build_csr
    from hipporag_amd import synth
    v, n_p = 12, 4
    src = [0, 1, 2, 3, 4, 8, 9, 10, 11, 5]
    dst = [1, 2, 3, 4, 0, 0, 1, 2, 3, 6]
    g = build_csr(v, src, dst, np.ones(len(src)))
    pv = np.array([8, 9, 10, 11], np.int32)
    subj = np.array([0, 0, 1, 5, 7, 2, 3], np.int32)
    obj = np.array([1, 2, -1, 5, 0, 3, 4], np.int32)
    nchunks = np.array([2, 1, 3, 0, 1, 1, 1, 4, 0, 0, 0, 0], np.int32)
    pe = synth.make_embeddings_np(n_p, 64, 1)
    fe = synth.make_embeddings_np(len(subj), 64, 2)
    a = oracle.build_symmetric_csr(v, src, dst, np.ones(len(src)))
    index = oracle.RefIndex(bf16_bits_to_float(fe), bf16_bits_to_float(pe), subj, obj, nchunks, pv,
                            oracle.column_normalize(a))
    scores = np.array([0.9, 0.5, 0.7, 0.3, 0.2, 0.0, 0.6], np.float32)
    plans = [[0, 1, 2], [3, 4], [5], [0, 1, 2, 4, 6], [2]]
    b = len(plans)
    kept_idx = np.full((b, 5), -1, np.int32)
    kept_sc = np.zeros((b, 5), np.float32)
    cnt = np.zeros(b, np.int32)
    for q, p in enumerate(plans):
        kept_idx[q, :len(p)] = p
        kept_sc[q, :len(p)] = scores[p]
        cnt[q] = len(p)
    qp_bits, _ = synth.make_queries_np(pe, b, 9)
    with HippoRAGEngine(g, pv, pe, fe, subj, obj, nchunks, max_batch=8, max_topk=4) as eng:
        out = eng.retrieve(_bf16(qp_bits, gpu_device), _t(kept_idx, gpu_device), _t(kept_sc, gpu_device),
                           _t(cnt, gpu_device), ppr_iters=40, k=4)
    flags = out.flags.cpu().numpy()
    qp = bf16_bits_to_float(qp_bits)
    for q, p in enumerate(plans):
        try:
            ids, w = oracle.seed_weights(index, scores, p)
            asserted = False
        except AssertionError:
            asserted = True
        assert bool(flags[q] & 4) == asserted, q
        if asserted:
            continue
        dpr_ids, dpr_sc = oracle.dense_passage_scores(index.passage_emb, qp[q])
        by_p = np.empty_like(dpr_sc)
        by_p[dpr_ids] = dpr_sc
        reset = oracle.reset_vector(index, ids, w, by_p)
        want_ids, want_sc, x = oracle.run_ppr(index, reset)
        assert tie_aware_equal(out.doc_idx.cpu().numpy()[q], want_ids[:4], want_sc[:4], rel_gap=2e-5)
        np.testing.assert_allclose(out.doc_score.cpu().numpy()[q], x[pv][out.doc_idx.cpu().numpy()[q]],
                                   rtol=1e-5)


# ----------------------------------------------------------------------------- two-stage fp16 PPR state
def test_retrieve_f8_and_f16_state_paths_vs_f32_state_path_and_oracle(case, gpu_device):
    """hrag_retrieve switches to the staged fp8 state (csrc/ppr8.hip) for batch > 64 and to the
    two-stage fp16 state (csrc/ppr16.hip) for batch > 8 (or with HRAG_OPT_NO_FP8), ppr_iters >= 16.
    Same inputs through (a) the fp8 path, (b) the fp16 path, (c) an engine created with
    HRAG_OPT_F32_STATE, (d) the oracle: all device paths must meet the 1e-5 bar, including filter
    subsets, DPR-fallback rows and queries in different 64-wide slabs seeding the same vertices."""
    import torch
    from hipporag_amd.engine import HippoRAGEngine
    from hipporag_amd._lib import OPT_F32_STATE, OPT_NO_FP8
    kg, eng8 = case["kg"], case["eng"]
    b = 70
    qf_bits, qp_bits = case["qf_bits"][:b].copy(), case["qp_bits"][:b].copy()
    qf_bits[65] = qf_bits[1]            # slab 1 seeds the same entities as slab 0
    qf_bits[66] = qf_bits[2]
    sub = dict(case, qf_bits=qf_bits, qp_bits=qp_bits)
    idx, sc = eng8.score_facts(_bf16(qf_bits, gpu_device), k=5)
    idx_h, sc_h = idx.cpu().numpy(), sc.cpu().numpy()
    rng = np.random.default_rng(4)
    kept_idx = np.full((b, 5), -1, np.int32)
    kept_sc = np.zeros((b, 5), np.float32)
    kept_cnt = np.zeros(b, np.int32)
    kept_lists = []
    for q in range(b):
        n = 5 if q in (1, 2, 65, 66) else int(rng.integers(0, 6))       # some rows keep nothing
        plan = rng.permutation(5)[:n].tolist() if q not in (1, 2, 65, 66) else [0, 1, 2, 3, 4]
        kept_idx[q, :n] = idx_h[q, plan]
        kept_sc[q, :n] = sc_h[q, plan]
        kept_cnt[q] = n
        kept_lists.append(idx_h[q, plan].tolist())
    refs = _oracle_batch(sub, kept_lists)
    eng32 = HippoRAGEngine(kg.csr, kg.passage_vertex, case["pass_bits"], case["fact_bits"], kg.subj_vertex,
                           kg.obj_vertex, kg.num_chunks, max_batch=80, max_topk=200, flags=OPT_F32_STATE)
    eng16 = HippoRAGEngine(kg.csr, kg.passage_vertex, case["pass_bits"], case["fact_bits"], kg.subj_vertex,
                           kg.obj_vertex, kg.num_chunks, max_batch=80, max_topk=200, flags=OPT_NO_FP8)
    outs = {}
    for name, eng in (("f8", eng8), ("f16", eng16), ("f32", eng32)):
        out = eng.retrieve(_bf16(qp_bits, gpu_device), _t(kept_idx, gpu_device), _t(kept_sc, gpu_device),
                           _t(kept_cnt, gpu_device), ppr_iters=20, k=200)
        torch.cuda.synchronize()
        outs[name] = (out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy(), out.flags.cpu().numpy())
    assert eng8.timings()["slab_width"] == 128
    assert eng16.timings()["slab_width"] == 64 and eng32.timings()["slab_width"] == 32
    eng32.close()
    eng16.close()
    pv = kg.passage_vertex
    worst = {"f8": 0.0, "f16": 0.0, "f32": 0.0}
    for name, (got_idx, got_sc, flags) in outs.items():
        for q in range(b):
            ref = refs[q]
            assert bool(flags[q] & 1) == ref.used_dpr == (kept_cnt[q] == 0), (name, q)
            want_ids, want_sc = ref.sorted_doc_ids[:200], ref.sorted_doc_scores[:200]
            if ref.used_dpr:
                assert tie_aware_equal(got_idx[q], want_ids, want_sc, abs_gap=3e-6), (name, q)
                np.testing.assert_allclose(got_sc[q], want_sc, rtol=0, atol=3e-6)
            else:
                assert tie_aware_equal(got_idx[q], want_ids, want_sc, rel_gap=2e-5), (name, q)
                want = ref.x[pv][got_idx[q]]
                rel = np.abs(got_sc[q] - want) / want
                worst[name] = max(worst[name], float(rel.max()))
                assert rel.max() < 1e-5, (name, q, rel.max())
    # the reduced-precision states follow the fp32 trajectory: their error stays within a small factor
    assert worst["f16"] < 5e-6 and worst["f8"] < 5e-6, worst


@pytest.mark.parametrize("b,iters", [(65, 20), (128, 20), (130, 22), (200, 30), (257, 24), (129, 21)])
def test_retrieve_f8_state_batches_and_iteration_counts(gpu_device, b, iters):
    """The staged fp8 path (csrc/ppr8.hip) over partially filled 128-wide slabs (65, 130, 200, 257
    queries), stage plans of different length (20 / 21 / 22 / 24 / 30 sweeps), long rows cut into
    segments, against the exact solution."""
    from hipporag_amd.engine import HippoRAGEngine
    from hipporag_amd import synth
    kg, pass_bits, fact_bits, index = make_case(9000, 90000, 64, seed=31 + b, power_law=True)
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=3)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=4)
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex,
                        kg.num_chunks, max_batch=b, max_topk=100) as eng:
        idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
        cnt = _t(np.full(b, 5, np.int32), gpu_device)
        out = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, ppr_iters=iters, k=100)
        got_idx, got_sc = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
        assert eng.timings()["slab_width"] == 128
        assert np.all(out.flags.cpu().numpy() == 0)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    worst = 0.0
    for q in list(range(0, b, 7)) + [b - 1]:
        ref = oracle.retrieve_one(index, qf[q], qp[q])
        assert tie_aware_equal(got_idx[q], ref.sorted_doc_ids[:100], ref.sorted_doc_scores[:100], rel_gap=2e-5), q
        want = ref.x[kg.passage_vertex][got_idx[q]]
        worst = max(worst, float((np.abs(got_sc[q] - want) / want).max()))
    assert worst < (1e-5 if iters < 20 else 3e-6), worst


def test_ppr_sweeps_hook_f16(case, gpu_device):
    """Measurement hook on the fp16 kernels runs (bench.py's roofline leg) and leaves no NaN."""
    import torch
    eng = case["eng"]
    b = 70
    idx, sc = eng.score_facts(_bf16(case["qf_bits"][:b], gpu_device), k=5)
    cnt = torch.full((b,), 5, dtype=torch.int32, device=gpu_device)
    ref = eng.retrieve(_bf16(case["qp_bits"][:b], gpu_device), idx, sc, cnt, ppr_iters=20, k=50)
    eng.ppr_sweeps(64, 3, 0.5, main_only=False, f16=True)     # fp16 state: batches <= 64 on an fp8-ready engine
    eng.ppr_sweeps(64, 2, 0.5, main_only=True, f16=True)
    eng.ppr_sweeps(b, 3, 0.5, main_only=False, f8=True)
    eng.ppr_sweeps(b, 2, 0.5, main_only=True, f8=True)
    out = eng.retrieve(_bf16(case["qp_bits"][:b], gpu_device), idx, sc, cnt, ppr_iters=20, k=50)
    torch.cuda.synchronize()
    assert torch.equal(ref.doc_idx, out.doc_idx) and torch.equal(ref.doc_score, out.doc_score)


@pytest.mark.parametrize("b", [1, 40, 70])
def test_hot_path_is_graph_capturable(case, gpu_device, b):
    """include/hrag.h promises that the compute entry points only enqueue work (no allocation, no
    synchronisation): phase A + B are captured into a HIP graph once and replayed on new query contents;
    every replay must reproduce the eager result bit for bit (B = 1 small-batch kernels, 40 fp16 state,
    70 fused top-k + fp8 state)."""
    import torch
    from hipporag_amd.engine import CapturedPipeline
    eng = case["eng"]
    qf_all, qp_all = _bf16(case["qf_bits"], gpu_device), _bf16(case["qp_bits"], gpu_device)
    cnt = torch.full((b,), 5, dtype=torch.int32, device=gpu_device)
    pipe = CapturedPipeline(eng, b, k_f=5, k=50, ppr_iters=20)
    for lo in (0, 7, 20):                                      # three different query sets through the same graph
        qf, qp = torch.roll(qf_all, lo, 0)[:b].contiguous(), torch.roll(qp_all, lo, 0)[:b].contiguous()
        got = [t.clone() for t in pipe(qf, qp)]
        idx, sc = eng.score_facts(qf, k=5)
        out = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=50)
        torch.cuda.synchronize()
        for a_, w_ in zip(got, (idx, sc, out.doc_idx, out.doc_score, out.flags)):
            assert torch.equal(a_, w_)


@pytest.mark.parametrize("damping,iters", [(0.3, 16), (0.3, 20), (0.6, 28), (0.7, 30), (0.85, 30)])
def test_retrieve_f8_state_other_damping_factors(gpu_device, damping, iters):
    """The fp8 stage scales follow the damping factor (residual contraction a per sweep, iterate growth
    (1 - a^m) / (1 - a) per stage).  The path is taken while damping^ppr_iters <= 2^-20 (the truncation
    error of the sweep count itself); beyond that (0.7 / 30, 0.85 / 30) hrag_retrieve takes the fp32 slabs."""
    from hipporag_amd.engine import HippoRAGEngine
    from hipporag_amd._lib import OPT_F32_STATE
    from hipporag_amd import synth
    import dataclasses
    b = 96
    kg, pass_bits, fact_bits, index = make_case(7000, 70000, 64, seed=77)
    index = dataclasses.replace(index, damping=damping)
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=13)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=14)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    errs = {}
    takes_f8 = damping ** iters <= 2.0 ** -20
    for name, flags, width in (("f8", 0, 128 if takes_f8 else 32), ("f32", OPT_F32_STATE, 32)):
        with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex,
                            kg.num_chunks, max_batch=b, max_topk=100, flags=flags) as eng:
            idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
            cnt = _t(np.full(b, 5, np.int32), gpu_device)
            out = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, damping=damping, ppr_iters=iters, k=100)
            got_idx, got_sc = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
            assert eng.timings()["slab_width"] == width
        worst = 0.0
        for q in range(0, b, 11):
            ref = oracle.retrieve_one(index, qf[q], qp[q])
            want = ref.x[kg.passage_vertex][got_idx[q]]
            worst = max(worst, float((np.abs(got_sc[q] - want) / want).max()))
        errs[name] = worst
    if takes_f8:
        assert errs["f8"] < (3e-6 if damping <= 0.5 and iters >= 20 else 1e-5), errs
    else:
        assert errs["f8"] == errs["f32"], errs          # the same kernels served both engines


@pytest.mark.parametrize("iters", [8, 40])
def test_retrieve_wide_batch_outside_the_fp8_iteration_range_takes_the_fp32_slabs(case, gpu_device, iters):
    """B > 64 with ppr_iters outside [16, 30] (fp8 stage plan) and no fp16 state for that width: the
    fp32 slab kernels serve the call; 8 sweeps are compared with the oracle's 8-sweep power iteration
    (same start x0 = v), 40 sweeps with the exact solution."""
    eng, kg, index = case["eng"], case["kg"], case["index"]
    b = 70
    idx, sc = eng.score_facts(_bf16(case["qf_bits"][:b], gpu_device), k=5)
    cnt = _t(np.full(b, 5, np.int32), gpu_device)
    out = eng.retrieve(_bf16(case["qp_bits"][:b], gpu_device), idx, sc, cnt, ppr_iters=iters, k=100)
    got_idx, got_sc = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
    assert eng.timings()["slab_width"] == 32
    qf, qp = bf16_bits_to_float(case["qf_bits"]), bf16_bits_to_float(case["qp_bits"])
    for q in range(0, b, 9):
        ref = oracle.retrieve_one(index, qf[q], qp[q], ppr_mode="power" if iters == 8 else "exact", ppr_iters=iters)
        want = ref.x[kg.passage_vertex][got_idx[q]]
        assert (np.abs(got_sc[q] - want) / want).max() < 1e-5, q
        assert tie_aware_equal(got_idx[q], ref.sorted_doc_ids[:100], ref.sorted_doc_scores[:100], rel_gap=2e-5), q


# ----------------------------------------------------------------------------- full size (BASELINE configs[2])
def test_full_size_cfg3_properties_and_spot_parity(gpu_device):
    """1M-node / 10M-edge KG, 1M x 768 bf16 embeddings, batch 256, 20 sweeps (staged fp8 state; the
    64-query sub-batch takes the two-stage fp16 state).
    Size-independent properties over the whole batch + the oracle on a few queries:
      * every doc-score row is sorted (score desc, index desc), ids are unique and in range;
      * determinism: a second run is bit-identical (no atomics anywhere on the path);
      * batch-independence: the same queries as a batch of 64 (fp16 state instead of fp8) give the
        same ids and scores within the tolerance;
      * spot parity on 32 queries, the 2048 best-ranked passages of each (the selection kernel's maximum k): ids
        identical (tie-class aware) and scores <= 1e-5 relative vs the oracle."""
    import torch
    from hipporag_amd import synth
    from hipporag_amd.engine import HippoRAGEngine
    V, E, D, B, seed = 1_000_000, 10_000_000, 768, 256, 1237
    kg = synth.make_kg(V, E, seed)
    pass_emb = synth.make_embeddings_torch(kg.n_passages, D, seed + 1, gpu_device)
    fact_emb = synth.make_embeddings_torch(kg.n_facts, D, seed + 2, gpu_device)
    qf = synth.make_queries_torch(fact_emb, B, 11)[0]
    qp = synth.make_queries_torch(pass_emb, B, 12)[0]
    cnt = torch.full((B,), 5, dtype=torch.int32, device=gpu_device)
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_emb, fact_emb, kg.subj_vertex, kg.obj_vertex,
                        kg.num_chunks, max_batch=B, max_topk=2048) as eng:
        idx, sc = eng.score_facts(qf, k=5)
        out = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=200)
        out2 = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=200)
        w256 = eng.timings()["slab_width"]
        deep = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=2048)
        deep_ids, deep_sc = deep.doc_idx.cpu().numpy(), deep.doc_score.cpu().numpy()
        # what the mirror and the adapter run by default: accelerated stages under the convergence contract
        eng.set_flags(_lib.OPT_ACCEL, True)
        acc = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=2048, ppr_tol=1.5e-6, ppr_max_iters=30)
        eng.set_flags(_lib.OPT_ACCEL, False)
        acc_ids, acc_sc = acc.doc_idx.cpu().numpy(), acc.doc_score.cpu().numpy()
        acc_used, acc_resid, acc_flags = (t.cpu().numpy() for t in (acc.iters_used, acc.residual, acc.flags))
        sub = eng.retrieve(qp[:64], idx[:64], sc[:64], cnt[:64], ppr_iters=20, k=200)
        torch.cuda.synchronize()
        assert w256 == 128 and eng.timings()["slab_width"] == 64
    ids, scores = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
    assert torch.equal(out.doc_idx, out2.doc_idx) and torch.equal(out.doc_score, out2.doc_score)
    assert np.all(out.flags.cpu().numpy() == 0)
    assert ids.min() >= 0 and ids.max() < kg.n_passages
    for q in range(B):
        assert len(np.unique(ids[q])) == 200
        d = np.diff(scores[q])
        assert np.all(d <= 0)
        tie = d == 0
        assert np.all(ids[q][1:][tie] < ids[q][:-1][tie])
    sub_ids, sub_sc = sub.doc_idx.cpu().numpy(), sub.doc_score.cpu().numpy()
    for q in range(64):
        assert tie_aware_equal(sub_ids[q], ids[q], scores[q], rel_gap=4e-6), q
        full = dict(zip(ids[q].tolist(), scores[q].tolist()))
        common = [i for i in sub_ids[q].tolist() if i in full]
        got = np.array([sub_sc[q][list(sub_ids[q]).index(i)] for i in common[:50]])
        want = np.array([full[i] for i in common[:50]])
        np.testing.assert_allclose(got, want, rtol=4e-6)   # two different reduced-precision states
    assert np.array_equal(deep_ids[:, :200], ids) and np.array_equal(deep_sc[:, :200], scores)
    # oracle on 32 queries (PRPACK port: ~0.3 s each + 1.5 s index preparation)
    a = oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight)
    index = oracle.RefIndex(fact_emb=fact_emb.float().cpu().numpy(), passage_emb=pass_emb.float().cpu().numpy(),
                            subj_vertex=kg.subj_vertex, obj_vertex=kg.obj_vertex, num_chunks=kg.num_chunks,
                            passage_vertex=kg.passage_vertex, p=oracle.column_normalize(a))
    qf_h, qp_h = qf.float().cpu().numpy(), qp.float().cpu().numpy()
    worst, gap, exact, npos, worst_a, exact_a = 0.0, 0.0, 0, 0, 0.0, 0
    ulp4_queries = 0          # queries whose 2048 ids agree at SURVEY 8(c)'s OWN window (4 ulp of fp32 = 4.8e-7): reported
    from oracle.checks import ulp4_report
    for q in list(range(0, B, 8)):
        ref = oracle.retrieve_one(index, qf_h[q], qp_h[q])
        # the tie window follows the measured error (tests/helpers.ranked_parity): 2e-6 at this size, not 2e-5
        rep = ranked_parity(deep_ids[q], deep_sc[q], ref.sorted_doc_ids, ref.sorted_doc_scores, ref.x[kg.passage_vertex])
        assert rep["equal"], (q, rep)
        ulp4_queries += int(ulp4_report(deep_ids[q], ref.sorted_doc_ids, ref.sorted_doc_scores)["equal"])
        worst, gap = max(worst, rep["worst_rel_err"]), max(gap, rep["rel_gap"])
        exact += rep["exact_positions"]; npos += rep["n"]
        rep_a = ranked_parity(acc_ids[q], acc_sc[q], ref.sorted_doc_ids, ref.sorted_doc_scores, ref.x[kg.passage_vertex])
        assert rep_a["equal"], (q, rep_a)
        worst_a, exact_a = max(worst_a, rep_a["worst_rel_err"]), exact_a + rep_a["exact_positions"]
    write_test_report("cfg3_full_size_parity", {"queries": B // 8, "ranks_per_query": 2048, "max_rel_score_err": worst,
                                                "exact_id_fraction": exact / npos, "tie_window_rel": gap,
                                                "queries_with_ids_equal_at_the_4ulp_window": ulp4_queries,
                                                "accel_contract": {"max_rel_score_err": worst_a, "exact_id_fraction": exact_a / npos,
                                                                   "sweeps_used_max": int(acc_used.max()),
                                                                   "residual_max": float(acc_resid.max())}})
    assert worst < 1e-5, worst
    assert gap <= 2.5e-6, gap                     # i.e. the ids agree outside a window 8x narrower than round 3's
    # accelerated stages + contract: converged (no flag), fewer sweeps than the plain 20, the same parity bar
    assert np.all(acc_flags == 0) and acc_resid.max() <= 1.5e-6 and 16 <= acc_used.max() < 20, (acc_used.max(), acc_resid.max())
    assert worst_a < 1e-5, worst_a


# ----------------------------------------------------------------------------- full size (BASELINE configs[1])
def test_full_size_cfg2_fp16_state_parity(gpu_device):
    """BASELINE configs[1] at its own sizes through the path it takes in production: 100k-node / 1M-edge KG,
    100k x 768 bf16 embeddings, batch 64, 20 sweeps -> the two-stage fp16 state (csrc/ppr16.hip).  16 queries
    against the fp64 oracle (reference call site HippoRAG.py:1736-1749): ranked ids identical outside a tie window
    that follows the measured error, every score within 1e-5 relative; plus determinism and sortedness of the whole
    batch.  Reports exact_id_fraction (gpurun_out/test_reports/cfg2_full_size_parity.json)."""
    import torch
    from hipporag_amd import synth
    from hipporag_amd.engine import HippoRAGEngine
    V, E, D, B, seed, K = 100_000, 1_000_000, 768, 64, 1236, 200
    kg = synth.make_kg(V, E, seed)
    pass_emb = synth.make_embeddings_torch(kg.n_passages, D, seed + 1, gpu_device)
    fact_emb = synth.make_embeddings_torch(kg.n_facts, D, seed + 2, gpu_device)
    qf = synth.make_queries_torch(fact_emb, B, 21)[0]
    qp = synth.make_queries_torch(pass_emb, B, 22)[0]
    cnt = torch.full((B,), 5, dtype=torch.int32, device=gpu_device)
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_emb, fact_emb, kg.subj_vertex, kg.obj_vertex,
                        kg.num_chunks, max_batch=B, max_topk=K) as eng:
        idx, sc = eng.score_facts(qf, k=5)
        out = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=K)
        out2 = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=K)
        torch.cuda.synchronize()
        assert eng.timings()["slab_width"] == 64         # the fp16 state served the call (not the fp8 / fp32 one)
    ids, scores = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
    assert torch.equal(out.doc_idx, out2.doc_idx) and torch.equal(out.doc_score, out2.doc_score)
    assert np.all(out.flags.cpu().numpy() == 0) and int(out.iters_used.max()) == 20
    for q in range(B):
        assert len(np.unique(ids[q])) == K and ids[q].min() >= 0 and ids[q].max() < kg.n_passages
        d = np.diff(scores[q])
        assert np.all(d <= 0) and np.all(ids[q][1:][d == 0] < ids[q][:-1][d == 0])
    a = oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight)
    index = oracle.RefIndex(fact_emb=fact_emb.float().cpu().numpy(), passage_emb=pass_emb.float().cpu().numpy(),
                            subj_vertex=kg.subj_vertex, obj_vertex=kg.obj_vertex, num_chunks=kg.num_chunks,
                            passage_vertex=kg.passage_vertex, p=oracle.column_normalize(a))
    qf_h, qp_h = qf.float().cpu().numpy(), qp.float().cpu().numpy()
    fact_ids = idx.cpu().numpy()
    worst, gap, exact, npos = 0.0, 0.0, 0, 0
    for q in range(0, B, 4):                                # 16 queries
        ref = oracle.retrieve_one(index, qf_h[q], qp_h[q])
        np.testing.assert_array_equal(fact_ids[q], ref.fact_candidates)
        rep = ranked_parity(ids[q], scores[q], ref.sorted_doc_ids, ref.sorted_doc_scores, ref.x[kg.passage_vertex])
        assert rep["equal"], (q, rep)
        worst, gap = max(worst, rep["worst_rel_err"]), max(gap, rep["rel_gap"])
        exact += rep["exact_positions"]; npos += rep["n"]
    write_test_report("cfg2_full_size_parity", {"queries": 16, "ranks_per_query": K, "max_rel_score_err": worst,
                                                "exact_id_fraction": exact / npos, "tie_window_rel": gap})
    assert worst < 1e-5, worst


# ----------------------------------------------------------------------------- small-batch kernels (B <= 8)
@pytest.mark.parametrize("b", [1, 2, 3, 5, 8])
def test_retrieve_small_batch_path_vs_oracle(case, gpu_device, b):
    """hrag_retrieve takes the small-batch kernels (csrc/ppr_sv.hip) for B <= 8 -- the IRCoT /
    per-method-seam shape.  Filter subsets, a DPR-fallback row and shared seeds included."""
    eng, kg = case["eng"], case["kg"]
    qf_bits, qp_bits = case["qf_bits"][10:10 + b].copy(), case["qp_bits"][10:10 + b].copy()
    if b >= 3:
        qf_bits[2] = qf_bits[0]                       # two queries seed the same entities
    sub = dict(case, qf_bits=qf_bits, qp_bits=qp_bits)
    idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
    idx_h, sc_h = idx.cpu().numpy(), sc.cpu().numpy()
    plans = [[0, 1, 2, 3, 4], [], [0, 1, 2, 3, 4], [3, 1], [2], [4, 0, 1], [0], [1, 2, 3]][:b]
    if b == 1:
        plans = [[0, 1, 2, 3, 4]]
    kept_idx = np.full((b, 5), -1, np.int32)
    kept_sc = np.zeros((b, 5), np.float32)
    kept_cnt = np.zeros(b, np.int32)
    kept_lists = []
    for q, plan in enumerate(plans):
        kept_idx[q, :len(plan)] = idx_h[q, plan]
        kept_sc[q, :len(plan)] = sc_h[q, plan]
        kept_cnt[q] = len(plan)
        kept_lists.append(idx_h[q, plan].tolist())
    out = eng.retrieve(_bf16(qp_bits, gpu_device), _t(kept_idx, gpu_device), _t(kept_sc, gpu_device),
                       _t(kept_cnt, gpu_device), ppr_iters=20, k=100)
    got_idx, got_sc, flags = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy(), out.flags.cpu().numpy()
    assert eng.timings()["slab_width"] == {1: 1, 2: 2, 3: 4, 5: 8, 8: 8}[b]
    refs = _oracle_batch(sub, kept_lists)
    for q in range(b):
        ref = refs[q]
        assert bool(flags[q] & 1) == ref.used_dpr == (len(plans[q]) == 0)
        want_ids, want_sc = ref.sorted_doc_ids[:100], ref.sorted_doc_scores[:100]
        if ref.used_dpr:
            assert tie_aware_equal(got_idx[q], want_ids, want_sc, abs_gap=3e-6), q
            np.testing.assert_allclose(got_sc[q], want_sc, rtol=0, atol=3e-6)
        else:
            assert tie_aware_equal(got_idx[q], want_ids, want_sc, rel_gap=2e-5), q
            np.testing.assert_allclose(got_sc[q], ref.x[kg.passage_vertex][got_idx[q]], rtol=1e-5, atol=0)


def test_small_batch_split_topk_and_gemv_at_scale(gpu_device):
    """B <= 8 on a 300k-vertex index: cosine scores come from the GEMV kernel and every top-k row is
    split over several workgroups (two-level selection) -- ids and scores must equal the oracle's."""
    from hipporag_amd.engine import HippoRAGEngine
    from hipporag_amd import synth
    kg, pass_bits, fact_bits, index = make_case(300_000, 1_500_000, 32, seed=5)
    assert kg.n_facts >= 32768 and kg.n_passages >= 32768
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex,
                        kg.num_chunks, max_batch=8, max_topk=200) as eng:
        for b in (1, 3, 8):
            qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=40 + b)
            qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=50 + b)
            qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
            idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
            d_idx, d_sc = eng.dense_retrieve(_bf16(qp_bits, gpu_device), k=200)
            idx, sc, d_idx, d_sc = (t.cpu().numpy() for t in (idx, sc, d_idx, d_sc))
            for q in range(b):
                s = oracle.fact_scores(index.fact_emb, qf[q])
                want = oracle.topk_desc(s, 5)
                assert tie_aware_equal(idx[q], want, s[want], abs_gap=2e-6), (b, q)
                np.testing.assert_allclose(sc[q], s[idx[q]], rtol=0, atol=2e-6)
                ids, scores = oracle.retrieve_dpr_one(index, qp[q])
                assert tie_aware_equal(d_idx[q], ids[:200], scores[:200], abs_gap=3e-6), (b, q)
                np.testing.assert_allclose(d_sc[q], scores[:200], rtol=0, atol=3e-6)


# ----------------------------------------------------------------------------- awkward graphs, all three PPR paths
@pytest.mark.parametrize("b", [4, 40, 100])
def test_retrieve_on_graph_with_dangling_hub_and_parallel_edges(gpu_device, b):
    """Isolated passages (dangling: they keep teleport mass but have no edges, HippoRAG.py:1171-1187
    adds every stored passage as a vertex), isolated entities -- one of them a seed --, a hub with
    ~2000 neighbours (long-row segments), duplicated parallel edges (summed, :1189-1223), interleaved
    passage / entity vertex numbering, V not a multiple of 8, and a fact whose subject vertex is a
    passage.  B = 4 takes the small-batch kernels, B = 40 the two-stage fp16 kernels, B = 100 the staged
    fp8 kernels; an engine
    created with HRAG_OPT_F32_STATE runs the fp32 slab kernels on the same input."""
    from hipporag_amd.engine import HippoRAGEngine
    from hipporag_amd.graph import build_csr
    from hipporag_amd._lib import OPT_F32_STATE
    from hipporag_amd import synth
    rng = np.random.default_rng(99)
    kg0 = synth.make_kg(2997, 20000, seed=3)
    v = kg0.num_vertices
    relabel = rng.permutation(v)                                   # mix passage and entity ids
    src, dst, w = relabel[kg0.src], relabel[kg0.dst], kg0.weight.copy()
    pv = relabel[kg0.passage_vertex].astype(np.int32)
    ents = relabel[:kg0.n_entities]
    iso_p, iso_e = pv[:6], ents[:4]                                # cut every edge of these vertices
    dead = np.isin(src, np.concatenate([iso_p, iso_e])) | np.isin(dst, np.concatenate([iso_p, iso_e]))
    src, dst, w = src[~dead], dst[~dead], w[~dead]
    hub = ents[10]
    nb = rng.choice(ents[20:], 2000, replace=False)
    src = np.concatenate([src, np.full(2000, hub), src[:500]])     # hub row + 500 parallel duplicates
    dst = np.concatenate([dst, nb, dst[:500]])
    w = np.concatenate([w, rng.uniform(0.5, 3.0, 2000), w[:500]])
    csr = build_csr(v, src, dst, w)
    subj, obj = relabel[kg0.subj_vertex].astype(np.int32), relabel[kg0.obj_vertex].astype(np.int32)
    subj[0], obj[0] = iso_e[0], hub                                # fact 0 seeds an isolated entity + the hub
    subj[1] = pv[20]                                               # a "phrase" that is a passage vertex
    obj[2] = -1                                                    # phrase absent from the graph
    nchunks = np.zeros(v, np.int32)
    nchunks[relabel] = kg0.num_chunks
    pass_bits = synth.make_embeddings_np(kg0.n_passages, 64, 7)
    fact_bits = synth.make_embeddings_np(kg0.n_facts, 64, 8)
    a = oracle.build_symmetric_csr(v, src, dst, w)
    index = oracle.RefIndex(bf16_bits_to_float(fact_bits), bf16_bits_to_float(pass_bits), subj, obj, nchunks, pv,
                            oracle.column_normalize(a))
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=1)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=2)
    qf_bits[:3] = fact_bits[:3]                                    # queries 0..2 retrieve facts 0..2 first
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    refs = [oracle.retrieve_one(index, qf[q], qp[q]) for q in range(b)]
    assert refs[0].fact_candidates[0] == 0 and refs[1].fact_candidates[0] == 1
    for flags in (0, OPT_F32_STATE):
        with HippoRAGEngine(csr, pv, pass_bits, fact_bits, subj, obj, nchunks, max_batch=b, max_topk=100,
                            flags=flags) as eng:
            idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
            cnt = _t(np.full(b, 5, np.int32), gpu_device)
            out = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, ppr_iters=30, k=100)
            got_idx, got_sc = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
            assert np.all(out.flags.cpu().numpy() == 0)
            width = eng.timings()["slab_width"]
            assert width == ({4: 4, 40: 64, 100: 128}[b] if flags == 0 else {4: 4, 40: 32, 100: 32}[b])
        for q in range(b):
            ref = refs[q]
            assert tie_aware_equal(got_idx[q], ref.sorted_doc_ids[:100], ref.sorted_doc_scores[:100], rel_gap=2e-5), (flags, q)
            want = ref.x[pv][got_idx[q]]
            assert (np.abs(got_sc[q] - want) / want).max() < 1e-5, (flags, q)


# ----------------------------------------------------------------------------- wide-batch GEMM (sim_gemm256.hip)
@pytest.mark.parametrize("rows,dim,batch", [(5000, 768, 256), (300, 64, 65), (1029, 128, 130), (40000, 1024, 200),
                                            (256, 64, 128), (777, 192, 129)])
def test_sim_gemm256_is_bit_identical_to_the_small_tile_kernel(gpu_device, rows, dim, batch):
    """batch > 64 with dim % 64 == 0 takes the 256-row workgroup tile with LDS-direct loads; every score must be
    the same chain of MFMAs as in sim_gemm_kernel (the fused top-k's rescore kernel depends on it).  The
    accumulate path (out += product) always runs the small-tile kernel: into a zeroed buffer it yields that
    kernel's plain result."""
    import ctypes as C
    import torch
    from hipporag_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=gpu_device)
    g.manual_seed(rows + batch)
    emb = torch.randn((rows, dim), generator=g, device=gpu_device).to(torch.bfloat16).contiguous()
    q = torch.randn((batch, dim), generator=g, device=gpu_device).to(torch.bfloat16).contiguous()
    ld = rows + (-rows) % 4
    new = torch.full((batch, ld), float("nan"), device=gpu_device)
    old = torch.zeros((batch, ld), device=gpu_device)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.hrag_sim_gemm(emb.data_ptr(), rows, dim, q.data_ptr(), batch, new.data_ptr(), ld, 0, 0, stream))
    _lib.check(lib.hrag_sim_gemm(emb.data_ptr(), rows, dim, q.data_ptr(), batch, old.data_ptr(), ld, 1, 0, stream))
    torch.cuda.synchronize()
    assert torch.equal(new[:, :rows], old[:, :rows])
    want = q.double() @ emb.double().T
    assert float((new[:, :rows].double() - want).abs().max()) < 3e-4 * float(want.abs().max() + 1)
