"""Rows longer than 512 entries are cut into <= 64 segments that different wavefronts (on different XCDs, each with a
private L2) sum up; the segment that arrives last combines the partial sums inside the sweep kernel (sc1 stores and
loads + one agent-scope arrival counter per row, csrc/ppr16.hip).  A hand-off that read a stale partial sum would
show up as a result that changes from call to call or leaves the parity bar -- so: a graph with 48 hub rows of
600 .. 20 000 entries (four of them passages: the passage-rows-only last sweep has long rows too), every PPR kernel
family (B <= 8, fp16 state, fp8 state), every call repeated, all passage scores compared."""

import numpy as np
import pytest

import oracle
from hipporag_amd import synth
from hipporag_amd.graph import bf16_bits_to_float

from .test_gpu_fp8_adversarial import _bf16, _index, _t

pytestmark = pytest.mark.gpu


def _hub_graph():
    rng = np.random.default_rng(19)
    n, n_hub = 30000, 48
    hubs = np.arange(n_hub)
    degs = np.round(np.geomspace(600, 20000, n_hub)).astype(np.int64)
    src = [rng.integers(n_hub, n, 90000)]
    dst = [rng.integers(n_hub, n, 90000)]
    for h, d in zip(hubs, degs):
        src.append(np.full(d, h))
        dst.append(rng.choice(np.arange(n_hub, n), size=d, replace=False))
    src, dst = np.concatenate(src), np.concatenate(dst)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    w = rng.uniform(0.5, 4.0, len(src))
    pv = np.concatenate([[3, 11, 29, 47], np.arange(n_hub, n, 16)])
    return n, src, dst, w, pv


@pytest.fixture(scope="module")
def hub_case():
    n, src, dst, w, pv = _hub_graph()
    csr, pass_bits, fact_bits, index = _index(n, src, dst, w, pv, 64, seed=23)
    assert int((np.diff(csr.row_ptr) > 512).sum()) >= 48
    return csr, pass_bits, fact_bits, index


@pytest.mark.parametrize("b,width", [(1, 1), (3, 4), (8, 8), (40, 64), (130, 128), (256, 128)])
def test_long_rows_are_finished_inside_the_sweep_kernel(gpu_device, hub_case, b, width):
    import torch
    from hipporag_amd.engine import HippoRAGEngine
    csr, pass_bits, fact_bits, index = hub_case
    n_p = len(index.passage_vertex)
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=5)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=6)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    runs = []
    with HippoRAGEngine(csr, index.passage_vertex, pass_bits, fact_bits, index.subj_vertex, index.obj_vertex,
                        index.num_chunks, max_batch=b, max_topk=n_p) as eng:
        idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=5)
        cnt = _t(np.full(b, 5, np.int32), gpu_device)
        for _ in range(6):
            out = eng.retrieve(_bf16(qp_bits, gpu_device), idx, sc, cnt, ppr_iters=20, k=n_p)
            torch.cuda.synchronize()
            runs.append((out.doc_idx.cpu().numpy().copy(), out.doc_score.cpu().numpy().copy()))
            assert np.all(out.flags.cpu().numpy() == 0)
        assert eng.timings()["slab_width"] == width
    for ids, scores in runs[1:]:
        assert np.array_equal(ids, runs[0][0]) and np.array_equal(scores, runs[0][1])
    got_idx, got_sc = runs[0]
    worst = 0.0
    for q in sorted({0, b // 2, b - 1}):
        want = oracle.retrieve_one(index, qf[q], qp[q]).x[index.passage_vertex]
        worst = max(worst, float(np.abs(got_sc[q] / want[got_idx[q]] - 1).max()))
    assert worst < 1e-5, worst


def test_fused_fact_topk_hand_off_is_repeatable(gpu_device):
    """Pass 3 of the fused fact top-k (csrc/sim_gemm.hip) hands the candidate keys of k workgroups to the one that
    arrives last (the same sc1 + arrival-counter hand-off): 12 calls in a row -- the per-query records and their
    counters are reused every time -- must give identical ids and scores, equal to the exact ranking."""
    import torch
    from hipporag_amd.engine import HippoRAGEngine
    from hipporag_amd.graph import build_csr
    rng = np.random.default_rng(3)
    n, n_p, n_f, b, dim = 4000, 500, 40000, 200, 128
    src, dst = rng.integers(0, n, 20000), rng.integers(0, n, 20000)
    keep = src != dst
    csr = build_csr(n, src[keep], dst[keep], np.ones(int(keep.sum())))
    pv = np.arange(n_p, dtype=np.int32)
    pass_bits, fact_bits = synth.make_embeddings_np(n_p, dim, 1), synth.make_embeddings_np(n_f, dim, 2)
    subj = rng.integers(n_p, n, n_f).astype(np.int32)
    obj = ((subj - n_p + 1 + rng.integers(0, n - n_p - 1, n_f)) % (n - n_p) + n_p).astype(np.int32)
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=9)
    with HippoRAGEngine(csr, pv, pass_bits, fact_bits, subj, obj, np.ones(n, np.int32), max_batch=b, max_topk=100) as eng:
        runs = []
        for _ in range(12):
            idx, sc = eng.score_facts(_bf16(qf_bits, gpu_device), k=16)
            torch.cuda.synchronize()
            runs.append((idx.cpu().numpy().copy(), sc.cpu().numpy().copy()))
    for idx, sc in runs[1:]:
        assert np.array_equal(idx, runs[0][0]) and np.array_equal(sc, runs[0][1])
    scores = bf16_bits_to_float(qf_bits).astype(np.float64) @ bf16_bits_to_float(fact_bits).astype(np.float64).T
    for q in range(0, b, 17):
        order = np.lexsort((-np.arange(n_f), -scores[q]))[:16]
        got = runs[0][0][q]
        assert set(got.tolist()) == set(order.tolist()) or np.allclose(np.sort(scores[q][got]), np.sort(scores[q][order]), rtol=0, atol=2e-6)
