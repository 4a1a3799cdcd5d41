"""hrag_workspace_create / hrag_engine_stats: a second workspace on the SAME index lets another thread
retrieve on its own stream while the engine is inside a call -- no HRAG_EBUSY, no second copy of the index, results
bit-identical to the sequential calls.  The reference has nothing to compare with here (HippoRAG.py:459 is a serial loop
in one thread); the check is identity with the single-handle path, whose parity tests/test_gpu_parity.py carries."""

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bf16(bits, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).to(dev).view(torch.bfloat16)


@pytest.fixture(scope="module")
def case():
    import torch
    from hipporag_amd import synth
    from hipporag_amd.engine import HippoRAGEngine
    from tests.helpers import make_case
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    kg, pass_bits, fact_bits, index = make_case(6000, 60000, 128, seed=4242)
    eng = HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                         max_batch=130, max_topk=50)
    yield dev, kg, pass_bits, fact_bits, eng, synth
    eng.close()


def _run(eng, qf, qp, k=50):
    import torch
    b = qf.shape[0]
    idx, sc = eng.score_facts(qf, k=5)
    cnt = torch.full((b,), 5, dtype=torch.int32, device=qf.device)
    out = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=k)
    return idx, sc, out.doc_idx, out.doc_score, out.flags


def test_a_workspace_shares_the_index_and_reports_it(case):
    import torch
    dev, kg, pass_bits, fact_bits, eng, synth = case
    free0 = torch.cuda.mem_get_info(dev)[0]
    w = eng.workspace()
    torch.cuda.synchronize()
    used = free0 - torch.cuda.mem_get_info(dev)[0]
    se, sw = eng.stats(), w.stats()
    assert se["is_workspace"] == 0 and sw["is_workspace"] == 1 and se["live_workspaces"] == 1
    assert sw["index_bytes"] == se["index_bytes"] > 0
    assert sw["workspace_bytes"] == se["workspace_bytes"] > 0           # the same per-call half, allocated again
    # ... and ONLY that: the index (graph, matrices, embeddings) was not copied (allocator granularity: 2 MB slack per buffer)
    assert used <= sw["workspace_bytes"] + 128 * (2 << 20), (used, sw)
    assert se["ppr_states"] & 8 and se["fp8_unavailable"] == 0            # this engine has the staged e4m3 state
    # the C ABI refuses to destroy an engine whose index is still borrowed
    from hipporag_amd._lib import HragError
    with pytest.raises(HragError):
        from hipporag_amd._lib import check
        check(eng._lib.hrag_engine_destroy(eng._handle))
    w.close()
    assert eng.stats()["live_workspaces"] == 0


@pytest.mark.parametrize("batch", [2, 9, 130])          # the small-batch, fp16-state and e4m3-state paths
def test_two_threads_two_streams_one_index_bit_identical(case, batch):
    import torch
    dev, kg, pass_bits, fact_bits, eng, synth = case
    qa = (_bf16(synth.make_queries_np(fact_bits, batch, seed=11)[0], dev), _bf16(synth.make_queries_np(pass_bits, batch, seed=12)[0], dev))
    qb = (_bf16(synth.make_queries_np(fact_bits, batch, seed=21)[0], dev), _bf16(synth.make_queries_np(pass_bits, batch, seed=22)[0], dev))
    ref_a = [t.clone() for t in _run(eng, *qa)]
    ref_b = [t.clone() for t in _run(eng, *qb)]
    torch.cuda.synchronize()
    assert not torch.equal(ref_a[2], ref_b[2])
    w = eng.workspace()
    errors, rounds = [], 25
    start = threading.Barrier(2)

    def worker(handle, q, ref, name):
        try:
            torch.cuda.set_device(dev)
            stream = torch.cuda.Stream(device=dev)
            start.wait()
            with torch.cuda.stream(stream):
                for _ in range(rounds):
                    got = _run(handle, *q)
                    stream.synchronize()
                    for g, r in zip(got, ref):
                        if not torch.equal(g, r):
                            raise AssertionError(f"{name}: result differs from the sequential call")
        except Exception as exc:   # HragError(HRAG_EBUSY) would land here
            errors.append((name, repr(exc)))

    ts = [threading.Thread(target=worker, args=(eng, qa, ref_a, "engine")),
          threading.Thread(target=worker, args=(w, qb, ref_b, "workspace"))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errors, errors
    st = w.stats()
    assert st["calls_retrieve"] == rounds and st["queries"] == rounds * batch
    assert st["last_ppr_state"] == (4 if batch <= 8 else 2 if batch <= 64 else 8)
    w.close()


def test_an_engine_beyond_the_e4m3_limits_says_so():
    """include/hrag.h: no col_sum (or V + 1 > 2^24) => no staged e4m3 state; batches > 64 then run on the two-stage fp16
    state (fp32 slabs beyond V * 128 >= 2^32) and hrag_engine_stats names the reason instead of a silent slowdown."""
    import torch
    from hipporag_amd import synth
    from hipporag_amd.engine import HippoRAGEngine
    from tests.helpers import make_case
    dev = torch.device("cuda", 0)
    kg, pass_bits, fact_bits, _ = make_case(3000, 30000, 64, seed=99)
    import dataclasses
    bare = dataclasses.replace(kg.csr, col_sum=None)            # what a caller without the weighted degrees hands over
    with HippoRAGEngine(bare, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex, kg.num_chunks,
                        max_batch=130, max_topk=20) as eng:
        st = eng.stats()
        assert not (st["ppr_states"] & 8) and st["fp8_unavailable"] & 1 and st["fp8_unavailable_reasons"]
        qf = _bf16(synth.make_queries_np(fact_bits, 130, seed=1)[0], dev)
        qp = _bf16(synth.make_queries_np(pass_bits, 130, seed=2)[0], dev)
        _run(eng, qf, qp, k=20)
        torch.cuda.synchronize()
        assert eng.stats()["last_ppr_state"] == 2            # the fp16 state took the wide batch
