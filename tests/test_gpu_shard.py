"""Row-sharded hot path (include/hrag.h hrag_shard_*, hipporag_amd/dist.py) on ONE device: the
`world` shard engines run as threads of this process and meet at barriers (dist.LocalComm), sharing the
three e4m3 state buffers -- the emulated gather SURVEY.md 8(e) prescribes.  Every output row of every
sweep is computed by exactly one shard from the same replicated iterate, in the same order of
additions as the single-GPU kernel, so the result must be BIT-IDENTICAL to the unsharded engine on the
same (relabelled) index, and within the 1e-5 parity bar of the oracle on the original index."""

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


def _run_shards(world, sidx, pass_bits, fact_bits, qf, qp, kw, groups, device, max_topk, filter_fn=None):
    """All shards of `sidx` as threads on one device (dist.run_local_shards); returns rank 0's (fact idx, fact
    score, doc idx, doc score, flags) after asserting that every rank computed the same replicated result."""
    from hipporag_amd import dist as hd
    return hd.run_local_shards(world, sidx, pass_bits, fact_bits, qf, qp, kw, groups, device, max_topk, filter_fn)


@pytest.mark.parametrize("world,b,groups,iters", [(8, 130, 2, 20), (8, 256, 1, 20), (3, 70, 0, 24), (2, 40, 2, 20)])
def test_shards_on_one_device_are_bit_identical_to_the_single_gpu_engine(gpu_device, world, b, groups, iters):
    import torch
    from hipporag_amd import dist as hd
    from hipporag_amd.engine import HippoRAGEngine
    kg, pass_bits, fact_bits, index = make_case(12000, 120000, 128, seed=500 + world, power_law=(world == 3))
    sidx = hd.shard_index(kg.csr, kg.passage_vertex, world, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
    assert sidx.num_vertices % world == 0 and sidx.num_vertices >= kg.num_vertices
    nnz_shard = [int(sidx.csr.row_ptr[(g + 1) * sidx.rows_per_shard] - sidx.csr.row_ptr[g * sidx.rows_per_shard])
                 for g in range(world)]
    assert max(nnz_shard) < 1.02 * (sum(nnz_shard) / world) + 64, nnz_shard      # balanced by construction
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=3)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=4)
    qf_t, qp_t = _bf16(qf_bits, gpu_device), _bf16(qp_bits, gpu_device)
    kw = dict(link_top_k=5, damping=0.5, passage_node_weight=0.05, ppr_iters=iters, k=100)
    got = _run_shards(world, sidx, pass_bits, fact_bits, qf_t, qp_t, kw, groups, gpu_device, 100)
    # ---- the single-GPU engine on the same relabelled index (every query on the fp8 state: batch > 64)
    with HippoRAGEngine(sidx.csr, sidx.passage_vertex, pass_bits, fact_bits, sidx.subj_vertex, sidx.obj_vertex,
                        sidx.num_chunks, max_batch=max(b, 65), max_topk=100) as eng:
        idx, sc = eng.score_facts(qf_t, k=5)
        cnt = torch.full((b,), 5, dtype=torch.int32, device=gpu_device)
        out = eng.retrieve(qp_t, idx, sc, cnt, **kw)
        torch.cuda.synchronize()
        one = tuple(t.cpu().numpy() for t in (idx, sc, out.doc_idx, out.doc_score, out.flags))
        path_width = eng.timings()["slab_width"]
    np.testing.assert_array_equal(got[0], one[0])                 # fact ids
    np.testing.assert_array_equal(got[1], one[1])                 # normalised fact scores, bit for bit
    assert np.all(got[4] == 0) and np.all(one[4] == 0)
    if b > 64:
        assert path_width == 128
        np.testing.assert_array_equal(got[2], one[2])             # ranked passage positions
        np.testing.assert_array_equal(got[3], one[3])             # PPR probabilities, bit for bit
    # ---- and the oracle on the ORIGINAL index (passage positions do not change under the relabelling)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    worst = 0.0
    for q in list(range(0, b, 13)) + [b - 1]:
        ref = oracle.retrieve_one(index, qf[q], qp[q])
        assert tie_aware_equal(got[2][q], ref.sorted_doc_ids[:100], ref.sorted_doc_scores[:100], rel_gap=2e-5), q
        want = ref.x[kg.passage_vertex][got[2][q]]
        worst = max(worst, float((np.abs(got[3][q] - want) / want).max()))
    assert worst < 3e-6, worst


def test_shards_with_filter_subsets_dpr_fallback_isolated_passages_and_empty_fact_rows(gpu_device):
    """Rows that keep nothing (DPR ranking), partial filters, passages without any edge (their mass
    leaves the iteration: the closed-form normalisation must account for it) and a seed that is a
    passage vertex."""
    import torch
    from hipporag_amd import dist as hd
    world, b = 4, 96
    kg, pass_bits, fact_bits, index = make_case(6000, 50000, 64, seed=321)
    # cut every edge of 40 passages: isolated (dangling) passage vertices
    lonely = kg.passage_vertex[::19][:40]
    keep = ~(np.isin(kg.src, lonely) | np.isin(kg.dst, lonely))
    from hipporag_amd.graph import build_csr
    import dataclasses
    kg = dataclasses.replace(kg, src=kg.src[keep], dst=kg.dst[keep], weight=kg.weight[keep],
                             csr=build_csr(kg.num_vertices, kg.src[keep], kg.dst[keep], kg.weight[keep]))
    subj = kg.subj_vertex.copy()
    subj[:50] = kg.passage_vertex[:50]          # facts whose subject is a passage vertex
    kg = dataclasses.replace(kg, subj_vertex=subj)
    a = oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight)
    index = dataclasses.replace(index, p=oracle.column_normalize(a), subj_vertex=subj)
    sidx = hd.shard_index(kg.csr, kg.passage_vertex, world, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=7)
    qf_bits[:10] = fact_bits[:10]               # top fact = one of the passage-subject facts
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=8)
    rng = np.random.default_rng(1)
    n_keep = rng.integers(0, 6, b)
    n_keep[:10] = 5

    def filt(idx, sc):
        cnt = torch.from_numpy(n_keep.astype(np.int32)).to(idx.device)
        return idx, sc, cnt                      # the first n entries of every row survive

    kw = dict(link_top_k=5, damping=0.5, passage_node_weight=0.05, ppr_iters=20, k=50)
    got = _run_shards(world, sidx, pass_bits, fact_bits, _bf16(qf_bits, gpu_device), _bf16(qp_bits, gpu_device),
                      kw, 2, gpu_device, 50, filter_fn=filt)
    # the one-call driver (hrag_shard_retrieve) on the same filtered batch: rows that kept nothing, partial rows, isolated
    # passages and the passage-vertex seed take the same path inside the library -- bit for bit the Python host loop
    nat = hd.run_local_shards(world, sidx, pass_bits, fact_bits, _bf16(qf_bits, gpu_device), _bf16(qp_bits, gpu_device),
                              kw, 2, gpu_device, 50, filt, native=True)
    for g_, n_ in zip(got, nat):
        np.testing.assert_array_equal(g_, n_)
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    pv = kg.passage_vertex
    for q in range(b):
        kept = got[0][q][: n_keep[q]].tolist()
        ref = oracle.retrieve_one(index, qf[q], qp[q], filter_fn=lambda cand, kept=kept: kept)
        assert bool(got[4][q] & 1) == ref.used_dpr == (n_keep[q] == 0), q
        assert not (got[4][q] & 8), q
        want_ids, want_sc = ref.sorted_doc_ids[:50], ref.sorted_doc_scores[:50]
        if ref.used_dpr:
            assert tie_aware_equal(got[2][q], want_ids, want_sc, abs_gap=3e-6), q
            np.testing.assert_allclose(got[3][q], want_sc, rtol=0, atol=3e-6)
        else:
            assert tie_aware_equal(got[2][q], want_ids, want_sc, rel_gap=2e-5), q
            want = ref.x[pv][got[2][q]]
            assert (np.abs(got[3][q] - want) / want).max() < 1e-5, q


def test_convergence_contract_on_shards_matches_the_single_gpu_engine(gpu_device):
    """ppr_tol on row shards (hrag_shard_ppr_begin / _est / _decide): every shard measures the relative update of ITS
    passages, the measures are all-reduced before each device-side decision, so all shards run the same extension
    stages -- and the result, the residual and the sweep count are those of the single-GPU engine, bit for bit.
    The tolerance is set low enough that extension stages do run."""
    import threading
    import torch
    from hipporag_amd import dist as hd
    from hipporag_amd.engine import HippoRAGEngine, ShardStages
    world, b, tol = 4, 130, 1e-7
    kg, pass_bits, fact_bits, index = make_case(12000, 120000, 128, seed=611)
    sidx = hd.shard_index(kg.csr, kg.passage_vertex, world, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
    qf_t = _bf16(synth.make_queries_np(fact_bits, b, seed=3)[0], gpu_device)
    qp_t = _bf16(synth.make_queries_np(pass_bits, b, seed=4)[0], gpu_device)
    kw = dict(link_top_k=5, damping=0.5, passage_node_weight=0.05, ppr_iters=20, k=100, ppr_tol=tol, ppr_max_iters=29)
    shared, results, errors = {}, [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(gpu_device)
            eng = hd.build_shard_engine(sidx, pass_bits, fact_bits, rank, max_batch=b, max_topk=100, sell_seg_len=64)
            rs = hd.ShardedRetriever(ShardStages(eng), hd.LocalComm(rank, world, shared), groups=2)
            idx, sc = rs.score_facts(qf_t, k=5)
            cnt = torch.full((b,), 5, dtype=torch.int32, device=gpu_device)
            out = rs.retrieve(qp_t, idx, sc, cnt, **kw)
            torch.cuda.synchronize()
            results[rank] = tuple(t.cpu().numpy() for t in out)
            shared["_barrier"].wait()
            eng.close()
        except Exception as exc:
            errors.append((rank, exc))
            try:
                shared["_barrier"].abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    for r in range(1, world):
        for a, w in zip(results[r], results[0]):
            np.testing.assert_array_equal(a, w)
    d_idx, d_sc, flags, resid, used = results[0]
    with HippoRAGEngine(sidx.csr, sidx.passage_vertex, pass_bits, fact_bits, sidx.subj_vertex, sidx.obj_vertex,
                        sidx.num_chunks, max_batch=b, max_topk=100, sell_seg_len=64) as eng:
        idx, sc = eng.score_facts(qf_t, k=5)
        one = eng.retrieve(qp_t, idx, sc, torch.full((b,), 5, dtype=torch.int32, device=gpu_device), **kw)
        torch.cuda.synchronize()
    assert int(used.max()) > 20                                           # extension stages did run
    np.testing.assert_array_equal(used, one.iters_used.cpu().numpy())
    np.testing.assert_array_equal(resid, one.residual.cpu().numpy())
    np.testing.assert_array_equal(flags, one.flags.cpu().numpy())
    np.testing.assert_array_equal(d_idx, one.doc_idx.cpu().numpy())
    np.testing.assert_array_equal(d_sc, one.doc_score.cpu().numpy())


def test_shards_at_small_damping_keep_inside_the_fp8_range(gpu_device):
    """Damping 0.3: damping^m per stage is below what a stage's e4m3 rounding puts back into the residual, and a static
    scale chain built on damping^m alone drifted out of the range (round 4, found by tools/soak_random.py).  The
    single-GPU engine measures its scales there; a row shard cannot (that maximum would be an all-reduce per boundary),
    its chain assumes max(damping^m, 0.09) per stage.  Sparse power-law graph, 2 shards, 130 queries: no saturation
    flag on either path and both within the parity bar of the same-count oracle."""
    import dataclasses
    import torch
    from hipporag_amd import dist as hd
    from hipporag_amd.engine import HippoRAGEngine
    world, b, damping, iters = 2, 130, 0.3, 16
    kg, pass_bits, fact_bits, index = make_case(12000, 36000, 96, seed=31506654, passage_frac=0.3, power_law=True)
    index = dataclasses.replace(index, damping=damping)
    sidx = hd.shard_index(kg.csr, kg.passage_vertex, world, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=3)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=4)
    qf_t, qp_t = _bf16(qf_bits, gpu_device), _bf16(qp_bits, gpu_device)
    kw = dict(link_top_k=5, damping=damping, passage_node_weight=0.05, ppr_iters=iters, k=100)
    got = _run_shards(world, sidx, pass_bits, fact_bits, qf_t, qp_t, kw, 1, gpu_device, 100)
    with HippoRAGEngine(sidx.csr, sidx.passage_vertex, pass_bits, fact_bits, sidx.subj_vertex, sidx.obj_vertex,
                        sidx.num_chunks, max_batch=b, max_topk=100) as eng:
        idx, sc = eng.score_facts(qf_t, k=5)
        out = eng.retrieve(qp_t, idx, sc, torch.full((b,), 5, dtype=torch.int32, device=gpu_device), **kw)
        torch.cuda.synchronize()
        one = tuple(t.cpu().numpy() for t in (out.doc_idx, out.doc_score, out.flags))
        assert eng.timings()["slab_width"] == 128
    assert np.all(got[4] == 0), np.unique(got[4])                 # the shards: no HRAG_FLAG_FP8_SATURATED
    assert np.all(one[2] == 0), np.unique(one[2])                 # the single-GPU engine (measured scales): none either
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    for q in list(range(0, b, 17)) + [b - 1]:
        ref = oracle.retrieve_one(index, qf[q], qp[q])
        full = ref.x[kg.passage_vertex]
        for ids, scs in ((got[2][q], got[3][q]), (one[0][q], one[1][q])):
            assert float((np.abs(scs - full[ids]) / full[ids]).max()) < 1e-5, q
            assert tie_aware_equal(ids, ref.sorted_doc_ids[:100], ref.sorted_doc_scores[:100], rel_gap=2e-5), q


def test_one_shard_in_several_exchange_groups_measures_its_scales_over_the_whole_batch(gpu_device):
    """An engine that owns EVERY row, driven through the shard entry points in two exchange groups (the world = 1 legs
    of `bench.py --gpus 1` under HRAG_FORCE_DIST, or a host that pipelines groups on one GPU), at damping 0.3 where
    the plain plan MEASURES its stage scales: every boundary step is launched once per group, and the scale of the
    stage after next must come from the maximum over ALL groups (round-4 advisor finding: it was taken from the last
    group's slots alone, so the other groups ran on a scale that was not theirs).  300 queries = a two-slab group
    (pair kernel) + a one-slab group; the result must be the single-call engine's, bit for bit, without a saturation
    flag."""
    import dataclasses
    import torch
    from hipporag_amd import dist as hd
    from hipporag_amd.engine import HippoRAGEngine, ShardStages
    b, damping, iters = 300, 0.3, 16
    kg, pass_bits, fact_bits, index = make_case(12000, 36000, 96, seed=31506654, passage_frac=0.3, power_law=True)
    index = dataclasses.replace(index, damping=damping)
    qf_bits, _ = synth.make_queries_np(fact_bits, b, seed=13)
    qp_bits, _ = synth.make_queries_np(pass_bits, b, seed=14)
    qf_t, qp_t = _bf16(qf_bits, gpu_device), _bf16(qp_bits, gpu_device)
    kw = dict(link_top_k=5, damping=damping, passage_node_weight=0.05, ppr_iters=iters, k=100)
    cnt = torch.full((b,), 5, dtype=torch.int32, device=gpu_device)
    with HippoRAGEngine(kg.csr, kg.passage_vertex, pass_bits, fact_bits, kg.subj_vertex, kg.obj_vertex,
                        kg.num_chunks, max_batch=b, max_topk=100) as eng:
        idx, sc = eng.score_facts(qf_t, k=5)
        out = eng.retrieve(qp_t, idx, sc, cnt, **kw)
        torch.cuda.synchronize()
        assert eng.timings()["slab_width"] == 128
        one = tuple(t.cpu().numpy() for t in (out.doc_idx, out.doc_score, out.flags))
        rs = hd.ShardedRetriever(ShardStages(eng), hd.TorchComm(0, 1), groups=2)
        lay = eng.shard_layout(b, 2)
        assert lay.n_groups == 2 and lay.n_slabs == 3
        i2, s2 = rs.score_facts(qf_t, k=5)
        got = rs.retrieve(qp_t, i2, s2, cnt, **kw)
        torch.cuda.synchronize()
        got = tuple(t.cpu().numpy() for t in got)
    assert np.all(one[2] == 0) and np.all(got[2] == 0), (np.unique(one[2]), np.unique(got[2]))
    np.testing.assert_array_equal(got[0], one[0])
    np.testing.assert_array_equal(got[1], one[1])
    qf, qp = bf16_bits_to_float(qf_bits), bf16_bits_to_float(qp_bits)
    for q in (0, 127, 128, 255, 256, 299):
        ref = oracle.retrieve_one(index, qf[q], qp[q])
        full = ref.x[kg.passage_vertex]
        assert float((np.abs(got[1][q] - full[got[0][q]]) / full[got[0][q]]).max()) < 1e-5, q


def test_the_one_call_drivers_equal_the_python_host_loop_at_world_1(gpu_device):
    """hrag_shard_score_facts_all / hrag_shard_retrieve (csrc/shard_driver.hip, round 6): the host loop of
    dist.ShardedRetriever run inside the library.  World 1 (no collective is called: the C side of the drivers alone --
    layout, step / group order, exchange bookkeeping, contract decisions, candidate merge) against the Python loop on the
    same engine: bit-identical ids, scores, flags, residuals, sweep counts; the two-process form with real collectives
    is tests/test_gpu_multi.py."""
    import torch
    from hipporag_amd import dist as hd, synth
    from hipporag_amd.engine import ShardStages
    from tests.helpers import make_case
    kg, pass_bits, fact_bits, _ = make_case(9000, 90000, 64, seed=515)
    sidx = hd.shard_index(kg.csr, kg.passage_vertex, 1, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
    b = 300

    def bf16(bits):
        return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).to(gpu_device).view(torch.bfloat16)

    qf, qp = bf16(synth.make_queries_np(fact_bits, b, seed=3)[0]), bf16(synth.make_queries_np(pass_bits, b, seed=4)[0])
    cnt = torch.full((b,), 5, dtype=torch.int32, device=gpu_device)
    seng = hd.build_shard_engine(sidx, pass_bits, fact_bits, 0, max_batch=b, max_topk=60)
    try:
        py = hd.ShardedRetriever(ShardStages(seng), hd.TorchComm(0, 1), groups=2)
        nat = hd.NativeShardedRetriever(ShardStages(seng), hd.TorchComm(0, 1), groups=2)
        i0, s0 = py.score_facts(qf, k=5)
        i1, s1 = nat.score_facts(qf, k=5)
        assert torch.equal(i0, i1) and torch.equal(s0, s1)
        for kw in (dict(ppr_iters=20, k=60), dict(ppr_iters=20, k=60, ppr_tol=1.5e-6, ppr_max_iters=29),
                   dict(ppr_iters=16, k=7, damping=0.4)):
            want = py.retrieve(qp, i0, s0, cnt, **kw)
            got = nat.retrieve(qp, i0, s0, cnt, **kw)
            torch.cuda.synchronize()
            assert len(want) == len(got)
            for g, w in zip(got, want):
                assert torch.equal(g, w), kw
        # a too-small workspace and a bad comm are refused with a message, not a crash
        from hipporag_amd._lib import HragError
        nat._ws[(b, 60)] = nat._ws[(b, 60)][:1024]
        with pytest.raises(HragError):
            nat.retrieve(qp, i0, s0, cnt, ppr_iters=20, k=60)
    finally:
        seng.close()


@pytest.mark.parametrize("world,b,groups,kw_extra", [(8, 256, 2, {}), (3, 130, 0, dict(ppr_iters=24)),
                                                      (4, 200, 2, dict(ppr_tol=1.5e-6, ppr_max_iters=29))])
def test_the_one_call_drivers_with_emulated_ranks_equal_the_python_host_loop(gpu_device, world, b, groups, kw_extra):
    """hrag_shard_retrieve with `world` shard threads on one device calling back into dist.LocalComm (every collective is
    a barrier on shared buffers): 8 / 3 / 4 ranks, one or two exchange groups, fixed count and the contract -- every
    rank's replicated result equals what the Python host loop (dist.ShardedRetriever) returns on the same shards."""
    from hipporag_amd import dist as hd
    kg, pass_bits, fact_bits, _ = make_case(12000, 120000, 64, seed=640 + world, power_law=(world == 3))
    sidx = hd.shard_index(kg.csr, kg.passage_vertex, world, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
    qf = _bf16(synth.make_queries_np(fact_bits, b, seed=3)[0], gpu_device)
    qp = _bf16(synth.make_queries_np(pass_bits, b, seed=4)[0], gpu_device)
    kw = dict(link_top_k=5, damping=0.5, passage_node_weight=0.05, ppr_iters=20, k=80)
    kw.update(kw_extra)
    want = hd.run_local_shards(world, sidx, pass_bits, fact_bits, qf, qp, kw, groups, gpu_device, 80)
    got = hd.run_local_shards(world, sidx, pass_bits, fact_bits, qf, qp, kw, groups, gpu_device, 80, native=True)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    assert np.all(got[4] == 0)


def test_a_failing_collective_leaves_no_exchange_open(gpu_device):
    """hrag_shard_retrieve returns an error in the middle of the sweeps (the contract's all-reduce fails while every
    exchange group has an exchange in flight): the driver waits for every exchange it had begun before it returns
    (csrc/shard_driver.hip `Drain`; include/hrag.h hrag_comm), so the host's collective handles are never left open.
    Shard 0 of a 2-shard index with a stub comm (no peer: the collectives do nothing -- the control flow is under test,
    not the numbers)."""
    import torch
    from hipporag_amd import dist as hd
    from hipporag_amd.engine import ShardStages
    kg, pass_bits, fact_bits, _ = make_case(9000, 90000, 64, seed=733)
    sidx = hd.shard_index(kg.csr, kg.passage_vertex, 2, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
    b = 130

    class StubComm:
        rank, world = 0, 2

        def __init__(self):
            self.reduces, self.begun, self.waited, self.fail_at = 0, 0, 0, None

        def all_reduce(self, t, op):
            self.reduces += 1
            if self.fail_at is not None and self.reduces >= self.fail_at:
                raise RuntimeError("injected collective failure")

        def all_gather(self, t):
            return [t, t]

        def exchange(self, buf, lay, group):
            self.begun += 1
            return ("handle", group)

        def wait(self, handle):
            if handle is not None:
                self.waited += 1

    qf = _bf16(synth.make_queries_np(fact_bits, b, seed=3)[0], gpu_device)
    qp = _bf16(synth.make_queries_np(pass_bits, b, seed=4)[0], gpu_device)
    cnt = torch.full((b,), 5, dtype=torch.int32, device=gpu_device)
    seng = hd.build_shard_engine(sidx, pass_bits, fact_bits, 0, max_batch=b, max_topk=40)
    try:
        comm = StubComm()
        nat = hd.NativeShardedRetriever(ShardStages(seng), comm, groups=2)
        idx, sc = nat.score_facts(qf, k=5)
        kw = dict(ppr_iters=20, k=40, ppr_tol=1.5e-6, ppr_max_iters=29, check_saturation=False)
        nat.retrieve(qp, idx, sc, cnt, **kw)                       # a clean call: every exchange begun is waited for
        torch.cuda.synchronize()
        assert comm.begun > 0 and comm.begun == comm.waited and not nat._pend
        before = comm.reduces
        comm.reduces, comm.begun, comm.waited = 0, 0, 0
        comm.fail_at = 5          # min, max, zmax, mass pass; the 5th all-reduce is the contract's measure, mid-sweeps
        with pytest.raises(RuntimeError, match="injected"):
            nat.retrieve(qp, idx, sc, cnt, **kw)
        torch.cuda.synchronize()
        assert before >= 5 and comm.begun >= 2                     # both groups had an exchange in flight
        assert comm.begun == comm.waited and not nat._pend, (comm.begun, comm.waited)
        comm.fail_at = None                                        # and the handle is usable afterwards
        nat.retrieve(qp, idx, sc, cnt, **kw)
        torch.cuda.synchronize()
    finally:
        seng.close()
