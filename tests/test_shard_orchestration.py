"""CPU tests of the fp8-state row-shard orchestration (hipporag_amd/dist.py: shard_index, ShardedRetriever,
TorchComm over gloo with world_size 2, LocalComm with 4 in-process shards).

The HIP kernels cannot run here, so every shard's compute is a numpy stand-in with the interface of
hipporag_amd.engine.ShardStages (the hrag_shard_* entry points of include/hrag.h): same state-buffer
layout contract ([n_groups][V + 1][slabs_per_group][128 bytes], the owned rows of a group = one
contiguous block at rank * own_bytes), a plain fp32 power iteration as the sweep.  Under test is what
dist.py adds: the relabelling into equal-sized shards, the in-place all-gather of the owners' blocks per
exchange group (pipelined against the other groups' sweeps), the min / max / mass reductions, the
candidate merges with the global tie rule -- against the single-process oracle."""

import os
import socket
import threading
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from hipporag_amd import dist as hd
from hipporag_amd import synth
from hipporag_amd.graph import bf16_bits_to_float
from tests.helpers import make_case

SEED_STRIDE = 32
QPS = 32          # queries per 128-byte slab of the stand-in (fp32 state)


class FakeShardStages:
    def __init__(self, sidx, index, rank):
        self.sidx, self.ix, self.rank = sidx, index, rank
        rps = sidx.rows_per_shard
        self.v = sidx.num_vertices
        self.lo, self.hi = rank * rps, (rank + 1) * rps
        import scipy.sparse as sp
        c = sidx.csr
        p = sp.csr_matrix((c.val.astype(np.float64), c.col_idx, c.row_ptr), shape=(self.v, self.v))
        self.p_rows = p[self.lo:self.hi]
        self.p_lo, self.p_hi = sidx.passages[rank]
        self.f_lo, self.f_hi = sidx.facts[rank]
        self.pe = index.passage_emb[self.p_lo:self.p_hi].astype(np.float64)
        self.fe = index.fact_emb[self.f_lo:self.f_hi].astype(np.float64)
        self.pv_local = sidx.passage_vertex[self.p_lo:self.p_hi].astype(np.int64) - self.lo   # local rows
        assert self.pv_local.size == 0 or (self.pv_local.min() >= 0 and self.pv_local.max() < rps)
        # the oracle's seed arithmetic on the relabelled vertex ids
        import dataclasses
        self.ix_rel = dataclasses.replace(index, subj_vertex=sidx.subj_vertex, obj_vertex=sidx.obj_vertex,
                                          num_chunks=sidx.num_chunks, passage_vertex=sidx.passage_vertex, p=p)
        self.iso = np.asarray(c.col_sum == 0)

    # ---- layout contract of include/hrag.h
    def shard_layout(self, batch, groups=0):
        ns = -(-batch // QPS)
        spg = 1 if groups <= 0 else -(-ns // min(groups, ns))
        g = -(-ns // spg)
        gb = (self.v + 1) * spg * 128
        return SimpleNamespace(n_slabs=ns, n_groups=g, slabs_per_group=spg, state_bytes=g * gb, group_bytes=gb,
                               own_offset=self.lo * spg * 128, own_bytes=(self.hi - self.lo) * spg * 128)

    def new_state(self, lay):
        return torch.zeros((lay.state_bytes,), dtype=torch.uint8)

    def _view(self, buf):      # [G, V + 1, spg, 32] fp32 view of a state buffer
        l = self.lay
        return buf.view(torch.float32).view(l.n_groups, self.v + 1, l.slabs_per_group, QPS)

    # ---- similarity
    def topk(self, scores, k, idx_offset=0):
        s = scores.numpy()
        b, n = s.shape
        idx = np.full((b, k), -1, np.int32)
        val = np.zeros((b, k), np.float32)
        for r in range(b):
            o = oracle.topk_desc(s[r], k)
            idx[r, :len(o)] = o + idx_offset
            val[r, :len(o)] = s[r][o]
        return torch.from_numpy(idx), torch.from_numpy(val), None, None

    def shard_score_facts(self, q, k):
        s = (q.double().numpy() @ self.fe.T).astype(np.float32)
        idx, val, _, _ = self.topk(torch.from_numpy(s), k, idx_offset=self.f_lo)
        return idx, val, torch.from_numpy(s.min(1)), torch.from_numpy(s.max(1))

    def shard_passage_scores(self, q):
        self.s_local = (q.double().numpy() @ self.pe.T).astype(np.float32)
        return torch.from_numpy(self.s_local.min(1)), torch.from_numpy(self.s_local.max(1))

    def seeds(self, kept_idx, kept_score, kept_count, link_top_k):
        b = kept_idx.shape[0]
        sv = np.zeros((b, SEED_STRIDE), np.int32)
        sw = np.zeros((b, SEED_STRIDE), np.float32)
        sc = np.zeros(b, np.int32)
        flags = np.zeros(b, np.int32)
        for q in range(b):
            n = int(kept_count[q])
            if n == 0:
                flags[q] |= 1
                continue
            scores = np.zeros(len(self.ix_rel.subj_vertex), np.float32)
            kept = kept_idx[q, :n].numpy()
            scores[kept] = kept_score[q, :n].numpy()
            ids, w = oracle.seed_weights(self.ix_rel, scores, kept.tolist(), link_top_k)
            sv[q, :len(ids)], sw[q, :len(ids)], sc[q] = ids, w, len(ids)
        return torch.from_numpy(sv), torch.from_numpy(sw), torch.from_numpy(sc), torch.from_numpy(flags)

    def _prior(self, mn, mx, weight, flags):
        rng = (mx - mn).numpy()[:, None]
        nrm = np.where(rng == 0, 1.0, (self.s_local - mn.numpy()[:, None]) / np.where(rng == 0, 1, rng)).astype(np.float32)
        v = nrm * np.float32(weight)
        v[(flags.numpy() & 1) != 0] = 0
        return v                                                   # [B, p_rows]

    def shard_prior_stats(self, mn, mx, weight, flags):
        v = self._prior(mn, mx, weight, flags).astype(np.float64)
        iso_p = self.iso[self.lo + self.pv_local]
        return torch.zeros(mn.shape[0]), torch.from_numpy(np.concatenate([v.sum(1), v[:, iso_p].sum(1)]))

    def shard_ppr_begin(self, mn, mx, zmax, mass, weight, seeds, flags, damping, iters, n_groups, bufs):
        b = mn.shape[0]
        self.lay = self.shard_layout(b, n_groups)
        assert self.lay.n_groups == n_groups or n_groups <= 0
        self.bufs, self.iters, self.a, self.b = bufs, iters, damping, b
        sv, sw, sc = (t.numpy() for t in seeds)
        v = np.zeros((self.hi - self.lo, b))                         # owned rows only
        v[self.pv_local, :] = self._prior(mn, mx, weight, flags).T
        m_tot, m_iso = mass[:b].numpy().copy(), mass[b:].numpy().copy()
        for q in range(b):
            for j in range(int(sc[q])):
                g = int(sv[q, j])
                m_tot[q] += float(sw[q, j])
                if self.iso[g]:
                    m_iso[q] += float(sw[q, j])
                if self.lo <= g < self.hi:
                    v[g - self.lo, q] += float(sw[q, j])
        self.v_own = v
        be = 1.0 - damping
        m = m_tot.copy()
        for k in range(iters):                                       # the closed-form mass of csrc/ppr8.hip
            m = damping * (m - (m_iso if k == 0 else be * m_iso)) + be * m_tot
        self.mass = m
        self._store(bufs[0], range(self.lay.n_slabs), v)

    def _store(self, buf, slabs, rows):                              # rows: [n_own, B] -> owned rows of `slabs`
        view, spg = self._view(buf), self.lay.slabs_per_group
        for s in slabs:
            cols = slice(s * QPS, min((s + 1) * QPS, self.b))
            blk = np.zeros((self.hi - self.lo, QPS), np.float32)
            blk[:, : cols.stop - cols.start] = rows[:, cols]
            view[s // spg, self.lo:self.hi, s % spg, :] = torch.from_numpy(blk)

    def _load(self, buf, slabs):                                     # full x of `slabs`: [V, len(slabs) * 32]
        view, spg = self._view(buf), self.lay.slabs_per_group
        return np.concatenate([view[s // spg, : self.v, s % spg, :].numpy() for s in slabs], axis=1).astype(np.float64)

    def shard_ppr_sweep(self, i, g):
        spg = self.lay.slabs_per_group
        slabs = list(range(g * spg, min((g + 1) * spg, self.lay.n_slabs)))
        x = self._load(self.bufs[i % 2], slabs)
        cols = np.concatenate([np.arange(s * QPS, (s + 1) * QPS) for s in slabs])
        ok = cols < self.b
        v = np.zeros((self.hi - self.lo, len(cols)))
        v[:, ok] = self.v_own[:, cols[ok]]
        y = self.a * (self.p_rows @ x) + (1 - self.a) * v
        if i + 1 == self.iters:                                      # last sweep: only the local passage rows matter
            if not hasattr(self, "x_final") or self.x_final.shape[1] != self.lay.n_slabs * QPS:
                self.x_final = np.zeros((self.hi - self.lo, self.lay.n_slabs * QPS))
            self.x_final[:, cols] = y
            return -1
        full = np.zeros((self.hi - self.lo, self.lay.n_slabs * QPS))
        full[:, cols] = y
        out = (i + 1) % 2
        self._store_cols(self.bufs[out], slabs, full)
        return out

    def _store_cols(self, buf, slabs, full):
        view, spg = self._view(buf), self.lay.slabs_per_group
        for s in slabs:
            view[s // spg, self.lo:self.hi, s % spg, :] = torch.from_numpy(full[:, s * QPS:(s + 1) * QPS].astype(np.float32))

    def shard_finish(self, mn, mx, flags, k):
        b = self.b
        doc = np.zeros((b, self.p_hi - self.p_lo), np.float32)
        fl = flags.numpy()
        for q in range(b):
            if fl[q] & 1:
                rng = float(mx[q] - mn[q])
                doc[q] = np.ones_like(self.s_local[q]) if rng == 0 else (self.s_local[q] - float(mn[q])) / np.float32(rng)
            elif self.mass[q] > 0:
                doc[q] = (self.x_final[self.pv_local, q] / self.mass[q]).astype(np.float32)
            else:
                fl[q] |= 2
        idx, val, _, _ = self.topk(torch.from_numpy(doc), k, idx_offset=self.p_lo)
        return idx, val


def _problem(world):
    kg, pass_bits, fact_bits, index = make_case(1500, 12000, 32, seed=21, power_law=True)
    sidx = hd.shard_index(kg.csr, kg.passage_vertex, world, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
    b = 70                                                           # three 32-query slabs, the last one partial
    qf = torch.from_numpy(bf16_bits_to_float(synth.make_queries_np(fact_bits, b, 1)[0]))
    qp = torch.from_numpy(bf16_bits_to_float(synth.make_queries_np(pass_bits, b, 2)[0]))
    return kg, index, sidx, qf, qp, b


def _run_rank(rs, qf, qp, b):
    idx, sc = rs.score_facts(qf, k=5)
    cnt = torch.full((b,), 5, dtype=torch.int32)
    cnt[4] = 0                                                       # one DPR-fallback query
    cnt[9] = 2
    return (idx, sc) + tuple(rs.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=40))


def _check(index, qf, qp, b, idx, sc, doc_idx, doc_val, flags):
    for q in range(b):
        kept_n = 0 if q == 4 else 2 if q == 9 else 5
        flt = (lambda cand, n=kept_n: cand[:n])
        ref = oracle.retrieve_one(index, qf[q].numpy(), qp[q].numpy(), filter_fn=flt, ppr_mode="power", ppr_iters=20)
        np.testing.assert_array_equal(idx[q].numpy(), ref.fact_candidates)
        np.testing.assert_allclose(sc[q].numpy(), ref.fact_candidate_scores, rtol=0, atol=1e-6)
        assert bool(flags[q] & 1) == ref.used_dpr
        np.testing.assert_array_equal(doc_idx[q].numpy(), ref.sorted_doc_ids[:40])
        np.testing.assert_allclose(doc_val[q].numpy(), ref.sorted_doc_scores[:40], rtol=3e-6, atol=1e-7)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gloo_worker(rank, world, port, groups, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kg, index, sidx, qf, qp, b = _problem(world)
        rs = hd.ShardedRetriever(FakeShardStages(sidx, index, rank), hd.TorchComm(rank, world), groups=groups)
        out = _run_rank(rs, qf, qp, b)
        if rank == 0:
            _check(index, qf, qp, b, *out)
        gathered = [torch.empty_like(out[2]) for _ in range(world)]
        dist.all_gather(gathered, out[2])
        for g in gathered:
            assert torch.equal(g, out[2])                            # every rank holds the same answer
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("groups", [1, 2, 0])
def test_sharded_retriever_world2_gloo_matches_oracle(groups):
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_gloo_worker, args=(world, _free_port(), groups, ret), nprocs=world, join=True)
    assert dict(ret) == {0: 1, 1: 1}


def test_sharded_retriever_four_local_shards_match_oracle():
    """dist.LocalComm: the shards as threads of one process sharing the state buffers (the harness the GPU
    test tests/test_gpu_shard.py drives the real kernels with)."""
    world = 4
    kg, index, sidx, qf, qp, b = _problem(world)
    shared, results, errors = {}, [None] * world, []

    def worker(rank):
        try:
            rs = hd.ShardedRetriever(FakeShardStages(sidx, index, rank), hd.LocalComm(rank, world, shared), groups=2)
            results[rank] = _run_rank(rs, qf, qp, b)
        except Exception as exc:
            errors.append((rank, repr(exc)))
            shared["_barrier"].abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    _check(index, qf, qp, b, *results[0])
    for r in range(1, world):
        assert torch.equal(results[r][2], results[0][2]) and torch.equal(results[r][3], results[0][3])


def test_shard_index_is_an_isomorphic_balanced_relabelling():
    kg = synth.make_kg(12000, 120000, seed=5, power_law=True)
    for world in (1, 2, 3, 8):
        s = hd.shard_index(kg.csr, kg.passage_vertex, world, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
        rps = s.rows_per_shard
        assert s.num_vertices == world * rps >= kg.num_vertices
        deg_old, deg_new = np.diff(kg.csr.row_ptr), np.diff(s.csr.row_ptr)
        assert (deg_new[s.perm] == deg_old).all() and deg_new.sum() == deg_old.sum()
        np.testing.assert_allclose(s.csr.col_sum[s.perm], kg.csr.col_sum)
        # an edge (i, j, w) of the original is the edge (perm i, perm j, w) of the relabelled graph
        rows_old = np.repeat(np.arange(kg.num_vertices), deg_old)
        key_old = np.sort(s.perm[rows_old] * s.num_vertices + s.perm[kg.csr.col_idx])
        rows_new = np.repeat(np.arange(s.num_vertices), deg_new)
        key_new = rows_new * s.num_vertices + s.csr.col_idx
        np.testing.assert_array_equal(key_old, key_new)              # CSR of the new graph is sorted by (row, col)
        for g, (lo, hi) in enumerate(s.passages):                    # passage shard g lives on row shard g
            pv = s.passage_vertex[lo:hi]
            assert pv.min() >= g * rps and pv.max() < (g + 1) * rps
        nnz = [int(s.csr.row_ptr[(g + 1) * rps] - s.csr.row_ptr[g * rps]) for g in range(world)]
        assert max(nnz) <= 1.02 * (sum(nnz) / world) + 64, nnz
        np.testing.assert_array_equal(s.num_chunks[s.perm], kg.num_chunks)
        np.testing.assert_array_equal(s.subj_vertex, s.perm[kg.subj_vertex])


# ------------------------------------------------------------------------------------------ hybrid mode
class _FakePprEngine:
    """hipporag_amd.engine.HippoRAGEngine.retrieve_scored on the CPU: the oracle from given raw passage scores."""

    def __init__(self, index):
        self.ix = index

    def retrieve_scored(self, scores, kept_idx, kept_score, kept_count, *, link_top_k, damping, passage_node_weight,
                        ppr_iters, k, **_):
        import dataclasses
        ix = dataclasses.replace(self.ix, linking_top_k=link_top_k, damping=damping, passage_node_weight=passage_node_weight)
        b = scores.shape[0]
        d_idx, d_sc = np.full((b, k), -1, np.int32), np.zeros((b, k), np.float32)
        for i in range(b):
            n = int(kept_count[i])
            s = scores[i].numpy()
            by_p = oracle.min_max_normalize(s)
            if n == 0:
                order = oracle.topk_desc(by_p, k)
                d_idx[i, :len(order)], d_sc[i, :len(order)] = order, by_p[order]
                continue
            fs = np.zeros(len(ix.subj_vertex), np.float32)
            kept = kept_idx[i, :n].numpy()
            fs[kept] = kept_score[i, :n].numpy()
            sid, sw = oracle.seed_weights(ix, fs, kept.tolist(), link_top_k)
            ids, sc, _ = oracle.run_ppr(ix, oracle.reset_vector(ix, sid, sw, by_p), damping, "power", ppr_iters)
            d_idx[i, :min(k, len(ids))], d_sc[i, :min(k, len(ids))] = ids[:k], sc[:k]
        return SimpleNamespace(doc_idx=torch.from_numpy(d_idx), doc_score=torch.from_numpy(d_sc))


def _hybrid_rank(comm, sidx, index, rank, qf, qp, b):
    sim = FakeShardStages(sidx, index, rank)
    sim.e = SimpleNamespace(sim_scores=lambda which, q: torch.from_numpy((q.double().numpy() @ sim.pe.T).astype(np.float32)))
    hy = hd.HybridRetriever(sim, _FakePprEngine(index), comm, sidx.passages)
    idx, sc = hy.score_facts(qf, k=5)
    cnt = torch.full((b,), 5, dtype=torch.int32)
    cnt[1] = 0                                                        # a DPR-fallback query
    out = hy.retrieve(qp, idx, sc, cnt, link_top_k=5, damping=0.5, passage_node_weight=0.05, ppr_iters=40, k=40)
    return hy.my_rows(b), idx, cnt, out


def _check_hybrid(index, qf, qp, rows, cnt, out):
    for i, q in enumerate(range(rows.start, rows.stop)):
        flt = (lambda cand, n=int(cnt[q]): cand[:n])
        ref = oracle.retrieve_one(index, qf[q].float().numpy(), qp[q].float().numpy(), filter_fn=flt,
                                  ppr_mode="power", ppr_iters=40)
        np.testing.assert_array_equal(out.doc_idx[i].numpy(), ref.sorted_doc_ids[:40])
        np.testing.assert_allclose(out.doc_score[i].numpy(), ref.sorted_doc_scores[:40], rtol=3e-6, atol=1e-7)


def _gloo_hybrid_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kg, index, sidx, qf, qp, b = _problem(world)
        rows, idx, cnt, out = _hybrid_rank(hd.TorchComm(rank, world), sidx, index, rank, qf, qp, b)
        _check_hybrid(index, qf, qp, rows, cnt, out)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_hybrid_retriever_world2_gloo_matches_oracle():
    """dist.HybridRetriever over gloo (world 2): embeddings sharded, one all-to-all of passage-score rows (paired
    isend / irecv on gloo), PPR query-parallel -- every rank's queries against the single-process oracle."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_gloo_hybrid_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: 1, 1: 1}
