"""world_size-2 gloo tests of the row-sharded orchestration (hipporag_amd/dist.py) on CPU.

The HIP kernels cannot run here, so the per-rank compute is a numpy stand-in built from the
oracle that implements the same "stages" interface as hipporag_amd.engine.EngineStages; what is
under test is everything dist.py adds: shard plans, candidate merging with the global tie rule,
the passage-score all-gather, the per-sweep re-assembly of x (both collectives), the column-sum
all-reduce -- against the single-process oracle."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from hipporag_amd import synth
from hipporag_amd.dist import RowShardedRetriever, ShardPlan, balanced_row_shards, even_shards
from hipporag_amd.graph import bf16_bits_to_float
from tests.helpers import make_case

SEED_STRIDE = 32


class FakeStages:
    """CPU stand-in with the EngineStages interface (owned rows / embedding slices only)."""

    def __init__(self, index, rows, passages, facts, bc=4):
        self.ix, self.rows, self.pas, self.fac, self.bc = index, rows, passages, facts, bc
        self.v = index.num_vertices
        self.p_rows = index.p[rows[0]:rows[1]]                    # owned CSR rows
        self.pe = index.passage_emb[passages[0]:passages[1]].astype(np.float64)
        self.fe = index.fact_emb[facts[0]:facts[1]].astype(np.float64)

    def layout(self, b):
        return self.bc, -(-b // self.bc)

    def new_state(self, b):
        bc, ns = self.layout(b)
        return torch.zeros((ns, self.v, bc), dtype=torch.float32)

    def sim_scores(self, which, q):
        emb = self.fe if which == "facts" else self.pe
        return torch.from_numpy((q.double().numpy() @ emb.T).astype(np.float32))

    def topk(self, scores, k, idx_offset=0):
        s = scores.numpy()
        b, n = s.shape
        idx = np.full((b, k), -1, np.int32)
        val = np.zeros((b, k), np.float32)
        for r in range(b):
            o = oracle.topk_desc(s[r], k)
            idx[r, :len(o)] = o + idx_offset
            val[r, :len(o)] = s[r][o]
        mn = s.min(axis=1) if n else np.zeros(b, np.float32)
        mx = s.max(axis=1) if n else np.zeros(b, np.float32)
        return torch.from_numpy(idx), torch.from_numpy(val), torch.from_numpy(mn), torch.from_numpy(mx)

    def row_minmax(self, scores):
        s = scores.numpy()
        return torch.from_numpy(s.min(axis=1)), torch.from_numpy(s.max(axis=1))

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
            scores = np.zeros(len(self.ix.subj_vertex), np.float32)
            kept = kept_idx[q, :n].numpy()
            scores[kept] = kept_score[q, :n].numpy()
            ids, w = oracle.seed_weights(self.ix, scores, kept.tolist(), link_top_k)
            sv[q, :len(ids)], sw[q, :len(ids)], sc[q] = ids, w, len(ids)
        return torch.from_numpy(sv), torch.from_numpy(sw), torch.from_numpy(sc), torch.from_numpy(flags)

    def teleport(self, s_full, mn, mx, weight, flags):
        b = s_full.shape[0]
        bc, ns = self.layout(b)
        n_p = len(self.ix.passage_vertex)
        tele = np.zeros((ns, n_p, bc), np.float32)
        for q in range(b):
            if flags[q] & 1:
                continue
            norm = oracle.min_max_normalize(s_full[q].numpy())
            tele[q // bc, :, q % bc] = norm * np.float32(weight)
        return torch.from_numpy(tele)

    def _v_rows(self, tele, seeds, b):
        """teleport + seeds restricted to the owned rows: [ns, n_owned, bc] float64."""
        bc, ns = self.layout(b)
        lo, hi = self.rows
        v = np.zeros((ns, self.v, bc))
        v[:, self.ix.passage_vertex, :] = tele.numpy()
        sv, sw, sc = seeds
        for q in range(b):
            for j in range(int(sc[q])):
                v[q // bc, int(sv[q, j]), q % bc] += float(sw[q, j])
        return v[:, lo:hi, :]

    def ppr_init(self, tele, seeds, b, x):
        lo, hi = self.rows
        x[:, lo:hi, :] = torch.from_numpy(self._v_rows(tele, seeds, b).astype(np.float32))

    def ppr_step(self, tele, seeds, b, damping, x, y):
        lo, hi = self.rows
        v = self._v_rows(tele, seeds, b)
        xs = x.numpy().astype(np.float64)
        for s in range(xs.shape[0]):
            y[s, lo:hi, :] = torch.from_numpy((damping * (self.p_rows @ xs[s]) + (1 - damping) * v[s]).astype(np.float32))

    def colsum(self, x, b):
        lo, hi = self.rows
        bc, ns = self.layout(b)
        part = x[:, lo:hi, :].double().sum(dim=1)                 # [ns, bc]
        return part.reshape(-1)[:b].clone()

    def doc_scores(self, x, sums, b, s_full, mn, mx, flags):
        bc, ns = self.layout(b)
        pv = self.ix.passage_vertex
        out = np.zeros((b, len(pv)), np.float32)
        for q in range(b):
            if flags[q] & 1:
                out[q] = oracle.min_max_normalize(s_full[q].numpy())
            else:
                out[q] = (x[q // bc, pv, q % bc].double().numpy() / float(sums[q])).astype(np.float32)
        return torch.from_numpy(out)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, collective, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kg, pass_bits, fact_bits, index = make_case(900, 7200, 32, seed=13, power_law=True)
        plan = ShardPlan(balanced_row_shards(kg.csr.row_ptr, world), even_shards(kg.n_passages, world),
                         even_shards(kg.n_facts, world))
        st = FakeStages(index, plan.rows[rank], plan.passages[rank], plan.facts[rank])
        rs = RowShardedRetriever(st, plan, rank, world, collective=collective)
        b = 6
        qf = torch.from_numpy(bf16_bits_to_float(synth.make_queries_np(fact_bits, b, 1)[0]))
        qp = torch.from_numpy(bf16_bits_to_float(synth.make_queries_np(pass_bits, b, 2)[0]))
        idx, sc = rs.score_facts(qf, k=5)
        cnt = torch.full((b,), 5, dtype=torch.int32)
        cnt[4] = 0                                                # one DPR-fallback query
        doc_idx, doc_val, flags = rs.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=40)
        if rank == 0:
            for q in range(b):
                flt = (lambda cand: []) if q == 4 else None
                ref = oracle.retrieve_one(index, qf[q].numpy(), qp[q].numpy(), filter_fn=flt,
                                          ppr_mode="power", ppr_iters=20)
                np.testing.assert_array_equal(idx[q].numpy(), ref.fact_candidates)
                np.testing.assert_allclose(sc[q].numpy(), ref.fact_candidate_scores, rtol=0, atol=1e-6)
                assert bool(flags[q] & 1) == ref.used_dpr
                np.testing.assert_array_equal(doc_idx[q].numpy(), ref.sorted_doc_ids[:40])
                np.testing.assert_allclose(doc_val[q].numpy(), ref.sorted_doc_scores[:40], rtol=2e-6, atol=1e-7)
        # every rank must hold the same answer
        gathered = [torch.empty_like(doc_idx) for _ in range(world)]
        dist.all_gather(gathered, doc_idx)
        for g in gathered:
            assert torch.equal(g, doc_idx)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("collective", ["allgather", "allreduce"])
def test_row_sharded_world2_matches_oracle(collective):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, collective, ret), nprocs=world, join=True)
    assert dict(ret) == {0: 1, 1: 1}


def test_shard_plans_cover_everything():
    kg = synth.make_kg(5000, 50000, seed=2, power_law=True)
    for world in (1, 2, 3, 8):
        rows = balanced_row_shards(kg.csr.row_ptr, world)
        assert rows[0][0] == 0 and rows[-1][1] == kg.num_vertices
        assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
        nnz = [int(kg.csr.row_ptr[hi] - kg.csr.row_ptr[lo]) for lo, hi in rows]
        assert max(nnz) <= 1.6 * (kg.csr.nnz / world) + kg.csr.row_ptr[1:].max() * 0 + np.diff(kg.csr.row_ptr).max()
        ev = even_shards(kg.n_passages, world)
        assert ev[0][0] == 0 and ev[-1][1] == kg.n_passages and max(h - l for l, h in ev) - min(h - l for l, h in ev) <= 1
