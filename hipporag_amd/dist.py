"""Multi-GPU modes of the retrieval hot path: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) / gloo in the CPU tests.  The reference has no distributed code at all
(zero NCCL / MPI call sites; src/hipporag/HippoRAG.py:459 is a serial loop over the
queries); all three modes are MI355X-native additions.

rowshard  (the layout BASELINE.json's north star names; `value` of `bench.py --gpus N` when parity-green):
          the corpus is sharded row-wise -- rank g owns the CSR rows, the passage and the fact embedding rows of ITS
          equal-sized shard of a relabelled index (shard_index) -- and the e4m3 PPR iterate is replicated.
            * phase A : all-gather of each rank's local top-k fact candidates + MIN / MAX all-reduce (2 * B floats);
            * phase B : MIN / MAX all-reduce of the passage scores, MAX / SUM of the prior statistics; then per PPR
                        sweep every rank computes ITS rows of the new iterate and the owners' blocks are exchanged over
                        xGMI, one collective per exchange group, overlapped with the sweep of the other group(s).
                        TorchComm(collective="allgather") -- in-place all_gather_into_tensor of the owned blocks, 1 byte
                        per vertex and query -- or "allreduce": the north star's literal form (every rank zeroes the
                        rows it does not own, all-reduce SUM over the bytes: disjoint supports, so the sum IS the
                        gather; twice the wire bytes, kept for comparison);
            * final   : all-gather + merge of the local top-k lists.
          Every output row is produced by exactly one rank from the same replicated iterate: bit-identical to the
          single-GPU engine whatever the world size (tests/test_gpu_shard.py).
hybrid    embeddings row-sharded, ONE all-to-all of passage-score rows, PPR query-parallel on a replicated graph
          (HybridRetriever): 300x fewer wire bytes than the row-sharded PPR.
replica   queries are independent (the reference loop carries no cross-query state), every rank holds the whole
          index and serves its own slice of the batch with the single-GPU engine.  No data-path collective.

The compute of a rank is behind a small "stages" interface (hipporag_amd.engine.ShardStages = the hrag_shard_* C
entry points); tests drive the same orchestration with a CPU stand-in over gloo (tests/test_shard_orchestration.py,
tests/test_dist_gloo.py).
"""

from __future__ import annotations

import json
import os
import sys
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np


def _td():
    import torch
    import torch.distributed as dist
    return torch, dist


def even_shards(n: int, world: int) -> List[Tuple[int, int]]:
    base, rem = divmod(n, world)
    out, lo = [], 0
    for g in range(world):
        hi = lo + base + (1 if g < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


# --------------------------------------------------------------------------------------------
# fp8-state row shards (hrag_shard_* entry points): equal-sized shards in an internal vertex order
# --------------------------------------------------------------------------------------------
@dataclass
class ShardedIndex:
    """The index relabelled so that shard g owns the vertex ids [g * rows_per_shard, (g + 1) * rows_per_shard):
    the passages of passage-embedding shard g (passage order), then entity vertices dealt round-robin by
    descending degree (every shard gets the same number of rows and ~the same number of matrix entries),
    then isolated padding vertices.  Passage POSITIONS (the ids the caller sees) are unchanged."""
    world: int
    rows_per_shard: int
    perm: np.ndarray                 # int64 [V_original]: original vertex id -> internal id
    csr: "object"                    # CSRGraph over world * rows_per_shard internal vertices
    passage_vertex: np.ndarray
    subj_vertex: Optional[np.ndarray]
    obj_vertex: Optional[np.ndarray]
    num_chunks: Optional[np.ndarray]
    passages: List[Tuple[int, int]]
    facts: List[Tuple[int, int]]

    @property
    def num_vertices(self) -> int:
        return self.world * self.rows_per_shard


def shard_index(csr, passage_vertex, world: int, subj_vertex=None, obj_vertex=None, num_chunks=None,
                n_facts: Optional[int] = None) -> ShardedIndex:
    from .graph import CSRGraph
    v = int(csr.num_vertices)
    pv = np.asarray(passage_vertex, dtype=np.int64)
    n_p = pv.shape[0]
    pshards = even_shards(n_p, world)
    deg = np.diff(np.asarray(csr.row_ptr, dtype=np.int64))
    is_pass = np.zeros(v, dtype=bool)
    is_pass[pv] = True
    ent = np.flatnonzero(~is_pass)
    ent = ent[np.argsort(-deg[ent], kind="stable")]               # heaviest first
    n_pass_of = np.array([hi - lo for lo, hi in pshards], dtype=np.int64)
    # 2 % of slack rows per shard: the balance below trades row counts for entry counts
    cap_ent = -(-ent.shape[0] // world) + max(1, ent.shape[0] // (50 * world))
    rps = int(n_pass_of.max()) + cap_ent
    perm = np.full(v, -1, dtype=np.int64)
    for g, (lo, hi) in enumerate(pshards):
        perm[pv[lo:hi]] = g * rps + np.arange(hi - lo)
    # longest-processing-time greedy: every entity (heaviest first) goes to the shard with the fewest matrix
    # entries so far that still has a free row; the remaining rows of a shard are isolated padding vertices
    import heapq
    load0 = [int(deg[pv[lo:hi]].sum()) for lo, hi in pshards]
    heap = [(load0[g], g) for g in range(world)]
    heapq.heapify(heap)
    count = [0] * world
    shard = np.empty(ent.shape[0], dtype=np.int64)
    local = np.empty(ent.shape[0], dtype=np.int64)
    ent_deg = deg[ent].tolist()
    for i, d in enumerate(ent_deg):
        ld, g = heapq.heappop(heap)
        shard[i], local[i] = g, count[g]
        count[g] += 1
        if count[g] < cap_ent:
            heapq.heappush(heap, (ld + d, g))
    perm[ent] = shard * rps + n_pass_of[shard] + local
    assert (perm >= 0).all() and np.unique(perm).shape[0] == v
    v_pad = world * rps
    rows = np.repeat(np.arange(v, dtype=np.int64), deg)
    new_r, new_c = perm[rows], perm[np.asarray(csr.col_idx, dtype=np.int64)]
    order = np.argsort(new_r * v_pad + new_c, kind="stable")
    row_ptr = np.zeros(v_pad + 1, dtype=np.int64)
    np.cumsum(np.bincount(new_r, minlength=v_pad), out=row_ptr[1:])
    col_sum = None
    if csr.col_sum is not None:
        col_sum = np.zeros(v_pad, dtype=np.float64)
        col_sum[perm] = csr.col_sum
    new_csr = CSRGraph(v_pad, row_ptr.astype(np.int32), new_c[order].astype(np.int32),
                       np.asarray(csr.val)[order], np.asarray(csr.raw)[order], col_sum)

    def relabel(a):
        if a is None:
            return None
        a = np.asarray(a, dtype=np.int64)
        return np.where(a >= 0, perm[np.clip(a, 0, v - 1)], -1).astype(np.int32)

    nc = None
    if num_chunks is not None:
        nc = np.zeros(v_pad, dtype=np.int32)
        nc[perm] = np.asarray(num_chunks, dtype=np.int32)
    nf = int(n_facts if n_facts is not None else (len(subj_vertex) if subj_vertex is not None else 0))
    return ShardedIndex(world, rps, perm, new_csr, perm[pv].astype(np.int32), relabel(subj_vertex),
                        relabel(obj_vertex), nc, pshards, even_shards(nf, world))


def build_shard_engine(sidx: ShardedIndex, pass_emb, fact_emb, rank: int, max_batch: int, max_topk: int,
                       flags: int = 0, sell_seg_len: int = 0):
    """The engine of shard `rank` of a ShardedIndex (full embedding matrices are sliced here)."""
    from .engine import HippoRAGEngine
    rps = sidx.rows_per_shard
    p_lo, p_hi = sidx.passages[rank]
    f_lo, f_hi = sidx.facts[rank]
    has_facts = fact_emb is not None and sidx.subj_vertex is not None
    return HippoRAGEngine(sidx.csr.rows(rank * rps, (rank + 1) * rps), sidx.passage_vertex, pass_emb[p_lo:p_hi],
                          fact_emb[f_lo:f_hi] if has_facts else None, sidx.subj_vertex if has_facts else None,
                          sidx.obj_vertex if has_facts else None, sidx.num_chunks if has_facts else None,
                          max_batch=max_batch, max_topk=max_topk, row_offset=rank * rps, passage_offset=p_lo,
                          fact_offset=f_lo, n_passages=len(sidx.passage_vertex),
                          n_facts=sidx.facts[-1][1] if has_facts else None, flags=flags, sell_seg_len=sell_seg_len)


class TorchComm:
    """The exchange steps of the row-sharded path over torch.distributed (nccl = RCCL over xGMI; gloo in
    the CPU tests).  State exchange = ONE in-place collective per exchange group: the owned rows of a group
    are one contiguous block at rank * own_bytes.  collective="allgather" (default): all_gather_into_tensor of the
    owned blocks; "allreduce": BASELINE.json's literal wording -- every rank zeroes the blocks it does not own and
    the group region is all-reduced (SUM over uint8: the supports are disjoint, so no byte ever adds to another
    and the sum is the gather, bit for bit) -- twice the wire bytes of the all-gather, kept for comparison."""

    def __init__(self, rank: int, world: int, group=None, collective: str = "allgather"):
        if collective not in ("allgather", "allreduce"):
            raise ValueError(f"collective must be 'allgather' or 'allreduce', not {collective!r}")
        self.rank, self.world, self.group, self.collective = rank, world, group, collective

    def all_reduce(self, t, op: str):
        torch, dist = _td()
        if self.world > 1:
            dist.all_reduce(t, op={"min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX,
                                   "sum": dist.ReduceOp.SUM}[op], group=self.group)
        return t

    def all_gather(self, t):
        torch, dist = _td()
        if self.world == 1:
            return [t]
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return out

    def all_to_all(self, parts, shapes):
        """parts[d] goes to rank d; returns the list of what every rank sent here (shapes[s] = shape of rank s's
        part).  RCCL: one all_to_all; gloo (CPU tests) has none: paired isend / irecv."""
        torch, dist = _td()
        if self.world == 1:
            return [parts[0]]
        out = [torch.empty(tuple(shapes[s]), dtype=parts[0].dtype, device=parts[0].device) for s in range(self.world)]
        parts = [p.contiguous() for p in parts]
        if dist.get_backend(self.group) == "gloo":
            out[self.rank].copy_(parts[self.rank])
            reqs = []
            for d in range(self.world):
                if d != self.rank:
                    reqs.append(dist.isend(parts[d], self._global(d), group=self.group))
                    reqs.append(dist.irecv(out[d], self._global(d), group=self.group))
            for r in reqs:
                r.wait()
        else:
            dist.all_to_all(out, parts, group=self.group)
        return out

    def _global(self, g: int) -> int:
        torch, dist = _td()
        return g if self.group is None else dist.get_global_rank(self.group, g)

    def exchange(self, buf, lay, g: int):
        """Start the all-gather of exchange group g of state buffer `buf` (uint8 [state_bytes]); returns a
        handle for wait().  The collective is ordered after everything enqueued on the current stream."""
        torch, dist = _td()
        if self.world == 1:
            return None
        if lay.own_offset != self.rank * lay.own_bytes:
            raise ValueError("state exchange needs equal-sized row shards in rank order (dist.shard_index)")
        if self.world * lay.own_bytes + lay.slabs_per_group * 128 > lay.group_bytes:
            # the gathered region must end before the group's last row (row V, which stays zero: the target of the
            # masked-out gathers of the first sweep)
            raise ValueError("state exchange would overwrite the zero row: shards do not tile [0, V)")
        base = g * lay.group_bytes
        out = buf[base: base + self.world * lay.own_bytes]
        if self.collective == "allreduce":
            out[: lay.own_offset].zero_()
            out[lay.own_offset + lay.own_bytes:].zero_()
            return dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        inp = buf[base + lay.own_offset: base + lay.own_offset + lay.own_bytes]
        return dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True)

    def wait(self, handle):
        if handle is not None:
            handle.wait()          # the current stream waits for the collective; the host does not block


class LocalComm:
    """All `world` shards inside ONE process on ONE device, one thread per shard, meeting at barriers:
    the emulated gather SURVEY.md 8(e) asks for ("8 shards sequentially on one GPU must reproduce the
    single-GPU result").  The shards share the three state buffers (LocalComm.shared_buffers), so the
    state exchange is a barrier: every owner has enqueued its rows on the (common) stream before
    anybody enqueues the next sweep."""

    def __init__(self, rank: int, world: int, shared: dict):
        import threading
        self.rank, self.world, self.sh = rank, world, shared
        with shared.setdefault("_lock", threading.Lock()):
            if "_barrier" not in shared:
                shared["_barrier"] = threading.Barrier(world)
                shared["_slots"] = [None] * world

    def _barrier(self):
        self.sh["_barrier"].wait()

    def _collect(self, t):
        self.sh["_slots"][self.rank] = t
        self._barrier()
        parts = list(self.sh["_slots"])
        self._barrier()
        return parts

    def all_reduce(self, t, op: str):
        torch = _td()[0]
        parts = self._collect(t.clone())
        st = torch.stack(parts)
        # rank order: the same summation order on every shard (and in TorchComm's ring for world = 2)
        t.copy_({"min": lambda: st.min(0).values, "max": lambda: st.max(0).values, "sum": lambda: st.sum(0)}[op]())
        return t

    def all_gather(self, t):
        return self._collect(t)

    def all_to_all(self, parts, shapes):
        sent = self._collect(list(parts))            # sent[s][d] = what rank s sends to rank d
        return [sent[s][self.rank] for s in range(self.world)]

    def shared_buffers(self, key, make):
        if self.rank == 0:
            self.sh[key] = make()
        self._barrier()
        bufs = self.sh[key]
        self._barrier()
        return bufs

    def exchange(self, buf, lay, g: int):
        self._barrier()
        return None

    def wait(self, handle):
        pass


def merge_ranked(idx_parts, val_parts, k: int, topk):
    """Merge per-shard top-k lists (each sorted score desc, id desc; shard id ranges ascending in rank
    order) into the global top-k under the same rule.  Reversed and concatenated in rank order, "later
    position" == "larger (score, id)" among equal scores, so the library's positional tie rule
    reproduces the global order exactly.  Returns (ids int32 [B, k], values [B, k], -1 / 0 beyond)."""
    torch = _td()[0]
    cand_idx = torch.cat([t.flip(1) for t in idx_parts], dim=1).contiguous()
    cand_val = torch.cat([t.flip(1) for t in val_parts], dim=1)
    cand_val = torch.where(cand_idx < 0, torch.full_like(cand_val, float("-inf")), cand_val).contiguous()
    pos, top_val, _, _ = topk(cand_val, k)
    top_idx = torch.gather(cand_idx, 1, pos.clamp(min=0).long())
    top_idx = torch.where(pos < 0, torch.full_like(top_idx, -1), top_idx).to(torch.int32)
    top_val = torch.where(top_idx < 0, torch.zeros_like(top_val), top_val)
    return top_idx, top_val


class ShardedRetriever:
    """The hot path over fp8-state row shards: phase A / phase B with the exchange steps of
    include/hrag.h's hrag_shard_* section.  `stages` = the shard's engine (hipporag_amd.engine.ShardStages)
    or a CPU stand-in with the same methods (tests)."""

    def __init__(self, stages, comm, groups: int = 2):
        self.st, self.comm, self.groups = stages, comm, groups
        self._bufs = {}

    def _state(self, batch: int):
        if batch not in self._bufs:
            lay = self.st.shard_layout(batch, self.groups)
            make = lambda: [self.st.new_state(lay) for _ in range(3)]
            bufs = self.comm.shared_buffers(("state", batch), make) if hasattr(self.comm, "shared_buffers") else make()
            self._bufs[batch] = (lay, bufs)
        return self._bufs[batch]

    def score_facts(self, q_fact, k: int = 5):
        """Global (fact ids int32 [B, k], min-max normalised scores fp32 [B, k]), replicated."""
        torch = _td()[0]
        c = self.comm
        idx, val, mn, mx = self.st.shard_score_facts(q_fact, k)
        idx_all, val_all = c.all_gather(idx), c.all_gather(val)
        c.all_reduce(mn, "min")
        c.all_reduce(mx, "max")
        top_idx, top_val = merge_ranked(idx_all, val_all, k, self.st.topk)
        rng = (mx - mn).unsqueeze(1)
        norm = torch.where(rng == 0, torch.ones_like(top_val), (top_val - mn.unsqueeze(1)) / rng)   # misc_utils.py:130-139
        return top_idx, torch.where(top_idx < 0, torch.zeros_like(norm), norm)

    def retrieve(self, q_pass, kept_idx, kept_score, kept_count, *, link_top_k=5, damping=0.5,
                 passage_node_weight=0.05, ppr_iters=20, k=200, ppr_tol=0.0, ppr_max_iters=0,
                 check_saturation=True):
        """Returns (doc ids, doc scores, flags) -- and with ppr_tol > 0 additionally (residual, sweeps used): the
        convergence contract of hrag_retrieve on the shards.  Every shard measures the relative update of ITS
        passages; the measures are all-reduced (MAX) before every decision, so all shards run the same steps.
        check_saturation: the row shards have no wider state to repeat a batch on, so a batch that raised
        HRAG_FLAG_FP8_SATURATED (a violated scale bound: never on valid inputs) raises here instead of handing
        clipped scores on (include/hrag.h: "never return clipped scores"); it costs one host read per batch --
        a timing loop passes False and checks the flags afterwards."""
        st, c = self.st, self.comm
        b = q_pass.shape[0]
        lay, bufs = self._state(b)
        mn, mx = st.shard_passage_scores(q_pass)
        c.all_reduce(mn, "min")
        c.all_reduce(mx, "max")
        sv, sw, sc, flags = st.seeds(kept_idx, kept_score, kept_count, link_top_k)
        zmax, mass = st.shard_prior_stats(mn, mx, passage_node_weight, flags)
        c.all_reduce(zmax, "max")
        c.all_reduce(mass, "sum")
        contract = ppr_tol > 0
        if contract:
            n_steps = st.shard_ppr_begin(mn, mx, zmax, mass, passage_node_weight, (sv, sw, sc), flags, damping, ppr_iters,
                                         lay.n_groups, bufs, ppr_tol, ppr_max_iters)
        else:
            st.shard_ppr_begin(mn, mx, zmax, mass, passage_node_weight, (sv, sw, sc), flags, damping, ppr_iters,
                               lay.n_groups, bufs)
            n_steps = ppr_iters
        # group g's exchange overlaps with the sweep of the other groups: a sweep of group g only waits for
        # group g's previous exchange
        pend = [c.exchange(bufs[0], lay, g) for g in range(lay.n_groups)]
        est_global = False      # the engine's est already holds the all-reduced measure of the last final sweep
        for i in range(n_steps):
            ck = False
            for g in range(lay.n_groups):
                c.wait(pend[g])
                if contract:
                    xb, ck = st.shard_ppr_step(i, g)
                else:
                    xb = st.shard_ppr_sweep(i, g)
                pend[g] = c.exchange(bufs[xb], lay, g) if xb >= 0 else None
            est_global = False
            if ck:      # a final sweep that measured: the batch's residual over ALL passages, then the decision
                st.shard_ppr_est(c.all_reduce(st.shard_ppr_est(), "max"))
                st.shard_ppr_decide(i)
                est_global = True
                # the decision is the same on every shard (the measure was all-reduced).  Once it closes the gate of
                # the next step every later step is closed as well: stop issuing them -- each one would still cost a
                # full-state exchange per group for buffers nobody wrote (one host read per decision, <= 4 per batch)
                if not st.shard_ppr_gate_open(i + 1):
                    break
        for h in pend:          # an exchange started by the last step that ran
            c.wait(h)
        if contract:
            if not est_global:
                st.shard_ppr_est(c.all_reduce(st.shard_ppr_est(), "max"))
            idx, val, resid, used = st.shard_finish(mn, mx, flags, k, True)
        else:
            idx, val = st.shard_finish(mn, mx, flags, k)
        top_idx, top_val = merge_ranked(c.all_gather(idx), c.all_gather(val), k, st.topk)
        sat = (flags & 8).contiguous()               # raised on the shard that owns the row that saturated
        c.all_reduce(sat, "max")
        if check_saturation and bool(sat.any()):
            raise RuntimeError("fp8 PPR state saturated on a row shard (HRAG_FLAG_FP8_SATURATED): a static scale bound "
                               "was violated; the scores of the flagged queries are not trustworthy and the row-sharded "
                               "path has no wider state to repeat them on -- use the replica / hybrid mode for this batch")
        if contract:
            return top_idx, top_val, flags | sat, resid, used
        return top_idx, top_val, flags | sat


class HybridRetriever:
    """The hybrid multi-GPU mode SURVEY.md 8(e) ends on: the EMBEDDINGS are row-sharded over the GPUs (the part of
    the index that grows with the corpus: configs[4] holds 20 GB of them), the PPR runs QUERY-PARALLEL on a replicated
    graph (164 MB of CSR at configs[3]) with no exchange at all.

      phase A   every rank scores ALL queries of the global batch against ITS fact rows (local top-k) -> all-gather of
                the candidates + MIN / MAX all-reduce -> the same merge as the row-sharded mode (replicated result);
      phase B   every rank scores ALL queries against ITS passage rows [B, Np / N] -> ONE all-to-all hands rank r the
                score rows of ITS B / N queries over all passages [B / N, Np] -> hrag_retrieve_scored on those queries.

    Wire per global batch: B * Np * 4 bytes in total for the all-to-all ((N - 1) / N of it crosses links) plus the
    candidate lists -- 0.45 GB at configs[3] (B = 1024) against 20 sweeps x 0.91 GB RECEIVED PER GPU of the
    row-sharded PPR (146 GB in total): a factor 300.  Every passage score is the same MFMA chain whatever matrix slice
    it is computed from, so the result is bit-identical to the single-GPU engine on the same queries.
    sim = ShardStages of this rank's shard engine (dist.build_shard_engine: its embedding shards are what is used),
    ppr = a HippoRAGEngine WITHOUT embeddings over the whole graph, pshards = the passage ranges of the ranks."""

    def __init__(self, sim, ppr, comm, pshards):
        self.sim, self.ppr, self.comm, self.pshards = sim, ppr, comm, list(pshards)

    def score_facts(self, q_fact, k: int = 5):
        torch = _td()[0]
        c = self.comm
        idx, val, mn, mx = self.sim.shard_score_facts(q_fact, k)
        idx_all, val_all = c.all_gather(idx), c.all_gather(val)
        c.all_reduce(mn, "min")
        c.all_reduce(mx, "max")
        top_idx, top_val = merge_ranked(idx_all, val_all, k, self.sim.topk)
        rng = (mx - mn).unsqueeze(1)
        norm = torch.where(rng == 0, torch.ones_like(top_val), (top_val - mn.unsqueeze(1)) / rng)   # misc_utils.py:130-139
        return top_idx, torch.where(top_idx < 0, torch.zeros_like(norm), norm)

    def my_rows(self, batch: int):
        w, r = self.comm.world, self.comm.rank
        if batch % w:
            raise ValueError(f"the global batch ({batch}) must be a multiple of the world size ({w})")
        return slice(r * (batch // w), (r + 1) * (batch // w))

    def retrieve(self, q_pass, kept_idx, kept_score, kept_count, **kw):
        """q_pass / kept_*: the GLOBAL batch (replicated); returns this rank's RetrieveOutput for its B / N queries."""
        torch = _td()[0]
        c, w = self.comm, self.comm.world
        b = q_pass.shape[0]
        mine = self.my_rows(b)
        bn = b // w
        s_local = self.sim.e.sim_scores("passages", q_pass)                     # [B, Np_local]
        parts = [s_local[d * bn:(d + 1) * bn] for d in range(w)]
        got = c.all_to_all(parts, [(bn, hi - lo) for lo, hi in self.pshards])
        scores = torch.cat(got, dim=1).contiguous()                            # [B / N, Np] in passage order
        return self.ppr.retrieve_scored(scores, kept_idx[mine], kept_score[mine], kept_count[mine], **kw)


def run_local_shards(world: int, sidx: "ShardedIndex", pass_emb, fact_emb, q_fact, q_pass, retrieve_kw: dict,
                     groups: int, device, max_topk: int, filter_fn=None, timings: Optional[dict] = None,
                     sell_seg_len: int = 0):
    """All `world` shards of `sidx` as threads of THIS process on ONE device, meeting at barriers (LocalComm) and
    sharing the three e4m3 state buffers: the emulated gather SURVEY.md 8(e) prescribes.  Returns rank 0's
    (fact idx, fact score, doc idx, doc score, flags) as numpy arrays after checking that every rank computed the
    same replicated result.  Used by tests/test_gpu_shard.py and by `bench.py --config cfg4local` (the parity-checked
    form of BASELINE configs[3] when only one GPU is at hand)."""
    import threading
    torch = _td()[0]
    from .engine import ShardStages
    shared, results, errors = {}, [None] * world, []
    b = q_fact.shape[0]

    def worker(rank):
        try:
            torch.cuda.set_device(device)
            eng = build_shard_engine(sidx, pass_emb, fact_emb, rank, max_batch=b, max_topk=max_topk,
                                     sell_seg_len=sell_seg_len)
            rs = ShardedRetriever(ShardStages(eng), LocalComm(rank, world, shared), groups=groups)
            torch.cuda.synchronize()
            shared["_barrier"].wait()
            t0 = time.perf_counter()
            idx, sc = rs.score_facts(q_fact, k=5)
            cnt = torch.full((b,), 5, dtype=torch.int32, device=device)
            if filter_fn is not None:
                idx, sc, cnt = filter_fn(idx, sc)
            d_idx, d_sc, flags = rs.retrieve(q_pass, idx, sc, cnt, **retrieve_kw)[:3]   # ppr_tol > 0: + (residual, sweeps)
            torch.cuda.synchronize()
            if timings is not None and rank == 0:
                timings["wall_s_all_shards_on_one_device"] = time.perf_counter() - t0
            results[rank] = tuple(t.cpu().numpy() for t in (idx, sc, d_idx, d_sc, flags))
            shared["_barrier"].wait()          # nobody frees its engine while another shard still runs
            eng.close()
        except Exception as exc:               # a dead shard must not leave the others at a barrier for ever
            errors.append((rank, exc))
            try:
                shared["_barrier"].abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=1200)
    if errors:
        raise RuntimeError(f"shard threads failed: {errors}")
    for r in range(1, world):
        for a, w in zip(results[r], results[0]):
            np.testing.assert_array_equal(a, w)
    return results[0]


def build_ppr_engine(graph, passage_vertex, subj_vertex, obj_vertex, num_chunks, dim: int, max_batch: int,
                     max_topk: int, flags: int = 0):
    """The PPR side of the hybrid mode: the whole graph + the fact lookup arrays, NO embeddings."""
    from .engine import HippoRAGEngine
    return HippoRAGEngine(graph, passage_vertex, None, None, subj_vertex, obj_vertex, num_chunks, dim=dim,
                          max_batch=max_batch, max_topk=max_topk, flags=flags)


def run_local_hybrid(world: int, kg_arrays: dict, sidx: "ShardedIndex", pass_emb, fact_emb, q_fact, q_pass,
                     retrieve_kw: dict, device, max_topk: int, timings: Optional[dict] = None):
    """The hybrid mode with all `world` ranks as threads of this process on ONE device (LocalComm).  kg_arrays: the
    ORIGINAL index (csr, passage_vertex, subj_vertex, obj_vertex, num_chunks) for the PPR engines.  Returns, in
    query order, (fact idx, fact score) of the global batch and the concatenated per-rank (doc idx, doc score, flags)."""
    import threading
    torch = _td()[0]
    from .engine import ShardStages
    shared, results, errors = {}, [None] * world, []
    b = q_fact.shape[0]
    bn = b // world

    def worker(rank):
        try:
            torch.cuda.set_device(device)
            sim = build_shard_engine(sidx, pass_emb, fact_emb, rank, max_batch=b, max_topk=max_topk)
            ppr = build_ppr_engine(kg_arrays["csr"], kg_arrays["passage_vertex"], kg_arrays["subj_vertex"],
                                   kg_arrays["obj_vertex"], kg_arrays["num_chunks"], sim.dim, max_batch=bn,
                                   max_topk=max_topk)
            hy = HybridRetriever(ShardStages(sim), ppr, LocalComm(rank, world, shared), sidx.passages)
            torch.cuda.synchronize()
            shared["_barrier"].wait()
            t0 = time.perf_counter()
            idx, sc = hy.score_facts(q_fact, k=5)
            cnt = torch.full((b,), 5, dtype=torch.int32, device=device)
            out = hy.retrieve(q_pass, idx, sc, cnt, **retrieve_kw)
            torch.cuda.synchronize()
            if timings is not None and rank == 0:
                timings["wall_s_all_ranks_on_one_device"] = time.perf_counter() - t0
            results[rank] = tuple(t.cpu().numpy() for t in (idx, sc, out.doc_idx, out.doc_score, out.flags))
            shared["_barrier"].wait()
            sim.close()
            ppr.close()
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
        t.join(timeout=1200)
    if errors:
        raise RuntimeError(f"hybrid threads failed: {errors}")
    for r in range(1, world):
        np.testing.assert_array_equal(results[r][0], results[0][0])     # the merged fact candidates are replicated
        np.testing.assert_array_equal(results[r][1], results[0][1])
    return (results[0][0], results[0][1], np.concatenate([r[2] for r in results]),
            np.concatenate([r[3] for r in results]), np.concatenate([r[4] for r in results]))


# --------------------------------------------------------------------------------------------
# bench.py --gpus N (N > 1)
# --------------------------------------------------------------------------------------------
def pick_value_leg(mode: str, hybrid, rowshard) -> str:
    """Which leg of an N > 1 run becomes `value`: a leg counts only when it produced a rate AND its parity checks
    (bit / tolerance identity with the single-GPU engine on every rank; the fp64 oracle on rank 0) are green.
    auto: the ROW-SHARDED leg first -- the layout BASELINE.json's north star names and SURVEY.md 8(e) makes the primary
    figure -- then the hybrid leg (the better engineering: no per-sweep collective; always printed beside it as
    `value_hybrid`), the replica leg (which shards nothing) only when neither is green."""
    def green(leg):
        return isinstance(leg, dict) and "value" in leg and bool(leg.get("parity", {}).get("ok"))
    if mode in ("rowshard", "auto") and green(rowshard):
        return "rowshard"
    if mode in ("hybrid", "auto") and green(hybrid):
        return "hybrid"
    return "replica"            # mode "replica", or no green leg of the kind that was asked for


class OracleProbe:
    """The fp64 CPU oracle beside the N > 1 legs (rank 0 only; round-4 review: the legs were only ever compared with the
    single-GPU engine).  Built lazily -- host fp32 copies of the embeddings + the column-normalised matrix -- and shared
    by the legs; check() compares a leg's ranked ids / scores for a few queries of the global batch."""

    def __init__(self, kg, fact_emb, pass_emb):
        self.kg, self.fact_emb, self.pass_emb, self.index = kg, fact_emb, pass_emb, None

    def _build(self):
        import oracle
        kg = self.kg
        a = oracle.build_symmetric_csr(kg.num_vertices, kg.src, kg.dst, kg.weight)
        self.index = oracle.RefIndex(fact_emb=self.fact_emb.float().cpu().numpy(), passage_emb=self.pass_emb.float().cpu().numpy(),
                                     subj_vertex=kg.subj_vertex, obj_vertex=kg.obj_vertex, num_chunks=kg.num_chunks,
                                     passage_vertex=kg.passage_vertex, p=oracle.column_normalize(a))

    def check(self, qf, qp, doc_idx, doc_score, rows):
        """qf / qp: query tensors of the batch; doc_idx / doc_score: the leg's result rows for the same batch positions;
        rows: the positions to check.  Returns the parity record."""
        import oracle
        from tests.helpers import ranked_parity
        if self.index is None:
            self._build()
        qf_h, qp_h = qf.float().cpu().numpy(), qp.float().cpu().numpy()
        ids, sc = doc_idx.cpu().numpy(), doc_score.cpu().numpy()
        ok, worst, exact, n = True, 0.0, 0, 0
        for q in rows:
            ref = oracle.retrieve_one(self.index, qf_h[q], qp_h[q])
            rep = ranked_parity(ids[q], sc[q], ref.sorted_doc_ids, ref.sorted_doc_scores, ref.x[self.kg.passage_vertex])
            ok = ok and rep["equal"] and rep["worst_rel_err"] < 1e-5
            worst = max(worst, rep["worst_rel_err"])
            exact += rep["exact_positions"]; n += rep["n"]
        return {"against": "fp64 CPU oracle (oracle.retrieve_one), rank 0", "queries": [int(q) for q in rows],
                "topk_ids_equal": bool(ok), "exact_id_fraction": exact / max(n, 1), "max_rel_score_err": worst, "ok": bool(ok)}


def bench_main(args, configs, rank: int, local_rank: int, world: int, roofline_fn=None) -> int:
    torch, dist = _td()
    from . import synth
    from .engine import HippoRAGEngine

    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly --gpus ranks "
                         f"(plain `python bench.py --gpus {args.gpus}` spawns them itself; or "
                         f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < world:
        raise SystemExit(f"bench.py --gpus {world} needs {world} GPUs on this node (one rank per device); "
                         f"{n_dev} visible to rank {rank}")
    cfg = configs[args.config]
    strong = "global_batch" in cfg                  # configs[3]: the global batch is fixed, the per-GPU batch shrinks
    if strong and not args.batch and cfg["global_batch"] % world:
        raise SystemExit(f"--config {args.config}: the global batch {cfg['global_batch']} is not a multiple of --gpus {world}")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world == 1:      # HRAG_FORCE_DIST=1 on one GPU without a launcher: a rendezvous with ourselves
        from .launch import free_port
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
    # RCCL prints its version banner on the C-level stdout when the first communicator comes up: send that to stderr so
    # that the ONE JSON line is the only thing rank 0 ever writes to stdout (an external launcher does not filter)
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist.barrier()
        torch.cuda.synchronize()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    finally:
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    V, E, D, seed = cfg["V"], cfg["E"], cfg["D"], cfg["seed"]
    # per-GPU batch: fixed (weak scaling), or the fixed global batch dealt to the GPUs (strong scaling, configs[3])
    B = args.batch or (cfg["global_batch"] // world if strong else cfg["B"])
    K_F, K_P, ITERS, DAMP, PW = 5, 200, 20, 0.5, 0.05

    kg = synth.make_kg(V, E, seed, power_law=bool(cfg.get("power_law")))   # same seed on every rank => identical index
    emb_dtype = torch.float16 if cfg.get("fp16") else torch.bfloat16
    pass_emb = synth.make_embeddings_torch(kg.n_passages, D, seed + 1, dev, dtype=emb_dtype)
    fact_emb = synth.make_embeddings_torch(kg.n_facts, D, seed + 2, dev, dtype=emb_dtype)
    n_batches = args.steps + args.warmup
    cnt = torch.full((B,), K_F, dtype=torch.int32, device=dev)

    def barrier_sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds: float) -> float:
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---------------- replica mode: every rank serves its own B queries ----------------------
    eng = HippoRAGEngine(kg.csr, kg.passage_vertex, pass_emb, fact_emb, kg.subj_vertex, kg.obj_vertex,
                         kg.num_chunks, max_batch=B, max_topk=K_P, slab_width=args.slab_width,
                         locality=getattr(args, "locality", None))
    qf = [synth.make_queries_torch(fact_emb, B, seed + 100 + i + 1000 * rank)[0] for i in range(n_batches)]
    qp = [synth.make_queries_torch(pass_emb, B, seed + 500 + i + 1000 * rank)[0] for i in range(n_batches)]

    def step(i):
        idx, sc = eng.score_facts(qf[i], k=K_F)
        return eng.retrieve(qp[i], idx, sc, cnt, link_top_k=K_F, damping=DAMP, passage_node_weight=PW,
                            ppr_iters=ITERS, k=K_P)

    for i in range(args.warmup):
        step(i)
    barrier_sync()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_batches):
        step(i)
    barrier_sync()
    replica_s = max_over_ranks(time.perf_counter() - t0)
    replica_qps = world * B * args.steps / replica_s
    # phase breakdown + the dominant kernel's roofline, measured on this rank's engine (every rank runs
    # it so that the ranks stay in step; rank 0 reports)
    roofline = phases = None
    if roofline_fn is not None:
        eng.set_profiling(True)
        step(n_batches - 1)
        torch.cuda.synchronize()
        phases = eng.timings()
        eng.set_profiling(False)
        roofline, _, _ = roofline_fn(eng, kg, V, B, phases, args.config, getattr(args, "sweep_launches", 40))
        barrier_sync()
    result = {
        "metric": "retrieval_queries_per_sec", "value": replica_qps, "unit": "queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": replica_s * 1e3 / max(args.steps, 1), "higher_is_better": True,
        "scaling": "strong" if strong and not args.batch else "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "value_leg": "replica",
        "config": {"workload": cfg["label"], "V": V, "E": E, "nnz": kg.csr.nnz,
                   "n_passages": kg.n_passages, "n_facts": kg.n_facts, "dim": D,
                   "global_batch": world * B, "per_gpu_batch": B, "ppr_iters": ITERS,
                   "linking_top_k": K_F, "retrieval_top_k": K_P,
                   "parallelism": f"replica x{world} (queries sharded, no data-path collective)"},
        "roofline": roofline,
        "phases_ms": ({k: phases[k] for k in ("fact_sim_ms", "pass_sim_ms", "seed_ms", "ppr_ms", "rank_ms", "total_ms")}
                      if phases else None),
        "replica": {"value": replica_qps, "unit": "queries/s", "ms_per_step": replica_s * 1e3 / max(args.steps, 1),
                    "parallelism": f"replica x{world}: every GPU holds the whole index and serves its own {B} queries"},
        "rowshard": None, "hybrid": None,
        "value_rowshard": None, "value_hybrid": None, "value_replica": replica_qps,
        "multi_gpu_note": "`value` = the ROW-SHARDED leg when it is parity-green (`value_leg` names the leg): CSR rows + "
                          "embeddings sharded over the GPUs, one collective on the e4m3 PPR iterate per sweep -- the layout "
                          "BASELINE.json's north star names (SURVEY.md 8(e): the primary figure).  `value_hybrid` "
                          "(embeddings sharded, PPR query-parallel, no per-sweep collective: the faster design) and "
                          "`value_replica` (nothing sharded) are always printed beside it; a weak-scaling run on "
                          "configs[2] also carries `configs3_strong` = BASELINE configs[3]'s global batch of 1024 on the "
                          "same index.  No multi-GPU box was available to the rounds that wrote this code: every "
                          "N > 1 figure is the driver's to take",
    }

    # The sharded legs must never cost the line: a watchdog prints what has been measured (rank 0) and ends the
    # process if a leg or the teardown stalls.
    import threading
    printed = threading.Event()
    hybrid_box, strong_box = {}, {}

    def emit(rowshard):
        if rank == 0 and not printed.is_set():
            printed.set()
            result["rowshard"] = rowshard
            result["hybrid"] = hybrid_box.get("res")
            for name in ("rowshard", "hybrid"):
                leg = result[name]
                result["value_" + name] = leg.get("value") if isinstance(leg, dict) else None
            if strong_box:
                result["configs3_strong"] = strong_box
            pick = pick_value_leg(getattr(args, "mode", "auto"), hybrid_box.get("res"), rowshard)
            if pick != "replica":
                # primary number = a leg that shards the CORPUS; the replica figure stays beside it
                leg = result[pick]
                result["value"], result["ms_per_step"] = leg["value"], leg["ms_per_step"]
                result["config"]["parallelism"] = leg["parallelism"]
            result["value_leg"] = pick
            try:   # RCCL's version banner sits in the C stdio buffer: emit it first so that the JSON is the last line
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            print(json.dumps(result), flush=True)

    def watchdog(limit_s, why):
        done = threading.Event()

        def run():
            if not done.wait(limit_s):
                emit({"error": f"{why} exceeded {limit_s:.0f} s; leg abandoned"})
                os._exit(0)
        threading.Thread(target=run, daemon=True).start()
        return done

    limit = float(getattr(args, "rowshard_timeout_s", 240.0))
    rowshard = None
    if getattr(args, "no_rowshard", False) or limit <= 0:
        rowshard = {"skipped": True}
    else:
        leg_done = watchdog(limit, "hybrid + row-sharded legs")
        sidx = seng = None
        try:
            gb = world * B
            sidx = shard_index(kg.csr, kg.passage_vertex, world, kg.subj_vertex, kg.obj_vertex, kg.num_chunks)
            seng = build_shard_engine(sidx, pass_emb, fact_emb, rank, gb, K_P)
        except Exception as exc:
            rowshard = {"error": f"shard engine: {type(exc).__name__}: {exc}"}
        if seng is not None:
            probe = OracleProbe(kg, fact_emb, pass_emb) if rank == 0 and not getattr(args, "no_cpu_baseline", False) else None
            ctx = dict(args=args, kg=kg, sidx=sidx, seng=seng, pass_emb=pass_emb, fact_emb=fact_emb, rank=rank, world=world,
                       K_F=K_F, K_P=K_P, ITERS=ITERS, DAMP=DAMP, PW=PW, seed=seed, dev=dev, barrier_sync=barrier_sync,
                       max_over_ranks=max_over_ranks, replica_eng=eng, probe=probe)
            try:    # embeddings row-sharded, PPR query-parallel (no exchange in the PPR)
                hybrid_box["res"] = _hybrid_leg(B=B, **ctx)
            except Exception as exc:
                hybrid_box["res"] = {"error": f"{type(exc).__name__}: {exc}"}
            try:
                rowshard = _rowshard_leg(B=B, **ctx)
            except Exception as exc:  # the replica measurement above stays valid; report instead of dying
                rowshard = {"error": f"{type(exc).__name__}: {exc}"}
            # BASELINE configs[3] in the same line: a weak-scaling run on configs[2] (per-GPU batch 256) also measures the
            # STRONG figure -- the same index, global batch 1024 whatever N is -- so that one driver command yields both
            gb3 = int(os.environ.get("HRAG_STRONG_GLOBAL_BATCH", "1024"))     # env: exercise the code on one GPU (world 1)
            if (not strong and not args.batch and (world > 1 or "HRAG_STRONG_GLOBAL_BATCH" in os.environ)
                    and args.config == "cfg3" and gb3 % world == 0 and gb3 // world <= B
                    and not getattr(args, "no_strong", False)):
                b3 = gb3 // world
                strong_box.update({"workload": "configs[3]: the same 1M-node/10M-edge index sharded across the GPUs of one node, "
                                               f"GLOBAL batch {gb3} ({b3} per GPU): strong scaling", "global_batch": gb3,
                                   "per_gpu_batch": b3})
                for name, fn in (("rowshard", _rowshard_leg), ("hybrid", _hybrid_leg)):
                    try:
                        strong_box[name] = fn(B=b3, **dict(ctx, probe=None, seed=seed + 50000))
                    except Exception as exc:
                        strong_box[name] = {"error": f"{type(exc).__name__}: {exc}"}
                    strong_box["value_" + name] = strong_box[name].get("value")
                try:
                    sq = [synth.make_queries_torch(fact_emb, b3, seed + 61000 + i + 1000 * rank)[0] for i in range(2 + args.steps)]
                    sp = [synth.make_queries_torch(pass_emb, b3, seed + 62000 + i + 1000 * rank)[0] for i in range(2 + args.steps)]

                    def step3(i):
                        i3, s3 = eng.score_facts(sq[i], k=K_F)
                        return eng.retrieve(sp[i], i3, s3, cnt[:b3], link_top_k=K_F, damping=DAMP, passage_node_weight=PW,
                                            ppr_iters=ITERS, k=K_P)
                    step3(0); step3(1)
                    barrier_sync()
                    t3 = time.perf_counter()
                    for i in range(2, 2 + args.steps):
                        step3(i)
                    barrier_sync()
                    s3 = max_over_ranks(time.perf_counter() - t3)
                    strong_box["value_replica"] = gb3 * args.steps / s3
                except Exception as exc:
                    strong_box["value_replica"] = None
                    strong_box["replica_error"] = f"{type(exc).__name__}: {exc}"
                pick3 = pick_value_leg(getattr(args, "mode", "auto"), strong_box.get("hybrid"), strong_box.get("rowshard"))
                strong_box["value_leg"] = pick3
                strong_box["value"] = strong_box.get("value_" + pick3)
            seng.close()
        leg_done.set()
    eng.close()
    emit(rowshard)
    teardown_done = watchdog(30.0, "process-group teardown")
    dist.barrier()
    dist.destroy_process_group()
    teardown_done.set()
    return 0


def _hybrid_leg(*, args, kg, sidx, seng, pass_emb, fact_emb, rank, world, B, K_F, K_P, ITERS, DAMP, PW, seed, dev,
                barrier_sync, max_over_ranks, replica_eng, probe=None):
    """The global batch (world * B) in the hybrid mode: every rank scores all queries against ITS embedding rows, one
    all-to-all hands it the passage-score rows of its B queries, the PPR runs on the rank's own (replicated-graph)
    engine without any exchange.  Checked bit for bit against the single-GPU engine on the rank's queries (every rank)
    and, with `probe`, against the fp64 oracle on 4 of rank 0's queries."""
    torch, dist = _td()
    from . import synth
    from .engine import ShardStages
    gb = world * B
    hy = HybridRetriever(ShardStages(seng), replica_eng, TorchComm(rank, world), sidx.passages)
    steps, warm = max(1, args.steps), max(1, min(args.warmup, 2))
    n = steps + warm
    gqf = [synth.make_queries_torch(fact_emb, gb, seed + 7000 + i)[0] for i in range(n)]
    gqp = [synth.make_queries_torch(pass_emb, gb, seed + 7500 + i)[0] for i in range(n)]
    gcnt = torch.full((gb,), K_F, dtype=torch.int32, device=dev)
    kw = dict(link_top_k=K_F, damping=DAMP, passage_node_weight=PW, ppr_iters=ITERS, k=K_P)

    def step(i):
        idx, sc = hy.score_facts(gqf[i], k=K_F)
        return idx, sc, hy.retrieve(gqp[i], idx, sc, gcnt, **kw)

    for i in range(warm):
        step(i)
    barrier_sync()
    t0 = time.perf_counter()
    for i in range(warm, n):
        idx, sc, out = step(i)
    barrier_sync()
    sec = max_over_ranks(time.perf_counter() - t0)
    mine = hy.my_rows(gb)
    i1, s1 = replica_eng.score_facts(gqf[n - 1][mine], k=K_F)
    one = replica_eng.retrieve(gqp[n - 1][mine], i1, s1, gcnt[mine], **kw)
    torch.cuda.synchronize()
    same = bool(torch.equal(out.doc_idx, one.doc_idx) and torch.equal(out.doc_score, one.doc_score) and
                torch.equal(idx[mine], i1) and torch.equal(sc[mine], s1))
    ok = torch.tensor([1 if same else 0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    parity = {"against": "single-GPU engine on this rank's queries, every rank", "bit_identical_on_every_rank": bool(ok.item() == 1),
              "ok": bool(ok.item() == 1)}
    if probe is not None:       # rank 0's slice of the last global batch starts at row 0
        rows = sorted({0, B // 3, (2 * B) // 3, B - 1})
        parity["vs_oracle"] = probe.check(gqf[n - 1][mine], gqp[n - 1][mine], out.doc_idx, out.doc_score, rows)
        parity["ok"] = bool(parity["ok"] and parity["vs_oracle"]["ok"])
    barrier_sync()               # the other ranks wait for rank 0's oracle queries here, not inside a later collective
    np_total = len(sidx.passage_vertex)
    return {"value": gb * steps / sec, "unit": "queries/s", "global_batch": gb, "steps": steps,
            "ms_per_step": sec * 1e3 / steps,
            "parallelism": f"hybrid x{world}: fact / passage embeddings row-sharded, one all-to-all of passage-score rows, "
                           f"PPR query-parallel on a replicated graph (no exchange)",
            "wire_bytes_per_global_batch_total": int((world - 1) / world * gb * np_total * 4 + world * (world - 1) * gb * K_F * 8),
            "wire_bytes_received_per_gpu_per_global_batch": int((world - 1) / world * B * np_total * 4 + (world - 1) * gb * K_F * 8),
            "parity": parity}


def _rowshard_leg(*, args, kg, sidx, seng, pass_emb, fact_emb, rank, world, B, K_F, K_P, ITERS, DAMP, PW, seed, dev,
                  barrier_sync, max_over_ranks, replica_eng, probe=None):
    """The global batch (world * B) over the row-sharded corpus: fp8-state shards, one collective per exchange group and
    sweep (--collective allgather | allreduce); checked against the single-GPU engine on the same queries and, with
    `probe`, against the fp64 oracle on 4 queries of the global batch (rank 0)."""
    torch, dist = _td()
    from . import synth
    from .engine import ShardStages
    gb = world * B
    groups = int(getattr(args, "exchange_groups", 2))
    collective = getattr(args, "collective", "allgather")
    rs = ShardedRetriever(ShardStages(seng), TorchComm(rank, world, collective=collective), groups=groups)
    rs_steps, rs_warm = max(1, args.steps), max(1, min(args.warmup, 2))
    n = rs_steps + rs_warm
    gqf = [synth.make_queries_torch(fact_emb, gb, seed + 9000 + i)[0] for i in range(n)]
    gqp = [synth.make_queries_torch(pass_emb, gb, seed + 9500 + i)[0] for i in range(n)]
    gcnt = torch.full((gb,), K_F, dtype=torch.int32, device=dev)

    def rs_step(i):
        idx, sc = rs.score_facts(gqf[i], k=K_F)
        return rs.retrieve(gqp[i], idx, sc, gcnt, link_top_k=K_F, damping=DAMP, passage_node_weight=PW,
                           ppr_iters=ITERS, k=K_P, check_saturation=False)      # flags are checked after the timed loop

    for i in range(rs_warm):
        rs_step(i)
    barrier_sync()
    t0 = time.perf_counter()
    for i in range(rs_warm, n):
        out = rs_step(i)
    barrier_sync()
    rs_s = max_over_ranks(time.perf_counter() - t0)
    # parity: the last global batch's first B queries through the single-GPU engine of this rank
    qf, qp = gqf[n - 1][:B], gqp[n - 1][:B]
    idx1, sc1 = replica_eng.score_facts(qf, k=K_F)
    one = replica_eng.retrieve(qp, idx1, sc1, gcnt[:B], link_top_k=K_F, damping=DAMP, passage_node_weight=PW,
                               ppr_iters=ITERS, k=K_P)
    torch.cuda.synchronize()
    ids_s, sc_s = out[0][:B].cpu().numpy(), out[1][:B].cpu().numpy()
    ids_1, sc_1 = one.doc_idx.cpu().numpy(), one.doc_score.cpu().numpy()
    same_ids = float((ids_s == ids_1).mean())
    rel = np.abs(sc_s - sc_1) / np.maximum(np.abs(sc_1), 1e-30)
    flags_any = int(out[2].max().item())
    parity = {"against": "single-GPU engine, same queries (first per-GPU batch of the last global batch)",
              "queries": int(B), "fraction_of_ranked_ids_equal": same_ids, "max_rel_score_diff": float(rel.max()),
              "flags_or": flags_any, "ok": bool(rel.max() < 1e-5 and same_ids > 0.999 and not (flags_any & 8))}
    if probe is not None:       # queries spread over the WHOLE global batch (every rank's rows of the merged result)
        rows = sorted({0, gb // 3, (2 * gb) // 3, gb - 1})
        parity["vs_oracle"] = probe.check(gqf[n - 1], gqp[n - 1], out[0], out[1], rows)
        parity["ok"] = bool(parity["ok"] and parity["vs_oracle"]["ok"])
    barrier_sync()
    lay = seng.shard_layout(gb, groups)
    wire = (world - 1) / world * sidx.num_vertices * 128 * lay.n_slabs     # e4m3 bytes each GPU receives per sweep
    if collective == "allreduce":
        wire *= 2                                                          # ring all-reduce: reduce-scatter + all-gather
    nnz_own = int(sidx.csr.row_ptr[(rank + 1) * sidx.rows_per_shard] - sidx.csr.row_ptr[rank * sidx.rows_per_shard])
    return {"value": gb * rs_steps / rs_s, "unit": "queries/s", "global_batch": gb, "steps": rs_steps,
            "ms_per_step": rs_s * 1e3 / rs_steps,
            "parallelism": f"rowshard x{world}: CSR rows + passage / fact embeddings sharded, e4m3 PPR iterate "
                           f"replicated, one {collective} per exchange group and sweep",
            "exchange": ("in-place all_gather_into_tensor of the owners' row blocks (RCCL)" if collective == "allgather" else
                         "all-reduce SUM over the group region with the foreign blocks zeroed (the north star's literal form)")
                        + f", {lay.n_groups} exchange group(s) pipelined against the sweeps of the other group(s)",
            "collective": collective,
            "wire_bytes_received_per_gpu_per_sweep": wire, "state_bytes_per_buffer": int(lay.state_bytes),
            "n_slabs": int(lay.n_slabs), "exchange_groups": int(lay.n_groups),
            "wire_bytes_received_per_gpu_per_global_batch": wire * ITERS,
            "rows_per_shard": int(sidx.rows_per_shard), "nnz_this_shard": nnz_own, "parity": parity}
