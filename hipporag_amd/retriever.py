"""Host-side mirror of the reference's retrieval surface, backed by the MI355X engine.

``HippoRAG`` here keeps the names, argument meaning, return types and error behaviour of the
reference class for THIS path (reference src/hipporag/HippoRAG.py):

    retrieve(queries, num_to_retrieve=None, gold_docs=None)      :413-499
    retrieve_dpr(queries, num_to_retrieve=None, gold_docs=None)  :665-732
    rag_qa(queries, gold_docs=None, gold_answers=None)           :591-663   (QA LLM injected)
    get_fact_scores / dense_passage_retrieval / rerank_facts / run_ppr / get_query_embeddings /
    prepare_retrieval_objects / ready_to_retrieve                :1287-1749 (per-method seams)

What is NOT here (out of scope, stays with the reference): OpenIE by LLM, embedding models,
vector stores, prompts, evaluation.  The index is therefore built from what those produce:
documents + their extracted triples + an embedding callable (``index_from_openie``), or straight
from arrays (``from_arrays``).  The graph rules are the reference's (add_fact_edges :867-913,
add_passage_edges :915-957, add_new_nodes :1159-1187, add_new_edges :1189-1223); synonymy edges
(:959-1020, needs the index-time KNN) are accepted as an explicit edge list.

Differences to the reference, all deliberate:
  * all queries of a call are batched through the device instead of the per-query loop (:459);
  * rankings use the documented tie rule (score desc, index desc) where numpy leaves ties open;
  * facts are kept in first-occurrence order (the reference uses list(set(...)), i.e. hash order).
"""

from __future__ import annotations

import contextlib
import gc
import logging
import re
import time
from dataclasses import dataclass, field
from hashlib import md5
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .graph import CSRGraph, build_csr, float_to_bf16_bits

logger = logging.getLogger("hipporag_amd")


# ------------------------------------------------------------------ utils/misc_utils.py mirrors
@dataclass(frozen=True)
class Chunk:                                            # misc_utils.py:35-40
    content: str
    source_id: Optional[str] = None
    metadata: Dict[str, Any] = field(default_factory=dict)


@dataclass
class RetrievalResult:                                  # misc_utils.py:43-50
    query: str
    docs: List[str]
    scores: np.ndarray
    doc_metadata: List[Dict[str, Any]] = field(default_factory=list)
    graph_seeds: List[Tuple] = field(default_factory=list)


@dataclass
class QuerySolution:                                    # misc_utils.py:52-78
    question: str
    docs: List[str]
    doc_scores: np.ndarray = None
    answer: str = None
    gold_answers: List[str] = None
    gold_docs: Optional[List[str]] = None
    thoughts: Optional[List[str]] = None
    doc_metadata: Optional[List[Dict[str, Any]]] = None
    graph_seeds: Optional[List[Tuple]] = None

    def to_dict(self):
        return {
            "question": self.question, "answer": self.answer, "gold_answers": self.gold_answers,
            "docs": self.docs[:5],
            "doc_scores": [round(v, 4) for v in self.doc_scores.tolist()[:5]] if self.doc_scores is not None else None,
            "gold_docs": self.gold_docs,
            "doc_metadata": self.doc_metadata[:5] if self.doc_metadata is not None else None,
            "graph_seeds": self.graph_seeds,
            **({"thoughts": self.thoughts} if self.thoughts is not None else {}),
        }


def compute_mdhash_id(content: str, prefix: str = "") -> str:      # misc_utils.py:141-152
    return prefix + md5(content.encode()).hexdigest()


def text_processing(text):                                          # misc_utils.py:80-85
    if isinstance(text, (list, tuple)):
        return [text_processing(t) for t in text]
    if not isinstance(text, str):
        text = str(text)
    return re.sub("[^A-Za-z0-9 ]", " ", text.lower()).strip()


def filter_invalid_triples(triples):                                # llm_utils.py:222-252
    """What index() applies to every chunk's OpenIE triples before anything else (reformat_openie_results,
    misc_utils.py:87-108, called at HippoRAG.py:305): triples without exactly three elements are dropped, EXACT duplicates
    (raw strings, before text_processing) collapse, order is kept.  Two triples that differ in case only stay two -- and
    count twice in the graph once text_processing has lower-cased them, as in the reference."""
    seen, out = set(), []
    for t in triples:
        if len(t) != 3:
            continue
        v = tuple(str(x) for x in t)
        if v not in seen:
            seen.add(v)
            out.append(v)
    return out


def min_max_normalize(x):                                           # misc_utils.py:130-139
    mn, mx = np.min(x), np.max(x)
    rng = mx - mn
    if rng == 0:
        return np.ones_like(x)
    return (x - mn) / rng


# prompts/linking.py:1-10 -- the instruction sentences the reference hands to its (instruction-tuned)
# embedding models for the two query encodings of this path; stores embedded by the reference can only be
# searched with queries encoded under the same sentences
_QUERY_INSTRUCTIONS = {
    "query_to_fact": "Given a question, retrieve relevant triplet facts that matches this question.",
    "query_to_passage": "Given a question, retrieve relevant documents that best answer the question.",
}


def get_query_instruction(linking_method: str) -> str:
    return _QUERY_INSTRUCTIONS.get(linking_method, _QUERY_INSTRUCTIONS["query_to_passage"])


def sweeps_for_damping(damping: float, tol: float = 1e-6, lo: int = 16, hi: int = 400) -> int:
    """Number of power-iteration sweeps whose truncation error bound damping^K is <= tol (PRPACK iterates
    to 1e-10; the parity bar of this path is 1e-5 relative): 20 at the reference's 0.5, 86 at 0.85."""
    if not (0.0 < damping < 1.0):
        return lo
    return int(min(hi, max(lo, np.ceil(np.log(tol) / np.log(damping)))))


@dataclass
class RetrievalConfig:
    """The BaseConfig fields this path reads (utils/config_utils.py), same names and defaults."""
    linking_top_k: int = 5               # :184
    retrieval_top_k: int = 200           # :188
    damping: float = 0.5                 # :192
    passage_node_weight: float = 0.05    # :91
    qa_top_k: int = 5                    # :203
    embedding_return_as_normalized: bool = True   # :144
    is_directed_graph: bool = False      # :176
    # engine-only knobs (no reference analogue)
    embedding_precision: str = "f32"     # "f32": scores like the reference's fp32 np.dot (HippoRAG.py:1459,1496) -- the
                                         # stores' fp32 vectors as hi + lo bf16, three MFMA products per element
                                         # (HRAG_F32_SPLIT, 3x the embedding stream); "bf16": vectors rounded to bf16 (1x
                                         # stream; near-tied facts / passages may swap against the reference)
    ppr_iters: Optional[int] = None      # None: derived from damping (sweeps_for_damping: 20 at 0.5)
    locality: Optional[str] = "auto"     # the engine renumbers the vertices by the first passage that links them (the
                                         # reference's entity ids are in hash order) and, when that numbering has
                                         # locality, sweeps in SELL-C-sigma windows with an XCD-blocked launch
                                         # (DESIGN.md 4.1); None: the caller's numbering as it is
    # convergence contract (include/hrag.h, hrag_retrieve; the reference's PRPACK iterates to 1e-10,
    # HippoRAG.py:1736-1743): the engine measures the relative update of the passage scores and keeps sweeping
    # while that measure is above ppr_tol.  The measure can UNDER-read the true error: the worst ratio measured
    # (tools/probe_convergence.py, ring graph) is residual / error = 0.29, so a query that passes at ppr_tol may carry
    # ppr_tol / 0.29.  1.5e-6 puts that worst case at 5.2e-6 = the 1e-5 relative parity bar with a factor 1.9 to
    # spare (round 3 shipped 3e-6: 1.03e-5 in that worst case, no margin); 0 = exactly ppr_iters sweeps
    ppr_tol: float = 1.5e-6
    ppr_max_iters: int = 400             # bound on the sweeps a slowly mixing graph may cost (fp8 state: 30, then
                                         # the flagged queries are repeated on the wider state)
    ppr_base_iters_narrow: Optional[int] = 12     # batches of at most 64 queries (interactive calls, IRCoT steps) under the
                                         # contract (ppr_tol > 0): this many sweeps ALWAYS run (>= 11) instead of the
                                         # worst-case count above, and the measured residual adds stages of 1, 2, 3, 3
                                         # sweeps where the graph needs them -- PRPACK likewise stops when ITS residual
                                         # says so (HippoRAG.py:1736-1743).  Same error bound (include/hrag.h; asserted per
                                         # query on the adversarial graphs: tests/test_gpu_fp8_adversarial.py); 15 sweeps
                                         # instead of 20 on the benchmark graph: -20 ... -24 % per call
                                         # (profiles/r06d_sweep_smallb_cfg3.json).  None / 0: the count above always runs
    ppr_accel: bool = True               # HRAG_OPT_ACCEL (include/hrag.h): wide batches (> 64 queries) run Chebyshev steps
                                         # inside the fp8 stages -- 17 sweeps + what the measured residual asks for
                                         # instead of 20 + ...; the answer is tolerance-driven either way, like the
                                         # reference's PRPACK solve (HippoRAG.py:1736-1743).  HippoRAG graphs are
                                         # undirected (config_utils.py:176), which is what the polynomial needs
    max_batch: int = 256
    slab_width: int = 0


def identity_rerank_filter(query, candidate_items, candidate_indices, len_after_rerank=None):
    """Stand-in for DSPyFilter.__call__ (rerank.py:108-131): keeps every candidate."""
    n = len_after_rerank or len(candidate_items)
    return list(candidate_indices)[:n], list(candidate_items)[:n], {"confidence": None}


@contextlib.contextmanager
def gc_paused():
    """Materialising a batch allocates ~50 k small containers (lists of document strings, one metadata dict per
    document, HippoRAG.py:501-507); none of them can be part of a cycle, but they trip the cyclic collector every 700
    allocations and, every so often, a full pass over everything the process holds (the fact and passage tables: tens of
    milliseconds at 1 M facts).  retrieve() pauses the collector while it builds a batch's results: 7 -> 4 ms per
    256 x 200 documents."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class BatchRows(list):
    """What iter_batched_retrieve yields for a batch: the list of (doc ids, doc scores, kept facts) per query, plus the
    batch's arrays themselves (doc_idx int32 [b, k], doc_score fp32 [b, k]) for consumers that work on whole batches."""
    doc_idx = None
    doc_score = None


def batched_retrieve(eng, queries: List[str], q_tensor: Callable, facts: Sequence, rerank_filter: Callable, **kw):
    """The body of retrieve() (HippoRAG.py:459-480) for all queries at once: one (doc ids, doc scores, kept facts) per
    query.  See iter_batched_retrieve, which this flattens."""
    out_rows = []
    for _, rows in iter_batched_retrieve(eng, queries, q_tensor, facts, rerank_filter, **kw):
        out_rows.extend(rows)
    return out_rows


def iter_batched_retrieve(eng, queries: List[str], q_tensor: Callable, facts: Sequence, rerank_filter: Callable, *,
                          linking_top_k: int, damping: float, passage_node_weight: float, ppr_iters: int,
                          num_to_retrieve: int, n_passages: int, timers=None, ppr_tol: float = 0.0,
                          ppr_max_iters: int = 0, ppr_base_iters_narrow: Optional[int] = None):
    """The body of retrieve() (HippoRAG.py:459-480), shared by the mirror class below and by
    reference_adapter.attach(): phase A on the device, the recognition-memory filter on the host (rerank_facts
    :1659-1707), phase B on the device, in batches of eng.max_batch queries.  Yields (offset of the batch, [(doc ids,
    doc scores, kept facts) per query]) batch by batch, in order; raises where the reference's asserts would (:1541,
    :1644).  timers: object whose rerank_time / ppr_time attributes are advanced like the reference's accumulators
    (:184-186).

    The batches are PIPELINED against the host: phase A of batch i + 1 is enqueued before the filter loop of batch i
    runs, and batch i's results are waited for (events on the copies to pinned memory, not a drain of the stream) only
    after batch i + 1's phase B was enqueued -- so the filter loop and whatever the consumer does with a yielded batch
    (the mirror builds its lists of document strings) run while the device works on the next batch.  The device sees
    A(0) A(1) B(0) A(2) B(1) ...; phase B takes everything it needs from its arguments, so the order is free.

    ppr_tol / ppr_max_iters: the convergence contract (RetrievalConfig): queries the engine flags as not converged
    within its sweep budget are repeated -- those queries only -- on the wider state with the sweeps their
    residual asks for, so a slowly mixing graph costs time, never accuracy.
    ppr_base_iters_narrow (RetrievalConfig.ppr_base_iters_narrow): the sweeps that ALWAYS run for a batch of at most 64
    queries under a tolerance -- the measure adds stages where the graph needs them."""
    import torch
    from .engine import host_copy_async, host_wait
    k_f = int(linking_top_k)
    want = max(1, min(int(num_to_retrieve), n_passages))
    k_docs = min(want, eng.max_topk)
    # more documents than the device top-k holds (HippoRAG.py:501-507 slices any prefix of the full ranking): the
    # scores of all passages come back and are ranked here with the library's rule (score desc, larger index first)
    beyond = k_docs < want and hasattr(eng, "last_doc_scores")
    if k_docs < want and not beyond:
        logger.warning("num_to_retrieve=%d exceeds the engine's max_topk=%d: %d documents per query are returned "
                       "(create the engine with a larger retrieval_top_k, <= 2048)", num_to_retrieve, eng.max_topk, k_docs)
    use_facts = len(facts) > 0 and k_f > 0
    two_halves = hasattr(eng, "retrieve_converged_start")
    starts = list(range(0, len(queries), eng.max_batch))

    def tick(name, t0):
        if timers is not None:
            setattr(timers, name, getattr(timers, name, 0.0) + time.time() - t0)

    class Batch:
        pass

    def start_a(lo):                                                                   # phase A, enqueued
        bt = Batch()
        bt.lo, bt.qs = lo, queries[lo: lo + eng.max_batch]
        bt.a = None
        if use_facts:
            t0 = time.time()
            idx, sc = eng.score_facts(q_tensor(bt.qs, "triple"), k=k_f)
            bt.a = (host_copy_async(idx), host_copy_async(sc))
            tick("rerank_time", t0)
        return bt

    def filter_and_start_b(bt):                                                        # host: LLM filter; phase B, enqueued
        qs = bt.qs
        b = len(qs)
        kept_idx = np.full((b, max(k_f, 1)), -1, np.int32)
        kept_sc = np.zeros((b, max(k_f, 1)), np.float32)
        kept_cnt = np.zeros(b, np.int32)
        bt.seeds = [[] for _ in range(b)]
        t0 = time.time()
        if bt.a is not None:
            idx_l, sc_l = host_wait(bt.a[0]).tolist(), host_wait(bt.a[1]).tolist()     # python ints / floats, once
            for i, q in enumerate(qs):
                row, srow = idx_l[i], sc_l[i]
                cand = [j for j in row if j >= 0]
                try:
                    kidx, kfacts, _ = rerank_filter(q, [facts[j] for j in cand], cand, len_after_rerank=k_f)
                except Exception as exc:                                               # :1705-1707
                    logger.error("Error in rerank_facts: %s", exc)
                    kidx, kfacts = [], []
                score_of = {j: srow[p] for p, j in enumerate(row) if j >= 0}
                kidx = [j for j in map(int, kidx) if j in score_of][:k_f]
                n = len(kidx)
                if n:
                    kept_idx[i, :n] = kidx
                    kept_sc[i, :n] = [score_of[j] for j in kidx]
                kept_cnt[i] = n
                bt.seeds[i] = list(kfacts)
        tick("rerank_time", t0)
        t0 = time.time()
        args = (q_tensor(qs, "passage"), torch.from_numpy(kept_idx), torch.from_numpy(kept_sc), torch.from_numpy(kept_cnt))
        base = ppr_iters
        if ppr_base_iters_narrow and ppr_tol > 0 and len(qs) <= 64 and damping <= 0.5:     # (the two-stage state's short
            # split is an argument about damping^9: csrc/engine.hip two_stage_ok; other damping factors keep the count above)
            base = max(11, min(int(ppr_base_iters_narrow), ppr_iters))     # narrow batch under the contract: latency mode
        kw = dict(link_top_k=k_f, damping=damping, passage_node_weight=passage_node_weight, ppr_iters=base,
                  k=k_docs, ppr_tol=ppr_tol, ppr_max_iters=ppr_max_iters, **({"want_all_scores": True} if beyond else {}))
        if two_halves:
            bt.pending = eng.retrieve_converged_start(*args, **kw)
            o = bt.pending.out
            bt.copies = (host_copy_async(o.doc_idx), host_copy_async(o.doc_score))
        else:                                       # an engine with the one-call form only (test doubles)
            bt.pending, bt.out = None, eng.retrieve_converged(*args, **kw)
        tick("ppr_time", t0)

    def finish(bt):
        t0 = time.time()
        if bt.pending is not None:
            out = bt.pending.finish()
            if bt.pending.repeated:                 # something was run again: take what the tensors hold now
                d_idx, d_sc = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
            else:
                # out of the pinned staging buffers: the rows handed out must not keep page-locked memory alive
                d_idx, d_sc = host_wait(bt.copies[0]).copy(), host_wait(bt.copies[1]).copy()
            flags = bt.pending.flags                # host copy: reading out.flags would drain the stream
        else:
            out = bt.out
            d_idx, d_sc = out.doc_idx.cpu().numpy(), out.doc_score.cpu().numpy()
            flags = out.flags.cpu().numpy()
        if beyond:
            full = out.all_scores.cpu().numpy()
            d_idx = np.stack([np.argsort(r, kind="stable")[::-1][:want] for r in full]).astype(np.int32)
            d_sc = np.take_along_axis(full, d_idx.astype(np.int64), axis=1)
        tick("ppr_time", t0)
        rows = BatchRows()
        rows.doc_idx, rows.doc_score = d_idx, d_sc
        for i in range(len(bt.qs)):
            if flags[i] & 4:      # :1541
                raise AssertionError("count_nonzero(all_phrase_weights) != len(linking_score_map)")
            if flags[i] & 2:      # :1644
                raise AssertionError(f"No phrases found in the graph for the given facts: {bt.seeds[i]}")
            if flags[i] & 1:
                logger.info("No facts found after reranking, return DPR results")      # :468
            rows.append((d_idx[i], d_sc[i], bt.seeds[i]))
        return bt.lo, rows

    if not starts:
        return
    nxt, prev = start_a(starts[0]), None
    for n, lo in enumerate(starts):
        cur = nxt
        if bt_wait := cur.a:                        # batch n's fact candidates are needed on the host now
            t0 = time.time()
            host_wait(bt_wait[0])
            tick("rerank_time", t0)
        nxt = start_a(starts[n + 1]) if n + 1 < len(starts) else None
        filter_and_start_b(cur)
        if prev is not None:
            yield finish(prev)
        prev = cur
    yield finish(prev)


class HippoRAG:
    def __init__(self, global_config: Optional[RetrievalConfig] = None, embedding_model=None,
                 rerank_filter: Optional[Callable] = None, qa_fn: Optional[Callable] = None):
        """embedding_model: object with batch_encode(texts, instruction=..., norm=True) -> [n, D]
        (reference embedding_model/base.py); rerank_filter: callable like DSPyFilter (identity by
        default); qa_fn(query_solutions) -> (solutions, messages, metadata) like HippoRAG.qa (:808)."""
        self.global_config = global_config or RetrievalConfig()
        self.embedding_model = embedding_model
        self.rerank_filter = rerank_filter or identity_rerank_filter
        self.qa_fn = qa_fn
        self.ready_to_retrieve = False
        self.engine = None
        self.ppr_time = self.rerank_time = self.all_retrieval_time = 0.0      # :184-186
        self.query_to_embedding: Dict[str, Dict[str, np.ndarray]] = {"triple": {}, "passage": {}}
        # index state
        self.passage_node_keys: List[str] = []
        self.passage_texts: List[str] = []
        self.chunk_metadata: Dict[str, Dict[str, Any]] = {}
        self.entity_node_keys: List[str] = []
        self.entity_texts: List[str] = []
        self.fact_node_keys: List[str] = []
        self.facts: List[Tuple[str, str, str]] = []
        self.node_name_to_vertex_idx: Dict[str, int] = {}
        self.ent_node_to_chunk_ids: Dict[str, set] = {}
        self.node_to_node_stats: Dict[Tuple[str, str], float] = {}
        self.proc_triples_to_docs: Dict[Tuple[str, str, str], set] = {}
        self._chunk_triples: Dict[str, list] = {}          # chunk key -> processed triples (the OpenIE store)
        self._graph = None                                 # graph.IncrementalGraph (the igraph object)
        self._pass_emb = self._fact_emb = None             # fp32 rows in store order
        self._arrays = None

    # ------------------------------------------------------------------ index side
    def index_from_openie(self, docs: Sequence, chunk_triples: Sequence[Sequence[Sequence[str]]],
                          synonym_edges: Optional[Sequence[Tuple[str, str, float]]] = None,
                          passage_embeddings=None, entity_embeddings=None, fact_embeddings=None,
                          synonymy=None, synonymy_edge_topk: int = 2047, synonymy_edge_sim_threshold: float = 0.8):
        """index() after its LLM steps (HippoRAG.py:262-335): documents + their OpenIE triples in, graph /
        fact / passage arrays out.  Like the reference it is INCREMENTAL: a later call adds the chunks that
        are new (duplicates collapse on their hash), appends the entities / facts / vertices that did not
        exist (stores and igraph keep insertion order), and appends edges for the NEW chunks only
        (add_fact_edges counts a triple only while its chunk is not a graph vertex yet, :899-903;
        add_passage_edges likewise, :939-947) -- parallel to the existing ones, which is how an entity pair
        seen again gains weight (igraph multigraph; graph.build_csr sums).  Embeddings are taken as given
        (rows for the docs of THIS call / for the NEW entities and facts in the order this method appends
        them) or computed with embedding_model.batch_encode(texts) (embedding_store.py:131).

        Synonymy edges (add_synonymy_edges, :959-1020 -- in the reference part of every index() call that adds a chunk):
        `synonym_edges` = an explicit list [(phrase a, phrase b, score)], and / or `synonymy` = "knn": the KNN over ALL
        entities the store holds after this call on the GPU (hipporag_amd.knn.synonymy_candidates, the reference's
        selection rules, `synonymy_edge_topk` / `synonymy_edge_sim_threshold` like config_utils.py), or a callable
        (entity keys, entity texts, fp32 embeddings, topk=, sim_threshold=) -> [(key a, key b, score)] with the same contract.
        Needs entity embeddings (given, or an embedding_model)."""
        from .graph import IncrementalGraph
        chunks = [d if isinstance(d, Chunk) else Chunk(content=d) for d in docs]
        if len(chunks) != len(chunk_triples):
            raise ValueError("one triple list per document")
        if self._graph is None:
            self._graph = IncrementalGraph()
        g = self._graph
        new_chunk_keys, new_chunk_rows = [], []
        for pos, (ch, tr) in enumerate(zip(chunks, chunk_triples)):   # duplicate docs collapse on their hash
            key = compute_mdhash_id(ch.content, "chunk-")
            meta = dict(ch.metadata)
            if ch.source_id is not None:
                meta["source_id"] = ch.source_id
            self.chunk_metadata[key] = meta
            if key not in self._chunk_triples:
                self._chunk_triples[key] = [tuple(text_processing(list(t))) for t in filter_invalid_triples(tr)]
                self.passage_node_keys.append(key)
                self.passage_texts.append(ch.content)
                new_chunk_keys.append(key)
                new_chunk_rows.append(pos)
        proc_new = [self._chunk_triples[k] for k in new_chunk_keys]
        # entity / fact stores: insert_strings appends what is missing (:316-320); sorted unique entities
        # (extract_entity_nodes, misc_utils.py:110-121), facts in first-occurrence order (flatten_facts :123-128)
        known_e = set(self.entity_node_keys)
        new_ents = [e for e in sorted({e for tr in proc_new for t in tr for e in (t[0], t[2])})
                    if compute_mdhash_id(e, "entity-") not in known_e]
        self.entity_texts += new_ents
        self.entity_node_keys += [compute_mdhash_id(e, "entity-") for e in new_ents]

        def embed(given, strings, dim=None):
            if given is not None:
                given = np.asarray(given, dtype=np.float32)
                if given.shape[0] != len(strings):
                    raise ValueError(f"{given.shape[0]} embedding rows for {len(strings)} new strings")
                return given
            if not strings:
                return np.zeros((0, dim or 0), np.float32)
            if self.embedding_model is None:
                raise ValueError("no embeddings given and no embedding_model to compute them")
            return np.asarray(self.embedding_model.batch_encode(list(strings)), dtype=np.float32)

        # the entity store's rows first: the synonymy KNN below reads them
        if entity_embeddings is not None or (self.embedding_model is not None and new_ents):
            ee_new = embed(entity_embeddings, new_ents)
            self.entity_embeddings = ee_new if getattr(self, "entity_embeddings", None) is None else (
                np.concatenate([self.entity_embeddings, ee_new]) if ee_new.size else self.entity_embeddings)
        elif not hasattr(self, "entity_embeddings"):
            self.entity_embeddings = None
        known_f = set(self.facts)
        new_facts = []
        for tr in proc_new:
            for t in tr:
                if t not in known_f:
                    known_f.add(t)
                    new_facts.append(t)
        self.facts += new_facts
        self.fact_node_keys += [compute_mdhash_id(str(f), "fact-") for f in new_facts]
        # graph bookkeeping: add_fact_edges (:867-913) + add_passage_edges (:915-957), new chunks only
        self.node_to_node_stats = {}
        for chunk_key, tr in zip(new_chunk_keys, proc_new):
            in_chunk = set()
            for t in tr:
                a, b = compute_mdhash_id(t[0], "entity-"), compute_mdhash_id(t[2], "entity-")
                in_chunk.update((a, b))
                self.node_to_node_stats[(a, b)] = self.node_to_node_stats.get((a, b), 0.0) + 1
                self.node_to_node_stats[(b, a)] = self.node_to_node_stats.get((b, a), 0.0) + 1
                self.proc_triples_to_docs.setdefault(t, set()).add(chunk_key)          # :1353-1361
            for n in in_chunk:
                self.ent_node_to_chunk_ids.setdefault(n, set()).add(chunk_key)
            for e in sorted({e for t in tr for e in (t[0], t[2])}):
                self.node_to_node_stats[(chunk_key, compute_mdhash_id(e, "entity-"))] = 1.0
        for a, b, w in (synonym_edges or []):              # add_synonymy_edges result (:1007-1018)
            ka = compute_mdhash_id(text_processing(a), "entity-")
            kb = compute_mdhash_id(text_processing(b), "entity-")
            self.node_to_node_stats[(ka, kb)] = float(w)
        if synonymy is not None and new_chunk_keys and len(self.entity_node_keys) > 1:      # :329-335: only with new chunks
            if synonymy == "knn":
                from .knn import synonymy_candidates as synonymy
            elif not callable(synonymy):
                raise ValueError("synonymy must be None, 'knn' or a callable")
            if self.entity_embeddings is None or len(self.entity_embeddings) != len(self.entity_node_keys):
                raise ValueError("synonymy edges need the embeddings of every entity (entity_embeddings= or an embedding_model)")
            for ka, kb, w in synonymy(list(self.entity_node_keys), list(self.entity_texts), self.entity_embeddings,
                                      topk=synonymy_edge_topk, sim_threshold=synonymy_edge_sim_threshold):
                self.node_to_node_stats[(ka, kb)] = float(w)      # overwrites a fact-edge count of the same pair, like :1015
        if new_chunk_keys:                                 # augment_graph only when chunks were added (:329-335)
            # vertices: the entity store's rows, then the chunk store's, that are not vertices yet (:1171-1187)
            g.add_vertices(self.entity_node_keys)
            g.add_vertices(self.passage_node_keys)
            g.add_edges(list(self.node_to_node_stats.keys()), list(self.node_to_node_stats.values()))   # :1200-1223

        if passage_embeddings is not None:                 # rows of THIS call's docs: keep the new chunks' rows
            passage_embeddings = np.asarray(passage_embeddings, np.float32)[new_chunk_rows]
        pe_new = embed(passage_embeddings, [self.passage_texts[self.passage_node_keys.index(k)] for k in new_chunk_keys])
        dim = pe_new.shape[1] if pe_new.size else (self._pass_emb.shape[1] if self._pass_emb is not None else 0)
        fe_new = embed(fact_embeddings, [str(f) for f in new_facts], dim)
        self._pass_emb = pe_new if self._pass_emb is None else np.concatenate([self._pass_emb, pe_new])
        self._fact_emb = fe_new if self._fact_emb is None or not self._fact_emb.size else (
            np.concatenate([self._fact_emb, fe_new]) if fe_new.size else self._fact_emb)
        self._refresh_arrays()
        return self

    def _refresh_arrays(self):
        """The engine-facing arrays from the current graph / stores (what prepare_retrieval_objects reads,
        HippoRAG.py:1287-1389)."""
        g = self._graph
        self.node_name_to_vertex_idx = dict(g.index)
        csr = g.to_csr()
        num_chunks = np.zeros(g.num_vertices, np.int32)
        for k, srcs in self.ent_node_to_chunk_ids.items():
            if k in g.index:
                num_chunks[g.index[k]] = len(srcs)

        def vid(phrase):                                   # HippoRAG.py:1584-1597
            return g.index.get(compute_mdhash_id(phrase.lower(), "entity-"), -1)

        subj = np.array([vid(f[0]) for f in self.facts], np.int32)
        obj = np.array([vid(f[2]) for f in self.facts], np.int32)
        pv = np.array([g.index[k] for k in self.passage_node_keys], np.int32)
        has_facts = len(self.facts) > 0
        self._arrays = dict(csr=csr, passage_vertex=pv, passage_emb=self._emb_for_engine(self._pass_emb),
                            fact_emb=self._emb_for_engine(self._fact_emb) if has_facts else None, subj=subj, obj=obj,
                            num_chunks=num_chunks)
        self.ready_to_retrieve = False                      # explicit invalidate hook

    def _emb_for_engine(self, emb):
        """The stores' fp32 rows as the engine takes them: fp32 (HRAG_F32_SPLIT) or bf16 bit patterns."""
        prec = self.global_config.embedding_precision
        if prec not in ("f32", "bf16"):
            raise ValueError(f"embedding_precision must be 'f32' or 'bf16', not {prec!r}")
        e = np.ascontiguousarray(emb, dtype=np.float32)
        return e if prec == "f32" else float_to_bf16_bits(e)

    # ------------------------------------------------------------------ delete :337-411
    def delete(self, docs_to_delete: Sequence[str]):
        """Remove documents: their chunk vertices, the facts no remaining chunk states, and the entities no
        remaining chunk mentions (stores, embeddings and graph vertices -- igraph renumbers the rest, :408).
        Like the reference, edges between entities that survive are NOT decremented (delete only removes
        vertices), and the surviving entities lose the deleted chunks from ent_node_to_chunk_ids (the
        divisor of the seed weights, :1600-1601)."""
        if self._graph is None:
            return self
        key_of = {t: k for k, t in zip(self.passage_node_keys, self.passage_texts)}
        chunk_ids = {key_of[d] for d in docs_to_delete if d in key_of}                 # :351-356
        if not chunk_ids:
            return self
        affected = {t for k in chunk_ids for t in self._chunk_triples[k]}                # :359-374
        dead_facts = set()
        for t in affected:
            left = self.proc_triples_to_docs.get(t, set()) - chunk_ids
            if left:
                self.proc_triples_to_docs[t] = left
            else:
                self.proc_triples_to_docs.pop(t, None)
                dead_facts.add(t)
        dead_ents = set()
        for e in {x for t in affected for x in (t[0], t[2])}:                            # :377-390
            k = compute_mdhash_id(e, "entity-")
            left = self.ent_node_to_chunk_ids.get(k, set()) - chunk_ids
            if left:
                self.ent_node_to_chunk_ids[k] = left
            else:
                self.ent_node_to_chunk_ids.pop(k, None)
                dead_ents.add(k)
        logger.info("Deleting %d Chunks", len(chunk_ids))
        logger.info("Deleting %d Triples", len(dead_facts))
        logger.info("Deleting %d Entities", len(dead_ents))
        keep_p = [i for i, k in enumerate(self.passage_node_keys) if k not in chunk_ids]   # stores (:398-403)
        keep_f = [i for i, f in enumerate(self.facts) if f not in dead_facts]
        keep_e = [i for i, k in enumerate(self.entity_node_keys) if k not in dead_ents]
        self.passage_node_keys = [self.passage_node_keys[i] for i in keep_p]
        self.passage_texts = [self.passage_texts[i] for i in keep_p]
        self._pass_emb = self._pass_emb[keep_p]
        self.facts = [self.facts[i] for i in keep_f]
        self.fact_node_keys = [self.fact_node_keys[i] for i in keep_f]
        self._fact_emb = self._fact_emb[keep_f] if self._fact_emb is not None and self._fact_emb.size else self._fact_emb
        self.entity_texts = [self.entity_texts[i] for i in keep_e]
        self.entity_node_keys = [self.entity_node_keys[i] for i in keep_e]
        if getattr(self, "entity_embeddings", None) is not None:
            self.entity_embeddings = self.entity_embeddings[keep_e]
        for k in chunk_ids:
            self._chunk_triples.pop(k, None)
            self.chunk_metadata.pop(k, None)
        self._graph.delete_vertices(list(dead_ents) + list(chunk_ids))                  # :408
        self._refresh_arrays()                                                          # :411 ready_to_retrieve = False
        return self

    @classmethod
    def from_arrays(cls, csr: CSRGraph, passage_vertex, passage_emb_bits, fact_emb_bits, subj, obj, num_chunks,
                    passage_texts: Optional[List[str]] = None, facts: Optional[List[Tuple]] = None, **kw):
        self = cls(**kw)
        n_p = len(passage_vertex)
        self.passage_texts = passage_texts or [f"passage {i}" for i in range(n_p)]
        self.passage_node_keys = [compute_mdhash_id(t, "chunk-") for t in self.passage_texts]
        n_f = 0 if fact_emb_bits is None else fact_emb_bits.shape[0]
        self.facts = facts or [(f"s{i}", "rel", f"o{i}") for i in range(n_f)]
        self.fact_node_keys = [compute_mdhash_id(str(f), "fact-") for f in self.facts]
        self._arrays = dict(csr=csr, passage_vertex=np.asarray(passage_vertex, np.int32),
                            passage_emb=passage_emb_bits, fact_emb=fact_emb_bits,
                            subj=np.asarray(subj, np.int32), obj=np.asarray(obj, np.int32),
                            num_chunks=np.asarray(num_chunks, np.int32))
        return self

    # ------------------------------------------------------------------ HippoRAG.py:1287-1389
    def prepare_retrieval_objects(self):
        """Stage the index on the device.  After an incremental index() / delete() the embedding rows the old
        engine already holds are gathered device-side into the new matrices (hrag_engine_gather_embeddings):
        only the rows that are new cross PCIe, the graph (CSR -> SELL-8) is recompiled from the edge list."""
        from . import _lib
        from .engine import HippoRAGEngine
        if self._arrays is None:
            raise RuntimeError("nothing indexed yet")
        a = self._arrays
        self.query_to_embedding = {"triple": {}, "passage": {}}
        has_facts = a["fact_emb"] is not None and a["fact_emb"].shape[0] > 0
        pe, fe = a["passage_emb"], (a["fact_emb"] if has_facts else None)
        old, held = self.engine, getattr(self, "_engine_rows", None)
        same_kind = old is not None and old.f32_split == (a["passage_emb"].dtype == np.float32)
        if old is not None and held is not None and same_kind and old.dim == a["passage_emb"].shape[1]:
            def compose(which, keys, bits, held_keys):
                pos = {k: i for i, k in enumerate(held_keys)}
                src, fresh = [], []
                for i, k in enumerate(keys):
                    if k in pos:
                        src.append(pos[k])
                    else:
                        fresh.append(i)
                        src.append(-len(fresh))
                return old.gather_embeddings(which, np.asarray(src, np.int32), bits[fresh] if fresh else None)
            pe = compose("passages", self.passage_node_keys, a["passage_emb"], held["passages"])
            if has_facts and held["facts"]:
                fe = compose("facts", self.fact_node_keys, a["fact_emb"], held["facts"])
        if old is not None:
            # the gathered matrices no longer depend on the old engine: free its index before the new one is built
            # (peak device memory = one index + the embedding matrices, not two indexes)
            import torch
            torch.cuda.synchronize(old.device)
            old.close()
            self.engine = None
        def build(pe_, fe_):
            return HippoRAGEngine(a["csr"], a["passage_vertex"], pe_, fe_,
                                  a["subj"] if has_facts else None, a["obj"] if has_facts else None,
                                  a["num_chunks"] if has_facts else None,
                                  max_batch=self.global_config.max_batch,
                                  max_topk=min(2048, max(self.global_config.retrieval_top_k, 1)),
                                  slab_width=self.global_config.slab_width,
                                  flags=_lib.OPT_ACCEL if self.global_config.ppr_accel else 0,
                                  locality=self.global_config.locality)
        try:
            engine = build(pe, fe)
        except Exception as exc:
            # the old engine is gone (closed above to keep the peak at ONE index): leave a consistent "not ready"
            # state -- the host arrays still hold the whole index, so the next prepare_retrieval_objects() rebuilds
            # from them -- and say what happened instead of serving from a half-built object
            self.engine, self._engine_rows, self.ready_to_retrieve = None, None, False
            raise RuntimeError("building the device index failed; the previous engine was already released (peak "
                               "memory = one index).  The retriever is NOT ready: free device memory (or lower "
                               "max_batch / use embedding_precision='bf16') and call prepare_retrieval_objects() "
                               f"again -- {type(exc).__name__}: {exc}") from exc
        self.engine = engine
        self._engine_rows = {"passages": list(self.passage_node_keys), "facts": list(self.fact_node_keys) if has_facts else []}
        self.passage_node_idxs = a["passage_vertex"].tolist()
        self.ready_to_retrieve = True

    # ------------------------------------------------------------------ HippoRAG.py:1391-1425
    def get_query_embeddings(self, queries: List):
        strings = []
        for q in queries:
            s = q.question if isinstance(q, QuerySolution) else q
            if s not in self.query_to_embedding["triple"] or s not in self.query_to_embedding["passage"]:
                strings.append(s)
        if strings:
            if self.embedding_model is None:
                raise ValueError("embedding_model is required to encode queries")
            for kind, method in (("triple", "query_to_fact"), ("passage", "query_to_passage")):   # :1414-1423
                embs = self.embedding_model.batch_encode(strings, instruction=get_query_instruction(method),
                                                         norm=True)
                for s, e in zip(strings, embs):
                    self.query_to_embedding[kind][s] = np.asarray(e, dtype=np.float32)

    def _q_tensor(self, queries: List[str], kind: str):
        import torch
        m = np.stack([np.asarray(self.query_to_embedding[kind][q], np.float32).reshape(-1) for q in queries])
        return torch.from_numpy(m).to(self.engine.device).to(self.engine.emb_dtype)

    # ------------------------------------------------------------------ per-method seams (B = 1)
    def get_fact_scores(self, query: str) -> np.ndarray:                  # :1427-1465
        self._ensure_ready()
        if len(self.fact_node_keys) == 0:
            return np.array([])
        self.get_query_embeddings([query])
        s = self.engine.sim_scores("facts", self._q_tensor([query], "triple"))[0].cpu().numpy()
        return min_max_normalize(s)

    def dense_passage_retrieval(self, query: str) -> Tuple[np.ndarray, np.ndarray]:   # :1467-1502
        self._ensure_ready()
        self.get_query_embeddings([query])
        s = self.engine.sim_scores("passages", self._q_tensor([query], "passage"))[0].cpu().numpy()
        s = min_max_normalize(s)
        ids = np.argsort(s, kind="stable")[::-1]
        return ids, s[ids]

    def run_ppr(self, reset_prob: np.ndarray, damping: float = 0.5) -> Tuple[np.ndarray, np.ndarray]:   # :1709-1749
        import torch
        self._ensure_ready()
        if damping is None:
            damping = 0.5
        r = torch.from_numpy(np.asarray(reset_prob, dtype=np.float32).reshape(1, -1))
        x, flags = self.engine.ppr(r, damping=damping, iters=self._ppr_iters(damping))
        if int(flags[0].item()) & 2:
            raise ValueError("reset vector has no positive entry")     # igraph raises here
        doc_scores = x[0].cpu().numpy()[self._arrays["passage_vertex"]]
        ids = np.argsort(doc_scores, kind="stable")[::-1]
        return ids, doc_scores[ids]

    def rerank_facts(self, query: str, query_fact_scores: np.ndarray):     # :1659-1707
        k = self.global_config.linking_top_k
        if len(query_fact_scores) == 0 or len(self.fact_node_keys) == 0:
            return [], [], {"facts_before_rerank": [], "facts_after_rerank": []}
        try:
            cand = np.argsort(query_fact_scores, kind="stable")[::-1][:k].tolist()
            cand_facts = [self.facts[i] for i in cand]
            idx, facts, _ = self.rerank_filter(query, cand_facts, cand, len_after_rerank=k)
            return list(idx), list(facts), {"facts_before_rerank": cand_facts, "facts_after_rerank": list(facts)}
        except Exception as exc:                                           # :1705-1707
            logger.error("Error in rerank_facts: %s", exc)
            return [], [], {"facts_before_rerank": [], "facts_after_rerank": [], "error": str(exc)}

    def _ensure_ready(self):
        if not self.ready_to_retrieve:
            self.prepare_retrieval_objects()

    def _ppr_iters(self, damping: Optional[float] = None) -> int:
        cfg = self.global_config
        return int(cfg.ppr_iters) if cfg.ppr_iters else sweeps_for_damping(cfg.damping if damping is None else damping)

    # ------------------------------------------------------------------ retrieve :413-499
    def retrieve(self, queries: List[str], num_to_retrieve: Optional[int] = None,
                 gold_docs: Optional[List[List[str]]] = None):
        t_start = time.time()
        cfg = self.global_config
        if num_to_retrieve is None:
            num_to_retrieve = cfg.retrieval_top_k
        self._ensure_ready()
        self.get_query_embeddings(queries)
        results: List[QuerySolution] = []
        self._retrieve_into(results, queries, num_to_retrieve)
        self.all_retrieval_time += time.time() - t_start
        logger.info("Total Retrieval Time %.2fs", self.all_retrieval_time)
        logger.info("Total Recognition Memory Time %.2fs", self.rerank_time)
        logger.info("Total PPR Time %.2fs", self.ppr_time)
        if gold_docs is not None:
            return results, self._recall(gold_docs, [r.docs for r in results])
        return results

    def _retrieve_into(self, results, queries, num_to_retrieve):
        cfg = self.global_config
        # batch by batch: the lists of document strings of one batch are built while the device works on the next
        for lo, rows in iter_batched_retrieve(self.engine, queries, self._q_tensor, self.facts if self.fact_node_keys else [],
                                              self.rerank_filter, linking_top_k=cfg.linking_top_k, damping=cfg.damping,
                                              passage_node_weight=cfg.passage_node_weight, ppr_iters=self._ppr_iters(),
                                              ppr_tol=cfg.ppr_tol, ppr_max_iters=cfg.ppr_max_iters,
                                              ppr_base_iters_narrow=cfg.ppr_base_iters_narrow,
                                              num_to_retrieve=num_to_retrieve, n_passages=len(self.passage_node_keys),
                                              timers=self):
            with gc_paused():       # around the materialisation only: the filter (an LLM call in production) runs outside
                results.extend(self._build_query_solutions(queries[lo: lo + len(rows)], rows, num_to_retrieve))

    # ------------------------------------------------------------------ retrieve_ircot :509-558
    def retrieve_ircot(self, queries: List[str], max_qa_steps: int, num_to_retrieve: Optional[int] = None,
                       gold_docs: Optional[List[List[str]]] = None, reason_fn: Optional[Callable] = None):
        """Iterative retrieval: alternate retrieval and one reasoning step per query, merging the document
        scores by max (:526-543).  reason_fn(query, ranked_docs, thoughts) -> str stands for
        utils/qa_utils.reason_step (the IRCoT prompt + reader LLM are outside this package).  The reference
        walks the queries one by one and retrieves with a batch of one per step; queries are independent, so
        here every step retrieves for ALL still-active queries in one batched device call -- same result per
        query, batch kernels instead of 2 * max_qa_steps single-query passes."""
        if max_qa_steps < 1:
            raise ValueError("max_qa_steps must be at least 1.")
        if max_qa_steps > 1 and reason_fn is None:
            raise ValueError("retrieve_ircot with max_qa_steps > 1 needs a reason_fn (the reader LLM is outside this package)")
        if num_to_retrieve is None:
            num_to_retrieve = self.global_config.retrieval_top_k
        first = self.retrieve(list(queries), num_to_retrieve=num_to_retrieve)
        merged = [dict(zip(r.docs, r.doc_scores.tolist())) for r in first]
        meta = [dict(zip(r.docs, r.doc_metadata or [])) for r in first]
        thoughts: List[List[str]] = [[] for _ in queries]
        active = list(range(len(queries)))
        for _ in range(1, max_qa_steps):
            step_q, step_i = [], []
            for i in active:
                ranked = sorted(merged[i], key=merged[i].get, reverse=True)
                thought = reason_fn(queries[i], ranked[:num_to_retrieve], thoughts[i])
                if not isinstance(thought, str):
                    raise TypeError(f"IRCoT reasoning expected a string response, got {type(thought).__name__}.")
                thoughts[i].append(thought)
                if "So the answer is:" in thought:
                    continue
                step_q.append(thought)
                step_i.append(i)
            if not step_q:
                break
            for i, r in zip(step_i, self.retrieve(step_q, num_to_retrieve=num_to_retrieve)):
                for doc, score in zip(r.docs, r.doc_scores.tolist()):
                    merged[i][doc] = max(merged[i].get(doc, float("-inf")), score)
                meta[i].update(dict(zip(r.docs, r.doc_metadata or [])))
            active = step_i
        results = []
        for i, q in enumerate(queries):
            items = sorted(merged[i].items(), key=lambda kv: kv[1], reverse=True)
            results.append(QuerySolution(question=q, docs=[d for d, _ in items],
                                         doc_scores=np.asarray([s for _, s in items]), thoughts=thoughts[i],
                                         doc_metadata=[meta[i].get(d, {}) for d, _ in items]))
        if gold_docs is not None:
            return results, self._recall(gold_docs, [r.docs for r in results])
        return results

    def _result_tables(self):
        """passage texts / keys as numpy object arrays: a batch materialises B x num_to_retrieve document strings
        (HippoRAG.py:501-507 returns them as lists), which one fancy index per query does ~5x faster than a list
        comprehension.  Rebuilt when index() appended or delete() replaced the lists."""
        t = getattr(self, "_tables", None)
        if t is None or t[0] is not self.passage_texts or t[1] != len(self.passage_texts):
            texts = np.empty(len(self.passage_texts), dtype=object)
            texts[:] = self.passage_texts
            keys = np.empty(len(self.passage_node_keys), dtype=object)
            keys[:] = self.passage_node_keys
            t = self._tables = (self.passage_texts, len(self.passage_texts), texts, keys)
        return t[2], t[3]

    def _build_query_solutions(self, queries, rows, num_to_retrieve) -> List[QuerySolution]:
        """_build_retrieval_result + the QuerySolution of retrieve() (:471-480) for one batch.  When every query of the
        batch has a full list of valid ids (the usual case) and there is no chunk metadata, the document strings of the
        whole batch come from ONE fancy index into the text table."""
        ids = None if rows.doc_idx is None else np.asarray(rows.doc_idx)[:, :num_to_retrieve]
        if ids is None or self.chunk_metadata or ids.size == 0 or int(ids.min()) < 0:
            out = []
            for q, (d_idx, d_sc, seeds) in zip(queries, rows):
                r = self._build_retrieval_result(q, d_idx, d_sc, num_to_retrieve, seeds)
                out.append(QuerySolution(question=r.query, docs=r.docs, doc_scores=r.scores,
                                         doc_metadata=r.doc_metadata, graph_seeds=r.graph_seeds))
            return out
        texts, _ = self._result_tables()
        docs = texts[ids].tolist()
        n = ids.shape[1]
        return [QuerySolution(question=q, docs=docs[i], doc_scores=np.asarray(rows[i][1][:n]),
                              doc_metadata=[{} for _ in range(n)], graph_seeds=rows[i][2] or [])
                for i, q in enumerate(queries)]

    def _build_retrieval_result(self, query, sorted_doc_ids, sorted_doc_scores, num_to_retrieve, graph_seeds=None):
        ids = np.asarray(sorted_doc_ids[:num_to_retrieve], dtype=np.int64)                  # :501-507
        ids = ids[ids >= 0]
        texts, keys = self._result_tables()
        meta = self.chunk_metadata
        return RetrievalResult(query=query, docs=texts[ids].tolist(),
                               scores=np.asarray(sorted_doc_scores[:len(ids)]),
                               doc_metadata=([dict(meta.get(k, {})) for k in keys[ids]] if meta
                                             else [{} for _ in range(len(ids))]),
                               graph_seeds=graph_seeds or [])

    # ------------------------------------------------------------------ retrieve_dpr :665-732
    def retrieve_dpr(self, queries: List[str], num_to_retrieve: Optional[int] = None,
                     gold_docs: Optional[List[List[str]]] = None):
        cfg = self.global_config
        if num_to_retrieve is None:
            num_to_retrieve = cfg.retrieval_top_k
        self._ensure_ready()
        self.get_query_embeddings(queries)
        eng = self.engine
        want = max(1, min(num_to_retrieve, len(self.passage_node_keys)))
        k_docs = min(want, eng.max_topk)
        # more documents than the device top-k holds (:704-714 slices any prefix of the full ranking): the raw scores of all
        # passages come back and are normalised and ranked here with the library's rule (score desc, larger index first)
        beyond = k_docs < want
        from .engine import host_copy_async, host_wait
        results = []

        def enqueue(lo):
            qs = queries[lo: lo + eng.max_batch]
            if beyond:
                return qs, host_copy_async(eng.sim_scores("passages", self._q_tensor(qs, "passage"))), None
            idx, sc = eng.dense_retrieve(self._q_tensor(qs, "passage"), k=k_docs)
            return qs, host_copy_async(idx), host_copy_async(sc)

        # one batch ahead of the host, like retrieve(): batch i's lists are built while the device scores batch i + 1
        starts = list(range(0, len(queries), eng.max_batch))
        pending = enqueue(starts[0]) if starts else None
        for n in range(len(starts)):
            qs, idx_h, sc_h = pending
            pending = enqueue(starts[n + 1]) if n + 1 < len(starts) else None
            if beyond:
                raw = host_wait(idx_h)[:, :len(self.passage_node_keys)]
                full = np.stack([min_max_normalize(r) for r in raw]).astype(np.float32)
                idx = np.stack([np.argsort(r, kind="stable")[::-1][:want] for r in full]).astype(np.int32)
                sc = np.take_along_axis(full, idx.astype(np.int64), axis=1)
            else:
                idx, sc = host_wait(idx_h).copy(), host_wait(sc_h).copy()
            with gc_paused():
                for i, q in enumerate(qs):
                    r = self._build_retrieval_result(q, idx[i], sc[i], num_to_retrieve)
                    results.append(QuerySolution(question=r.query, docs=r.docs, doc_scores=r.scores,
                                                 doc_metadata=r.doc_metadata))
        if gold_docs is not None:
            return results, self._recall(gold_docs, [r.docs for r in results])
        return results

    # ------------------------------------------------------------------ rag_qa :591-663
    def rag_qa(self, queries, gold_docs: Optional[List[List[str]]] = None,
               gold_answers: Optional[List[List[str]]] = None):
        if self.qa_fn is None:
            raise NotImplementedError("rag_qa needs a qa_fn (the reader LLM is outside this package)")
        retrieval_metrics = None
        if not isinstance(queries[0], QuerySolution):
            out = self.retrieve(queries=queries, gold_docs=gold_docs)
            queries, retrieval_metrics = out if gold_docs is not None else (out, None)
        solutions, messages, metadata = self.qa_fn(queries)
        if gold_answers is not None:
            for s, g in zip(solutions, gold_answers):
                s.gold_answers = list(g)
        if gold_docs is not None:
            for s, g in zip(solutions, gold_docs):
                s.gold_docs = g
            return solutions, messages, metadata, retrieval_metrics, None
        return solutions, messages, metadata

    @staticmethod
    def _recall(gold_docs, retrieved_docs, k_list=(1, 2, 5, 10, 20, 30, 50, 100, 150, 200)):
        """RetrievalRecall.calculate_metric_scores (evaluation/retrieval_eval.py:16-73), pooled only."""
        out = {}
        for k in k_list:
            vals = []
            for gold, got in zip(gold_docs, retrieved_docs):
                gold_set = set(gold)
                vals.append(len(gold_set & set(got[:k])) / len(gold_set) if gold_set else 0.0)
            out[f"Recall@{k}"] = round(float(np.mean(vals)) if vals else 0.0, 4)
        return out
