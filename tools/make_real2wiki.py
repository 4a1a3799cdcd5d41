"""Build tests/golden/real2wiki_triples.npz: a REAL-TOPOLOGY knowledge graph from data the reference ships.

Every graph the HIP path had run on came from hipporag_amd.synth (random endpoints) or a toy corpus.  The reference
holds one real corpus, /root/reference/reproduce/dataset/2wikimultihopqa_corpus.json (6 119 Wikipedia abstracts with
titles), but no OpenIE output for it and no LLM is reachable.  This script stands in for the OpenIE step with a
deterministic, LLM-free extractor and commits only INTEGER ids (no corpus text travels into the repository):

  entities  the passage title + every maximal span of capitalised tokens (connectors of / the / de / von / ... allowed
            inside a span) + 4-digit years; sentence-initial function words are dropped; strings go through the
            reference's text_processing (misc_utils.py:80-85: lower-case, non-alphanumerics -> space), so "St. Maurice's
            Abbey" and "St Maurice s Abbey" are one node, as they would be upstream;
  facts     (title, "mentions", e) for every entity e of the passage other than the title -- OpenIE on an encyclopedia
            abstract is subject-centric -- and (e_i, "co-occurs with", e_i+1) for consecutive entities of one sentence;
            per passage, exact duplicates collapse (filter_invalid_triples).

Entity ids = rank of the processed string in sorted order = the order the reference's entity store gives them when the
corpus is indexed in one index() call (extract_entity_nodes sorts, misc_utils.py:110-121; HippoRAG.py:316-320).
What this is NOT: real OpenIE triples.  It IS the topology question the review asked: which passages share which
entities in a real corpus (hubs, communities, long tail), which is what the sweep's gathers see.

    python tools/make_real2wiki.py            # needs /root/reference (authoring container only)
"""
from __future__ import annotations

import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CORPUS = "/root/reference/reproduce/dataset/2wikimultihopqa_corpus.json"
OUT = os.path.join(ROOT, "tests", "golden", "real2wiki_triples.npz")

CONNECT = r"(?:of|the|de|la|le|du|von|van|der|den|di|da|del|al|bin|ibn|y|zu|und|and)"
SPAN = re.compile(rf"\b[A-Z][\w'’\-]*\.?(?:\s+(?:{CONNECT}\s+)*[A-Z][\w'’\-]*\.?)*")
YEAR = re.compile(r"\b(1[0-9]{3}|20[0-2][0-9])\b")
STOP = {"the", "he", "she", "it", "in", "his", "her", "they", "this", "that", "these", "those", "after", "before",
        "during", "at", "on", "as", "a", "an", "its", "their", "there", "when", "while", "from", "with", "by", "for",
        "but", "and", "or", "however", "although", "also", "both", "since", "one", "two", "three", "four", "other",
        "many", "most", "some", "such", "who", "which", "where", "what", "if", "of", "to", "is", "was", "were", "are",
        "following", "according", "later", "early", "between", "among", "under", "over", "despite", "until", "several"}


def sentences(text: str):
    return [s for s in re.split(r"(?<=[.!?])\s+", text) if s.strip()]


def entities_of(sentence: str):
    out = []
    for m in SPAN.finditer(sentence):
        s = m.group(0).strip().rstrip(".")
        toks = s.split()
        # a span that opens the sentence with a function word loses it ("In Paris" -> "Paris")
        while toks and toks[0].lower() in STOP:
            toks = toks[1:]
        if not toks:
            continue
        s = " ".join(toks)
        if len(s) < 2:
            continue
        out.append((m.start(), s))
    out += [(m.start(), m.group(0)) for m in YEAR.finditer(sentence)]
    out.sort()
    return [s for _, s in out]


def main():
    from hipporag_amd.retriever import text_processing
    corpus = json.load(open(CORPUS))
    per_passage = []
    for doc in corpus:
        title = doc["title"]
        triples, seen = [], set()

        def add(s, p, o):
            if s == o or not s or not o:
                return
            t = (s, p, o)
            if t not in seen:
                seen.add(t)
                triples.append(t)

        for sent in sentences(doc["text"]):
            ents = entities_of(sent)
            for e in ents:
                add(title, 0, e)
            for a, b in zip(ents[:-1], ents[1:]):
                add(a, 1, b)
        per_passage.append(triples)
    # processed strings (what becomes a graph node), sorted -> ids
    proc = lambda s: text_processing(s)
    vocab = sorted({proc(x) for tr in per_passage for t in tr for x in (t[0], t[2])} - {""})
    eid = {s: i for i, s in enumerate(vocab)}
    ptr, subj, pred, obj = [0], [], [], []
    for tr in per_passage:
        seen = set()
        for s, p, o in tr:
            a, b = proc(s), proc(o)
            if not a or not b or a == b:
                continue
            key = (eid[a], p, eid[b])
            if key in seen:          # duplicates after text_processing would be two facts upstream only if the RAW strings
                continue             # differed; the fixture keeps one
            seen.add(key)
            subj.append(eid[a]); pred.append(p); obj.append(eid[b])
        ptr.append(len(subj))
    np.savez_compressed(OUT, ptr=np.asarray(ptr, np.int32), subj=np.asarray(subj, np.int32), pred=np.asarray(pred, np.int8),
                        obj=np.asarray(obj, np.int32), n_entities=np.int64(len(vocab)), n_passages=np.int64(len(corpus)))
    deg = np.bincount(np.concatenate([subj, obj]), minlength=len(vocab))
    print(json.dumps({"passages": len(corpus), "entities": len(vocab), "triples": len(subj),
                      "triples_per_passage_mean": len(subj) / len(corpus), "entity_mentions_max": int(deg.max()),
                      "entities_mentioned_once": int((deg == 1).sum()), "file_bytes": os.path.getsize(OUT)}))
    top = np.argsort(-deg)[:12]
    print("most mentioned:", [(vocab[i], int(deg[i])) for i in top])


if __name__ == "__main__":
    main()
