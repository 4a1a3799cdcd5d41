"""(Archived with docs/experiments/r05_b0_light_kernel.patch: apply the patch to csrc/ppr8.hip first -- the product
ignores HRAG_P8_B0_LIGHT.)  Round-5 experiment: the first boundary sweep (mode B0) in its light form (csrc/ppr8.hip ppr8_pair_b0_kernel) against the
common pair kernel -- launch time by HIP events and bit-identity of a whole retrieve.  HRAG_P8_B0_LIGHT is read once per
process, so every variant runs in a process of its own:  python tools/exp_b0_light.py (spawns 0 / 4 / 5)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child():
    import torch
    from hipporag_amd import synth
    from hipporag_amd.engine import HippoRAGEngine
    V, E, D, B, seed = 1_000_000, 10_000_000, 64, 256, 1237
    dev = torch.device("cuda", 0)
    kg = synth.make_kg(V, E, seed)
    pemb = synth.make_embeddings_torch(kg.n_passages, D, 1, dev)
    femb = synth.make_embeddings_torch(kg.n_facts, D, 2, dev)
    eng = HippoRAGEngine(kg.csr, kg.passage_vertex, pemb, femb, kg.subj_vertex, kg.obj_vertex, kg.num_chunks, max_batch=B, max_topk=200)
    qf, _ = synth.make_queries_torch(femb, B, 7)
    qp, _ = synth.make_queries_torch(pemb, B, 8)
    cnt = torch.full((B,), 5, dtype=torch.int32, device=dev)
    idx, sc = eng.score_facts(qf, k=5)
    out = eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=200)
    torch.cuda.synchronize()
    h = hashlib.sha256(out.doc_idx.cpu().numpy().tobytes() + out.doc_score.cpu().numpy().tobytes()).hexdigest()[:16]

    def t(mode, n=30):
        eng.ppr_sweeps(B, 4, 0.5, f8=True, f8_mode=mode)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.ppr_sweeps(B, n, 0.5, f8=True, f8_mode=mode); e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    eng.set_profiling(True)
    ms = []
    for i in range(5):
        eng.retrieve(qp, idx, sc, cnt, ppr_iters=20, k=200)
        torch.cuda.synchronize()
        ms.append(eng.timings()["ppr_ms"])
    print(json.dumps({"variant": os.environ.get("HRAG_P8_B0_LIGHT"), "b0_ms": t("B0"), "c_ms": t("C"), "ppr_ms_median": sorted(ms)[2],
                      "result_sha16": h}))


if __name__ == "__main__":
    if os.environ.get("HRAG_B0_CHILD"):
        child()
    else:
        for v in ("0", "4", "5"):
            env = dict(os.environ, HRAG_P8_B0_LIGHT=v, HRAG_B0_CHILD="1")
            p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            print(p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ("FAILED " + p.stderr[-800:]))
