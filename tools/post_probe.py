"""Full-size check + timing of the downstream stage (catchwords, topic model): device vs the CPU restatement.
Usage: python tools/post_probe.py [V D k]   (default: BASELINE config 2 shape)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tools.synth import Corpus
from oracle import oracle as O
import isle_amd

V, D, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (50000, 1000000, 200)
c = Corpus(V, D, k, 1)
cnt, rows, offs = c.A()
hp = isle_amd.HotPath()
hp.upload_counts(V, cnt, rows, offs)
info = hp.threshold(k)
B = hp.get_B()
oc = B["original_cols"].astype(np.int64)
assign = c.planted()[oc].astype(np.uint32)
r, rt = O.catchword_rank(D, k), O.model_rank_threshold(D, k)
hp.find_catchwords(k, r, assign=assign, fetch_thresholds=False)  # warm-up (allocations)
t = time.perf_counter(); got = hp.find_catchwords(k, r, assign=assign); t_cw = time.perf_counter() - t
t = time.perf_counter(); tm = hp.construct_topic_model(k, rt, D); t_tm = time.perf_counter() - t
cl = np.full(D, -1, np.int32); cl[oc] = assign.astype(np.int32)
t = time.perf_counter(); nv = O.post_normalize(offs, cnt, info["avg_doc_sz"]); t_nv = time.perf_counter() - t
t = time.perf_counter(); thr = O.post_catch_thresholds(V, offs, rows, nv, cl, k, r); t_thr = time.perf_counter() - t
t = time.perf_counter(); ct = O.post_find_catchwords(thr); t_ct = time.perf_counter() - t
t = time.perf_counter(); ref = O.post_topic_model(V, offs, rows, nv, cl, ct, k, rt); t_m = time.perf_counter() - t
ok = np.isfinite(ref["model"])
rel = np.abs(tm["model"][ok] - ref["model"][ok]) / np.maximum(np.abs(ref["model"][ok]), 1e-12)
print(json.dumps({
    "shape": [V, D, k], "nnz_A": int(offs[-1]), "r": r, "rank_threshold": rt,
    "device_ms": {"catchwords(incl. D2H of thresholds)": round(t_cw * 1e3, 2), "topic_model(incl. D2H)": round(t_tm * 1e3, 2)},
    "cpu_port_s": {"normalize": round(t_nv, 2), "catch_thresholds": round(t_thr, 2), "find_catchwords": round(t_ct, 2),
                   "topic_model": round(t_m, 2), "threads": O.effective_cpus()},
    "identical": {"thresholds": bool(np.array_equal(got["thresholds"], thr)), "catch_topic": bool(np.array_equal(got["catch_topic"], ct)),
                  "doc_topic_sums": bool(np.array_equal(tm["dts_val"], ref["dts_val"]) and np.array_equal(tm["dts_topic"], ref["dts_topic"])),
                  "model_threshold": bool(np.array_equal(tm["model_threshold"], ref["model_threshold"])),
                  "top_two": bool(np.array_equal(tm["top1"], ref["top1"]) and np.array_equal(tm["top2"], ref["top2"]))},
    "model_max_rel_err(entries > 1e-7)": float(rel[np.abs(ref["model"][ok]) > 1e-7].max()),
    "num_catchwords": int((ct >= 0).sum()), "doc_topic_sums": int(ref["dts_val"].shape[0])}))
