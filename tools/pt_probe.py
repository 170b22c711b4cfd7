"""Timing probe (not a test): first full assignment of the projected Lloyd loop at the C3-shard shape, kernel variants."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isle_amd import HotPath
from tools.synth import Corpus
V, D, k = 100_000, 1_250_000, 1000
c = Corpus(V, D, k, 31337); B = c.threshold(k, free_A=True)
hp = HotPath(0); hp.upload_csc(V, B["vals"], B["rows"], B["offs"])
r = hp.compute_block_ks(k, allow_noconv=True)
g = hp.kmeans_init_on_projected_space(k, rng_seed=1)
for label, env, reps in [("hamerly (mode 0), 1 iteration", {"ISLE_PROJ_BOUNDS": "hamerly"}, 1), ("tiles, 1 iteration", {}, 1),
                         ("tiles dbg1 (no stores), 1 iteration", {"ISLE_PT_DBG": "1"}, 1), ("tiles dbg2 (mode-0 epilogue), 1 iteration", {"ISLE_PT_DBG": "2"}, 1),
                         ("tiles, 10 iterations", {}, 10), ("tiles no tighten, 10 iterations", {"ISLE_PROJ_NOTIGHTEN": "1"}, 10),
                         ("hamerly, 10 iterations", {"ISLE_PROJ_BOUNDS": "hamerly"}, 10)]:
    for kk, v in env.items(): os.environ[kk] = v
    for rep in range(2):
        hp.timing_enable(1); hp.timing_reset()
        t = time.perf_counter(); lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"], max_reps=reps); dt = time.perf_counter() - t
        tm = hp.timing_get(); hp.timing_enable(0)
    print("%-44s wall %.1f ms device lloyd_proj %.1f ms iters %d hash %d" % (label, dt * 1e3, tm["lloyd_proj"][0], lp["iters"], int(lp["assign"].astype(np.int64).sum())), flush=True)
    for kk in env: del os.environ[kk]
