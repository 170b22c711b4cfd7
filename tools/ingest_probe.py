"""Full-size check + timing of tdf ingest on the device.  Usage: python tools/ingest_probe.py [V D k]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.synth import Corpus
import isle_amd

V, D, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (50000, 1000000, 200)
c = Corpus(V, D, k, 1)
cnt, rows, offs = c.A()
t = time.perf_counter(); text = c.tdf_bytes(); t_txt = time.perf_counter() - t
hp = isle_amd.HotPath()
hp.ingest_tdf(text[: 1 << 20].tobytes().rsplit(b"\n", 1)[0] + b"\n", V, D)  # warm-up
hp.timing_enable(True)
hp.timing_reset()
t = time.perf_counter(); info = hp.ingest_tdf(text, V, D, max_entries=len(cnt)); t_in = time.perf_counter() - t
dev = hp.timing_get()["ingest"][0]
gc, gr, go = hp.get_A()
same = bool(np.array_equal(go, offs) and np.array_equal(gr, rows) and np.array_equal(gc, cnt))
# CPU comparison: the host parser + std::sort of isle_amd/host (prestage_dump) is timed by tests on small inputs only;
# here: NumPy lexsort of the parsed triples as a floor for the sort alone
doc = np.repeat(np.arange(D, dtype=np.int64), np.diff(offs))
perm = np.random.default_rng(0).permutation(len(doc))
t = time.perf_counter(); np.lexsort((rows[perm], doc[perm])); t_sort = time.perf_counter() - t
print(json.dumps({"shape": [V, D, k], "text_bytes": int(text.size), "lines": int(len(cnt)), "wall_ms_incl_h2d": round(t_in * 1e3, 1),
                  "device_ms": round(dev, 2), "GB_per_s_text": round(text.size / max(dev, 1e-9) / 1e6, 1), "identical_to_generator_csc": same,
                  "numpy_lexsort_only_s": round(t_sort, 2), "text_generation_s": round(t_txt, 2)}))
