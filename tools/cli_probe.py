"""End-to-end run of the ISLETrain CLI at benchmark size: writes a tdf file + vocabulary, runs isle_amd/host/ISLETrain,
prints wall time and the reference-format timer log.  Usage: python tools/cli_probe.py [V D k] [edge_topics]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.synth import Corpus

V, D, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (50000, 1000000, 200)
edge = int(sys.argv[4]) if len(sys.argv) >= 5 else 0
tmp = tempfile.mkdtemp(prefix="isle_cli_", dir="/tmp")
c = Corpus(V, D, k, 1)
text = c.tdf_bytes()
tdf = os.path.join(tmp, "corpus.tdf")
text.tofile(tdf)
open(os.path.join(tmp, "vocab.txt"), "w").write("\n".join("w%d" % i for i in range(V)))
out = os.path.join(tmp, "out")
os.mkdir(out)
args = [os.path.join(ROOT, "isle_amd", "host", "ISLETrain"), tdf, os.path.join(tmp, "vocab.txt"), out, str(V), str(D), str(c.nnz_A), str(k),
        "0", "0", "0", str(edge), "5000"]
t = time.perf_counter()
r = subprocess.run(args, capture_output=True, text=True)
wall = time.perf_counter() - t
logdir = os.path.join(out, os.listdir(out)[0])
timer = open(os.path.join(logdir, "timerLog.txt")).read()
files = {f: os.path.getsize(os.path.join(logdir, f)) for f in sorted(os.listdir(logdir))}
print(json.dumps({"shape": [V, D, k], "tdf_bytes": int(text.size), "entries": c.nnz_A, "edge_topics": edge, "wall_s": round(wall, 2),
                  "returncode": r.returncode, "stderr_tail": r.stderr[-300:], "output_files_bytes": files,
                  "timer_log": [ln for ln in timer.splitlines() if ln.strip()]}, indent=1))
subprocess.run(["rm", "-rf", tmp])
