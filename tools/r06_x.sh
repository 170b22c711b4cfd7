#!/bin/bash
# round 6: configs 4 and 5 as bench workloads at HEAD (after the orthogonalisation change)
set -o pipefail
O=gpurun_out/r06_x; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for wl in c4 c5; do
  timeout -k 10 560 python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > $O/${wl}_bench.json 2> $O/${wl}_bench.err || { tail -8 $O/${wl}_bench.err; exit 1; }
  python3 - $O/${wl}_bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"][:60], "| value", d["value"], "ms_per_step", d["ms_per_step"], "gate", d["accuracy"]["gate"]["passed"])
print("   device", d["device_ms_per_step"])
print("   other", json.dumps(d.get("other_stages"))[:600])
PY
done
