#!/bin/bash
# round 6: the driver's N > 1 flow at the driver's own workload, rehearsed on one GPU (two ranks of 5 M documents sharing GPU 0, every collective
# staged through the host): not a measurement — a check of bench.py's control flow, teardown included, at full size
set -o pipefail
O=gpurun_out/r06_o; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
ISLE_BENCH_REHEARSE=1 ISLE_COMM_SELFTEST=1 timeout -k 10 1000 python3 bench.py --gpus 2 --steps 1 --warmup 1 > $O/rehearsal.json 2> $O/rehearsal.err || { tail -30 $O/rehearsal.err; exit 1; }
grep -i "self-test\|launcher" $O/rehearsal.err | head -5
python3 - $O/rehearsal.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["parallelism"]); print("ms_per_step", d["ms_per_step"], "sigma", d["accuracy"]["sigma_rel_err_bound"], "nonempty", d["config"]["kmeans"]["nonempty_clusters"], "purity", d["accuracy"]["planted_topic_agreement"])
print("families", {k:v["frac"] for k,v in d["roofline_by_family"].items()})
PY
