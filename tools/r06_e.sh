#!/bin/bash
# round 6, fifth GPU call: the bench lines of configs 4 and 5 at their own size
set -o pipefail
O=gpurun_out/r06_e; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
echo "== bench --workload c4"
timeout -k 10 500 python3 bench.py --workload c4 --steps 5 --warmup 2 > $O/c4.json 2> $O/c4.err || { tail -20 $O/c4.err; exit 1; }
tail -3 $O/c4.err
echo "== bench --workload c5"
timeout -k 10 650 python3 bench.py --workload c5 --steps 3 --warmup 1 > $O/c5.json 2> $O/c5.err || { tail -20 $O/c5.err; exit 1; }
tail -3 $O/c5.err
python3 - $O/c4.json $O/c5.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"])
    print("  other_stages", json.dumps(d.get("other_stages"))[:1500])
    print("  accuracy gate", d["accuracy"]["gate"], "cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
PY
