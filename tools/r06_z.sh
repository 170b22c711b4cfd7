#!/bin/bash
# round 6, last call: the whole GPU suite, smoke(), and the driver's command at HEAD
set -o pipefail
O=gpurun_out/r06_z; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -14 | tee $O/pytest_gpu.log
grep -q " passed" $O/pytest_gpu.log || exit 1
grep -q "failed" $O/pytest_gpu.log && exit 1
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
echo "== driver command"
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err || { tail -20 $O/driver_cmd.err; exit 1; }
python3 - $O/driver_cmd.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"])
print("device", d["device_ms_per_step"])
print("families", {k:v["frac"] for k,v in d["roofline_by_family"].items()})
print("c3shard", d["secondary_c3shard"]["ms_per_step"], d["secondary_c3shard"]["device_ms_per_step"])
print("c2", d["secondary_c2"]["ms_per_step"], "cli", d["full_cli_c2"].get("wall_s"), "cpu", d["cpu_baseline"]["value"])
PY
