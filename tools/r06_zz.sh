#!/bin/bash
# round 6: the driver's command once more at the final HEAD (bench.py gained step_traffic and the bf16-peak figure after the last full run)
set -o pipefail
O=gpurun_out/r06_zz; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err || { tail -20 $O/driver_cmd.err; exit 1; }
python3 - $O/driver_cmd.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"])
print("step_traffic", json.dumps(d["step_traffic"])[:400])
print("lloyd", d["roofline_by_family"]["lloyd_proj"]["frac"], d["roofline_by_family"]["lloyd_proj"]["frac_of_bf16_dense_peak"])
print("c3shard", d["secondary_c3shard"]["ms_per_step"], "evd", d["secondary_c3shard"]["device_ms_per_step"]["evd"], "qr", d["secondary_c3shard"]["device_ms_per_step"]["qr"])
print("gate", d["accuracy"]["gate"]["passed"], "cpu", d["cpu_baseline"]["value"])
PY
