#!/bin/bash
# round 6, first GPU call: hierarchical grid barrier and wave-0 panel-QR factor — timing (EVD probe both barriers, C3-shard bench, QR forms) and the GPU suite
# (the switch was ISLE_TD_FLATBAR when this ran; it is ISLE_TD_BAR = flat | hier since, the sharded form being the default)
set -o pipefail
O=gpurun_out/r06_a; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
echo "== evd probe, hierarchical barrier" | tee $O/evd.log
timeout -k 10 200 python3 tools/evd_probe.py 400 2000 2>&1 | tee -a $O/evd.log || exit 1
echo "== evd probe, one counter (ISLE_TD_FLATBAR=1)" | tee -a $O/evd.log
ISLE_TD_FLATBAR=1 timeout -k 10 200 python3 tools/evd_probe.py 400 2000 2>&1 | tee -a $O/evd.log || exit 1
echo "== c3shard bench (default)"
timeout -k 10 400 python3 bench.py --workload c3shard --steps 3 --warmup 1 --no-cpu-baseline --no-upstream > $O/shard.json 2> $O/shard.err || { tail -20 $O/shard.err; exit 1; }
python3 - $O/shard.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "device", d["device_ms_per_step"])
print({k:(v["frac"],v["achieved"],v["unit"]) for k,v in d["roofline_by_family"].items()})
PY
echo "== c3shard bench (ISLE_QR_FUSED=1)"
ISLE_QR_FUSED=1 timeout -k 10 400 python3 bench.py --workload c3shard --steps 3 --warmup 1 --no-cpu-baseline --no-upstream > $O/shard_fused.json 2> $O/shard_fused.err || { tail -20 $O/shard_fused.err; exit 1; }
python3 - $O/shard_fused.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "device", d["device_ms_per_step"])
PY
echo "== GPU suite"
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.log
