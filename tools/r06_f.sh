#!/bin/bash
# round 6, sixth GPU call: eigensolver tests behind the side-stream mailbox copy and the deeper factor batch, the shard line, then the PMC passes
set -o pipefail
O=gpurun_out/r06_f; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python3 -m pytest tests/test_gpu_ks_dense.py tests/test_gpu_multirank.py tests/test_gpu_comm_selftest.py "tests/test_gpu_parity.py" "tests/test_gpu_big_k.py" -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.log
grep -q " passed" $O/pytest.log || exit 1
grep -q "failed" $O/pytest.log && exit 1
echo "== c3shard bench"
timeout -k 10 400 python3 bench.py --workload c3shard --steps 5 --warmup 2 --no-cpu-baseline --no-upstream > $O/shard.json 2> $O/shard.err || { tail -20 $O/shard.err; exit 1; }
python3 - $O/shard.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "device", d["device_ms_per_step"])
PY
bash tools/r06_pmc.sh c3full c3shard
