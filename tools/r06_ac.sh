#!/bin/bash
set -o pipefail
O=gpurun_out/r06_ac; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_big_k.py tests/test_gpu_ks_dense.py tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_gpu_k_variants.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
for v in "" seq; do
  ISLE_TD_BACK=$v timeout -k 10 300 python3 bench.py --workload c3shard --steps 3 --warmup 1 --no-cpu-baseline --no-upstream > $O/shard_$v.json 2> $O/shard_$v.err || { tail -5 $O/shard_$v.err; exit 1; }
  python3 - $O/shard_$v.json "$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ISLE_TD_BACK=%s ms_per_step %.1f evd %.2f qr %.2f sigma %s" % (sys.argv[2], d["ms_per_step"], d["device_ms_per_step"]["evd"], d["device_ms_per_step"]["qr"], d["accuracy"]["sigma_rel_err_bound"]))
PY
done
