#!/bin/bash
# round 6: orthogonalisation kernels with twice the loads in flight (ORTHO_UNROLL=2 variant build) against the regular build, C3 shard
set -o pipefail
O=gpurun_out/r06_i; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for v in regular ortho2 regular ortho2; do
  if [ $v = ortho2 ]; then export ISLE_HIP_LIB=$PWD/tools/variants/libisle_ortho2.so; else unset ISLE_HIP_LIB; fi
  timeout -k 10 300 python3 bench.py --workload c3shard --steps 3 --warmup 1 --no-cpu-baseline --no-upstream > $O/$v.json 2> $O/$v.err || { tail -20 $O/$v.err; exit 1; }
  python3 - $O/$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms_per_step", d["ms_per_step"], "ortho", d["device_ms_per_step"]["ortho"], "sigma", d["accuracy"]["sigma_rel_err_bound"] if "accuracy" in d else None, "evals0", d["config"]["block_ks"])
PY
done
