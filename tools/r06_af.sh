#!/bin/bash
# Lloyd in span(U): the share of active documents beyond which an iteration runs the full product instead of the active rows' (same trajectory: bounds only)
set -o pipefail
O=gpurun_out/r06_af; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for v in default pf25 pf12 pf35; do
  if [ $v = default ]; then unset ISLE_HIP_LIB; else export ISLE_HIP_LIB=$PWD/tools/variants/libisle_$v.so; fi
  ISLE_DEBUG_HAMERLY=1 timeout -k 10 400 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-upstream > $O/$v.json 2> $O/$v.err || { tail -5 $O/$v.err; exit 1; }
  python3 - $O/$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); dm=d["device_ms_per_step"]
print(sys.argv[2], "ms_per_step", d["ms_per_step"], "lloyd_proj", dm["lloyd_proj"], "iters", d["config"].get("kmeans",{}).get("lloyd_projected_iterations"))
PY
  grep "tile bounds, projected" $O/$v.err | tail -10 | cut -c1-120
done
