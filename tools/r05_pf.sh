#!/bin/bash
# round 5: the full-pass threshold of Lloyd in span(U) (ISLE_PROJ_FULL_NUM / _DEN, api_kmeans.cpp) under variant builds
# (tools/build_variant.sh <name> "-DISLE_PROJ_FULL_NUM=a -DISLE_PROJ_FULL_DEN=b" api_kmeans.cpp), bench lines side by side on one box
set -o pipefail
O=gpurun_out/r05_pf; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in new pf50; do
  if [ $v != new ]; then export ISLE_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libisle_$v.so; else unset ISLE_HIP_LIB; fi
  timeout -k 10 400 python bench.py --workload c3shard --steps 4 --warmup 1 --no-upstream --no-cpu-baseline > $O/s_$v.json 2> $O/s_$v.err || { tail -5 $O/s_$v.err; exit 1; }
  python3 - $O/s_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c3shard", sys.argv[2], "ms_per_step %.1f"%d["ms_per_step"], "lloyd_proj %.1f"%d["device_ms_per_step"]["lloyd_proj"], "wall lloyd_projected %.1f"%d["host_wall_ms_per_step"]["lloyd_projected"])
PY
done
unset ISLE_HIP_LIB
timeout -k 10 900 python -m pytest tests/test_gpu_big_k.py tests/test_gpu_k_variants.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
