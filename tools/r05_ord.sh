#!/bin/bash
set -o pipefail
O=gpurun_out/r05_ord; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_big_k.py tests/test_gpu_kmeans_bounds.py tests/test_gpu_k_variants.py tests/test_gpu_multirank.py -m gpu -x -q 2>&1 | tail -3
for w in c3shard; do
 for s in "-" "ISLE_YY_ORDER=member"; do
  if [ "$s" = "-" ]; then envs=""; else envs=$s; fi
  env $envs timeout -k 10 400 python bench.py --workload $w --steps 4 --warmup 1 --no-upstream --no-cpu-baseline > $O/x.json 2> $O/x.err || { tail -5 $O/x.err; exit 1; }
  python3 - $O/x.json "$w $s" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-34s ms_per_step %.1f sparse_assign %.1f wall lloyd_sparse %.1f"%(sys.argv[2], d["ms_per_step"], d["device_ms_per_step"]["sparse_assign"], d["host_wall_ms_per_step"]["lloyd_sparse"]))
PY
 done
done
