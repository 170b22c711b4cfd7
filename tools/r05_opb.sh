#!/bin/bash
# round 5: operator build time under differently compiled libraries (bench.py's device_ms_per_step.op_build, 3 steps)
set -o pipefail
O=gpurun_out/r05_opb; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in ${VARIANTS:-new}; do
  if [ $v != new ]; then export ISLE_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libisle_$v.so; else unset ISLE_HIP_LIB; fi
  timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/$v.json 2> $O/$v.err || { tail -5 $O/$v.err; exit 1; }
  python3 - $O/$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms_per_step %.1f"%d["ms_per_step"], {k:round(v,1) for k,v in d["device_ms_per_step"].items()})
PY
done
