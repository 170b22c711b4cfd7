#!/bin/bash
# round 5: bench.py (config 3, 3 steps) under environment settings given as arguments ("-" = none), same box
set -o pipefail
O=gpurun_out/r05_env; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for setting in "$@"; do
  i=$((i+1))
  if [ "$setting" = "-" ]; then envs=""; else envs=$(echo $setting | tr ',' ' '); fi
  env $envs timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/$i.json 2> $O/$i.err || { tail -5 $O/$i.err; exit 1; }
  python3 - $O/$i.json "$setting" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-40s ms_per_step %.1f"%(sys.argv[2], d["ms_per_step"]), {k:round(v,1) for k,v in d["device_ms_per_step"].items() if k in ("ortho","qr","evd","rotate","kmpp","lloyd_proj","sparse_assign","op_build")}, "sigma %.1e"%d["accuracy"]["sigma_rel_err_bound"])
PY
done
