#!/bin/bash
# does the EVD's back-transformation form change the time of unrelated memory-bound kernels?  (config 3, same box, alternating)
set -o pipefail
O=gpurun_out/r06_ad; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for v in wy seq wy seq; do
  if [ $v = seq ]; then export ISLE_TD_BACK=seq; else unset ISLE_TD_BACK; fi
  timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-upstream > $O/$v.json 2> $O/$v.err || { tail -5 $O/$v.err; exit 1; }
  python3 - $O/$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); dm=d["device_ms_per_step"]
print(sys.argv[2], d["ms_per_step"], {k:round(v,1) for k,v in dm.items()})
PY
done
