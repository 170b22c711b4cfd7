#!/bin/bash
# round 5: the orthogonalisation kernels under rocprof (3 bench steps), default against ISLE_UPDATE_MFMA=0
set -o pipefail
O=gpurun_out/r05_ortho; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in mfma fma; do
  if [ $v = fma ]; then export ISLE_UPDATE_MFMA=0; else unset ISLE_UPDATE_MFMA; fi
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$v -o s -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/$v.json 2> $O/$v.err || { tail -5 $O/$v.err; exit 1; }
  f=$(find $O/p_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python3 - "$f" $O/$v.json <<'PY'
import csv,sys,json
for r in csv.DictReader(open(sys.argv[1])):
    if any(x in r["Name"] for x in ("update_k","update_mfma_k","vtf_mfma_k")): print("   %-40s calls %5s avg_us %9.1f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3))
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("   ms_per_step %.1f ortho %.1f sigma bound %s" % (d["ms_per_step"], d["device_ms_per_step"]["ortho"], d.get("accuracy",{}).get("sigma_rel_err_bound")))
PY
  find $O -name "*kernel_trace.csv" -delete
done
