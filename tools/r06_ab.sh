#!/bin/bash
set -o pipefail
O=gpurun_out/r06_ab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in wy seq; do
  if [ $v = seq ]; then export ISLE_TD_BACK=seq; else unset ISLE_TD_BACK; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$v -o s -- python3 tools/evd_probe.py 1000 2010 > $O/$v.log 2>&1 || { tail -5 $O/$v.log; exit 1; }
  f=$(find $O/p_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "td_" in r["Name"]: print("   %-60s calls %4s avg_us %10.1f min %10.1f max %10.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
  find $O -name "*kernel_trace.csv" -delete
done
