#!/bin/bash
# round 5: kernel stats of the operator build (tools/pmc_probe.py c3full: upload, build, three Gram applies) under rocprofv3 --kernel-trace --stats
set -o pipefail
O=gpurun_out/r05_opstats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o s -- python3 tools/pmc_probe.py c3full > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
f=$(find $O/p -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    if n.startswith(("gl_","rs_","isle_scan","cc_","csc_")) and not n.startswith("gl_apply") and not n.startswith("gl_reduce") and not n.startswith("gl_pack"):
        t=float(r["TotalDurationNs"])/1e6; tot+=t
        print("   %-60s calls %4s total_ms %8.2f" % (n[:60], r["Calls"], t))
print("   sum %.1f ms"%tot)
PY
find $O -name "*kernel_trace.csv" -delete
