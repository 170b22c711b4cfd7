#!/bin/bash
# round 6: slab loads in flight in gl_reduce_cm_k (variant builds), config 3: pass-2 scope = pass 2 + reduce in tools/gram_probe.py
set -o pipefail
O=gpurun_out/r06_q; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in regular red4 red16 red26; do
  if [ $v = regular ]; then unset ISLE_HIP_LIB; else export ISLE_HIP_LIB=$PWD/tools/variants/libisle_$v.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$v -o s -- python3 tools/pmc_probe.py c3full > $O/$v.log 2>&1 || { tail -5 $O/$v.log; exit 1; }
  f=$(find $O/p_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v" | tee -a $O/reduce.log; python3 - "$f" <<'PY' | tee -a $O/reduce.log
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gl_reduce_cm_k" in r["Name"] or "gl_apply_k" in r["Name"]:
        print("   %-40s calls %4s avg_us %9.1f" % (r["Name"].split("(")[1][:40] if r["Name"].startswith("void (") else r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $O/p_$v
done
