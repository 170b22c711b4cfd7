#!/bin/bash
# round 6: HBM-side traffic of every kernel of one config-3 step (FETCH_SIZE and WRITE_SIZE in passes of their own, kernel trace only beside them),
# reduced on the box to one row per kernel: dispatches, counter sum
set -o pipefail
O=gpurun_out/r06_steppmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cn in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 560 rocprofv3 --kernel-trace --pmc $cn --output-format csv -d $O/$cn -o c -- python3 tools/pmc_step.py > $O/$cn.log 2>&1 || { tail -5 $O/$cn.log; exit 1; }
  f=$(find $O/$cn -name "*counter_collection.csv" | head -1)
  python3 - "$f" $cn $O/${cn}_by_kernel.csv <<'PY'
import csv, sys
sys.path.insert(0, "tools")
import make_kernel_table as mk
acc, disp = {}, {}
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        if r.get("Counter_Name") != sys.argv[2]:
            continue
        k = mk.short(r["Kernel_Name"])
        acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
        disp.setdefault(k, set()).add(r["Dispatch_Id"])
with open(sys.argv[3], "w", newline="") as g:
    w = csv.writer(g)
    w.writerow(["kernel", "dispatches", sys.argv[2] + "_KB_sum"])
    for k in sorted(acc, key=lambda x: -acc[x]):
        w.writerow([k, len(disp[k]), "%.1f" % acc[k]])
PY
  rm -rf $O/$cn
  echo "== $cn"; head -12 $O/${cn}_by_kernel.csv | cut -c1-150
done
