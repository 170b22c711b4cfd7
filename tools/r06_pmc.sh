#!/bin/bash
# round 6: HBM traffic of the Gram apply from the PMC counters (separate FETCH_SIZE / WRITE_SIZE passes with --kernel-trace only), per workload;
# the summaries and the JSON entries come back under gpurun_out/r06_pmc (the entries are applied to profiles/pmc_traffic.json in the build container)
set -o pipefail
O=gpurun_out/r06_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in "$@"; do
  for cn in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $cn $wl"
    timeout -k 10 500 rocprofv3 --kernel-trace --pmc $cn --output-format csv -d $O/${wl}_$cn -o p -- python3 tools/pmc_probe.py $wl > $O/${wl}_$cn.log 2>&1 || { tail -5 $O/${wl}_$cn.log; exit 1; }
    f=$(find $O/${wl}_$cn -name "*counter_collection.csv" | head -1)
    python3 tools/pmc_summarise.py "$f" | head -8 | tee $O/${wl}_${cn}_summary.txt
    # keep only the rows of the kernels the figure uses (the raw file holds every dispatch of the probe)
    python3 - "$f" $O/${wl}_${cn}_rows.csv <<'PY'
import csv,sys
keep=("gl_apply_k","gl_pack_scale_k","gl_reduce_cm_k","sumsq_k")
with open(sys.argv[1],newline="") as f, open(sys.argv[2],"w",newline="") as g:
    r=csv.DictReader(f); w=csv.DictWriter(g,fieldnames=["Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"]); w.writeheader()
    for row in r:
        hit=[k for k in keep if k in row["Kernel_Name"]]
        if hit:
            w.writerow({"Dispatch_Id":row["Dispatch_Id"],"Kernel_Name":hit[0],"Counter_Name":row["Counter_Name"],"Counter_Value":row["Counter_Value"]})
PY
    rm -rf $O/${wl}_$cn
  done
  grep -h "^nnz" $O/${wl}_FETCH_SIZE.log | tail -1 > $O/${wl}_nnz.txt
done
