#!/bin/bash
# round 5: the filter phase of the by-group Yinyang iteration, HEAD build against the working tree's, k = 1000 and 1024
set -o pipefail
O=gpurun_out/r05_yyf; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in new reg8; do
  for kk in 1000; do
    export PROBE_K=$kk
    if [ $v != new ]; then export ISLE_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libisle_$v.so; else unset ISLE_HIP_LIB; fi
    timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_${v}_$kk -o s -- python3 tools/yy_probe.py c3full ISLE_YY_MODE=group ISLE_YY_MODE=group,ISLE_YY_FUSED=0 > $O/${v}_$kk.log 2>&1 || exit 1
    f=$(find $O/p_${v}_$kk -name "*kernel_stats.csv" | head -1)
    echo "== $v k=$kk"; grep run_lloyds $O/${v}_$kk.log | cut -c1-200
    python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("yy_filter_k","yy2_filter_tighten_k","yy2_tighten_k","yy2_scan_k","yy2_pack")): print("   %-28s calls %4s avg_us %9.1f" % (r["Name"][:28], r["Calls"], float(r["AverageNs"])/1e3))
PY
    find $O -name "*kernel_trace.csv" -delete
  done
done
unset ISLE_HIP_LIB PROBE_K
timeout -k 10 900 python -m pytest tests/test_gpu_big_k.py tests/test_gpu_kmeans_bounds.py -m gpu -x -q > $O/pytest.log 2>&1; echo rc=$?; tail -3 $O/pytest.log
