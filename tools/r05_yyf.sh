#!/bin/bash
# round 5: the by-group Yinyang iteration, settings side by side under rocprof (kernel stats), then the k-means tests
set -o pipefail
O=gpurun_out/r05_yyf; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in ${VARIANTS:-new}; do
  if [ $v != new ]; then export ISLE_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/libisle_$v.so; else unset ISLE_HIP_LIB; fi
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$v -o s -- python3 tools/yy_probe.py c3full ${SETTINGS:-ISLE_YY_MODE=group ISLE_YY_MODE=group,ISLE_YY_PIPE=0} > $O/$v.log 2>&1 || { tail -5 $O/$v.log; exit 1; }
  f=$(find $O/p_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep run_lloyds $O/$v.log | cut -c1-250
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("yy_filter_k","yy2_","void yy2_","yy_scan","pt_","void pt_","kmpp_","void kmpp_","tiles_comb","yy_first","cc_")): print("   %-34s calls %4s avg_us %9.1f" % (r["Name"][:34], r["Calls"], float(r["AverageNs"])/1e3))
PY
  find $O -name "*kernel_trace.csv" -delete
done
unset ISLE_HIP_LIB
if [ -z "$NOTESTS" ]; then timeout -k 10 900 python -m pytest tests/test_gpu_big_k.py tests/test_gpu_kmeans_bounds.py -m gpu -x -q > $O/pytest.log 2>&1; echo rc=$?; tail -3 $O/pytest.log; fi
