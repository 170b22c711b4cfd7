#!/bin/bash
# round 5: counters of the by-group Yinyang kernels (one --pmc pass per counter set; kernel trace only beside them)
set -o pipefail
O=gpurun_out/r05_yypmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "FETCH_SIZE" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o c -- python3 tools/yy_probe.py c3full ISLE_YY_MODE=group > $O/p$i.log 2>&1
  rc=$?; echo "== set $i: $set rc=$rc"
  if [ $rc = 0 ]; then
    f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
    python3 tools/pmc_summarise.py $f | grep -E "^yy2_scan_k|^yy2_filter_tighten_k|^yy2_combine|^yy_scan" | head -12
    find $O/p$i -name "*.csv" -size +5M -delete
  else tail -3 $O/p$i.log; fi
done
