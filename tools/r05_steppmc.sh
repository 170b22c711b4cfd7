#!/bin/bash
# round 5: HBM-side traffic of every kernel of one config-3 step (FETCH_SIZE and WRITE_SIZE in their own passes, kernel trace only beside them)
set -o pipefail
O=gpurun_out/r05_steppmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cn in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 550 rocprofv3 --kernel-trace --pmc $cn --output-format csv -d $O/$cn -o c -- python3 tools/pmc_step.py > $O/$cn.log 2>&1 || { tail -5 $O/$cn.log; exit 1; }
  f=$(find $O/$cn -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summarise.py $f | head -45 > $O/$cn.txt
  find $O/$cn -name "*.csv" -size +8M -delete
  echo "== $cn"; head -40 $O/$cn.txt | cut -c1-150
done
