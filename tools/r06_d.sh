#!/bin/bash
# round 6, fourth GPU call: timing experiment for the Gram-apply pricing through the unchanged kernel (count records zeroed: ISLE_GL_ABLATE_SKIP),
# config 5 at its own size again, kernel statistics of a C3-shard step
set -o pipefail
O=gpurun_out/r06_d; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for wl in c3full c3shard; do
  echo "== gram probe $wl (regular kernel; ISLE_GL_ABLATE_SKIP=m: every m-th (band, group) of a wave not walked, its ids not read; results wrong by construction)" | tee -a $O/gram_skip.log
  GRAM_PROBE_WORKLOAD=$wl timeout -k 10 400 python3 tools/gram_probe.py "" "ISLE_GL_ABLATE_SKIP=5" "ISLE_GL_ABLATE_SKIP=3" "ISLE_GL_ABLATE_SKIP=2" "" 2>&1 | grep -v amdgpu.ids | tee -a $O/gram_skip.log || exit 1
done
echo "== config 5 at its own size"
timeout -k 10 600 python3 -m pytest "tests/test_gpu_full_size.py::test_config5_at_its_own_size" -m gpu -x -q --durations=3 2>&1 | tail -12 | tee $O/pytest.log
echo "== kernel statistics of a C3-shard step"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shard -o s -- python3 bench.py --workload c3shard --steps 3 --warmup 1 --no-cpu-baseline --no-upstream > $O/shard_rocprof.json 2> $O/shard_rocprof.err || { tail -5 $O/shard_rocprof.err; exit 1; }
f=$(find $O/prof_shard -name "*kernel_stats.csv" | head -1); cp "$f" $O/shard_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python3 - $O/shard_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:45]:
    print("%-70s calls %6s total_ms %9.2f avg_us %9.2f"%(r["Name"].split("(")[0][-70:], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
