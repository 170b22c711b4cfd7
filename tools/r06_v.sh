#!/bin/bash
# round 6, after the orthogonalisation change: kernel statistics of config 3 and of a C3 shard at HEAD, then the step's counter passes
set -o pipefail
O=gpurun_out/r06_v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== kernel statistics, config 3 (3 timed steps)"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/c3_rocprof.json 2> $O/c3_rocprof.err || { tail -5 $O/c3_rocprof.err; exit 1; }
f=$(find $O/prof_c3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/c3full_kernel_stats.csv
echo "== kernel statistics of a C3-shard step"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shard -o s -- python3 bench.py --workload c3shard --steps 3 --warmup 1 --no-cpu-baseline --no-upstream > $O/shard_rocprof.json 2> $O/shard_rocprof.err || { tail -5 $O/shard_rocprof.err; exit 1; }
f=$(find $O/prof_shard -name "*kernel_stats.csv" | head -1); cp "$f" $O/c3shard_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
head -8 $O/c3full_kernel_stats.csv | cut -c1-160

