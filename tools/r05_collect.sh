#!/bin/bash
# Collects round-5 measurements on the GPU box (run through gpurun from the repo root); summaries are copied into profiles/ afterwards.
# usage: tools/r05_collect.sh <tag> <stage> [<stage> ...]     stages: benchfull bench prof pmc c2 shard tests
set -o pipefail
TAG=$1; shift
O=gpurun_out/r05_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for st in "$@"; do
case $st in
bench) echo "== bench default (c3 on one GPU), 5 steps"; timeout -k 10 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/c3full_bench.json 2> $O/c3full_bench.err; echo rc=$?; tail -c 400 $O/c3full_bench.err;;
benchfull) echo "== bench: the driver's command"; T0=$(date +%s); timeout -k 10 1100 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/c3full_bench_driver_cmd.json 2> $O/c3full_bench_driver_cmd.err; echo rc=$? wall_s=$(( $(date +%s) - T0 )); tail -c 600 $O/c3full_bench_driver_cmd.err;;
prof) echo "== rocprof stats c3 (1 step)"; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3full -o c3full -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/c3full_bench_under_rocprof.json 2> $O/c3full_rocprof.err; echo rc=$?;;
pmc) echo "== pmc FETCH_SIZE c3full"; timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 tools/pmc_probe.py c3full > $O/pmc_fetch.log 2>&1; echo rc=$?
     echo "== pmc WRITE_SIZE c3full"; timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 tools/pmc_probe.py c3full > $O/pmc_write.log 2>&1; echo rc=$?
     for f in $(find $O/pmc_fetch $O/pmc_write -name "*counter_collection.csv"); do echo $f; python3 tools/pmc_summarise.py $f | head -12; done > $O/pmc_summary.txt; cat $O/pmc_summary.txt;;
c2) echo "== bench c2"; timeout -k 10 400 python bench.py --workload c2 --steps 5 --warmup 2 > $O/c2_bench.json 2> $O/c2_bench.err; echo rc=$?;;
shard) echo "== bench c3shard"; timeout -k 10 400 python bench.py --workload c3shard --steps 3 --warmup 1 --no-upstream --no-cpu-baseline > $O/c3shard_bench.json 2> $O/c3shard_bench.err; echo rc=$?;;
tests) echo "== pytest -m gpu"; timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo rc=$?; tail -5 $O/pytest_gpu.log;;
esac
done
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*counter_collection.csv" -size +30M -delete
du -sh $O
