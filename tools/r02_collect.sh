#!/bin/bash
# Collects the round's measurements on the GPU box (run through gpurun from the repo root); summaries are copied into profiles/ afterwards.
set -o pipefail
O=gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== bench c2 (driver's default command)"; timeout -k 10 400 python bench.py > $O/c2_bench.json 2> $O/c2_bench.err; echo rc=$?
echo "== rocprof stats c2"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o c2 -- python3 bench.py --steps 3 --warmup 1 --no-upstream --no-cpu-baseline > $O/c2_bench_under_rocprof.json 2> $O/c2_rocprof.err; echo rc=$?
echo "== bench c3shard"; timeout -k 10 400 python bench.py --workload c3shard --steps 2 --warmup 1 --no-upstream > $O/c3shard_bench.json 2> $O/c3shard_bench.err; echo rc=$?
echo "== rocprof stats c3shard"; timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python3 bench.py --workload c3shard --steps 1 --warmup 1 --no-upstream --no-cpu-baseline > $O/c3shard_bench_under_rocprof.json 2> $O/c3_rocprof.err; echo rc=$?
if [ -z "$SKIP_PMC" ]; then
echo "== pmc FETCH_SIZE"; timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 tools/pmc_probe.py > $O/pmc_fetch.log 2>&1; echo rc=$?
echo "== pmc WRITE_SIZE"; timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 tools/pmc_probe.py > $O/pmc_write.log 2>&1; echo rc=$?
echo "== pmc MFMA busy (c3shard k-means + rotation kernels)"; timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_mfma -o m -- python3 bench.py --workload c3shard --steps 1 --warmup 0 --no-upstream --no-cpu-baseline > $O/pmc_mfma.json 2> $O/pmc_mfma.err; echo rc=$?
echo "== pmc MFMA busy c2"; timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_mfma_c2 -o m -- python3 bench.py --steps 1 --warmup 0 --no-upstream --no-cpu-baseline > $O/pmc_mfma_c2.json 2> $O/pmc_mfma_c2.err; echo rc=$?
fi
echo "== bench c2 with the centres copied to the host (PCIe-inclusive figure)"; timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-upstream --no-cpu-baseline --fetch-centers > $O/c2_bench_fetch_centers.json 2> $O/c2_bench_fetch_centers.err; echo rc=$?
echo "== bench c3shard with the centres copied to the host"; timeout -k 10 400 python bench.py --workload c3shard --steps 1 --warmup 1 --no-upstream --no-cpu-baseline --fetch-centers > $O/c3shard_bench_fetch_centers.json 2> $O/c3shard_bench_fetch_centers.err; echo rc=$?
echo "== bench c1"; timeout -k 10 300 python bench.py --workload c1 --steps 5 --warmup 1 --no-upstream > $O/c1_bench.json 2> $O/c1_bench.err; echo rc=$?
echo "== bench c3full"; timeout -k 10 600 python bench.py --workload c3full --steps 1 --warmup 1 --no-upstream --no-cpu-baseline > $O/c3full_bench.json 2> $O/c3full_bench.err; echo rc=$?
# keep the merged output small: traces are only needed as per-kernel statistics
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*counter_collection.csv" -size +30M -delete
du -sh $O; ls $O/prof_c2 | head
