#!/bin/bash
set -o pipefail
O=gpurun_out/r06_ae; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 700 python3 -m pytest tests/test_gpu_kmeans_bounds.py tests/test_gpu_parity.py tests/test_gpu_k_variants.py tests/test_gpu_multirank.py "tests/test_gpu_full_size.py::test_one_eighth_of_config3_at_k1000" -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o s -- python3 bench.py --workload c3shard --steps 3 --warmup 1 --no-cpu-baseline --no-upstream > $O/shard.json 2> $O/shard.err || { tail -5 $O/shard.err; exit 1; }
f=$(find $O/p -name "*kernel_stats.csv" | head -1)
python3 - "$f" $O/shard.json <<'PY'
import csv,sys,json
for r in csv.DictReader(open(sys.argv[1])):
    if any(x in r["Name"] for x in ("yy2_pack_k","scale_centers_k","cc_centers_k","colnorm_partial_k")): print("   %-40s calls %4s avg_us %9.1f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"])/1e3))
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); dm=d["device_ms_per_step"]
print("shard", d["ms_per_step"], "sparse_assign", dm["sparse_assign"], "sparse_update", dm["sparse_update"], "gate", d["accuracy"]["gate"]["passed"], d["accuracy"]["kmeans_vs_oracle"]["iterations_hip"], d["accuracy"]["kmeans_vs_oracle"]["iterations_oracle"])
PY
find $O -name "*kernel_trace.csv" -delete
