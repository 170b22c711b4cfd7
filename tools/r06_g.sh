#!/bin/bash
# round 6, seventh GPU call: the driver's command at HEAD, kernel statistics of the same workload, the C2 counter pass
set -o pipefail
O=gpurun_out/r06_g; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
echo "== driver command"
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err || { tail -20 $O/driver_cmd.err; exit 1; }
python3 - $O/driver_cmd.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["avg_launch_ms"])
print("device", d["device_ms_per_step"])
print("families", {k:v["frac"] for k,v in d["roofline_by_family"].items()})
print("c2", d["secondary_c2"]); print("c3shard", d["secondary_c3shard"]); print("cli", d["full_cli_c2"].get("wall_s"))
print("cpu", {k:v for k,v in d["cpu_baseline"].items() if k!="port_over_reference"})
print("acc", d["accuracy"]["sigma_rel_err_bound"], d["accuracy"]["kmeans_vs_oracle"]["partition_agreement_projected"], d["accuracy"]["kmeans_vs_oracle"]["partition_agreement_word_space"])
PY
echo "== kernel statistics, config 3 (3 timed steps)"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/c3_rocprof.json 2> $O/c3_rocprof.err || { tail -5 $O/c3_rocprof.err; exit 1; }
f=$(find $O/prof_c3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/c3full_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
bash tools/r06_pmc.sh c2
