#!/bin/bash
# round 6, closing call: the step's counter passes on the final library sources, then the whole GPU suite, smoke() and the driver's command
set -o pipefail
cd "$GRAFT_REPO_ROOT"
bash tools/r06_steppmc.sh > gpurun_out/r06_steppmc_final.log 2>&1 || { tail -5 gpurun_out/r06_steppmc_final.log; exit 1; }
cp gpurun_out/r06_steppmc/*_by_kernel.csv profiles/r06_steppmc/
python3 tools/step_traffic_to_json.py | tail -2
mkdir -p gpurun_out/r06_y; cp profiles/r06_step_traffic_by_family.json gpurun_out/r06_y/
bash tools/r06_z.sh
