#!/bin/bash
# round 6: the whole GPU suite at HEAD (as the driver runs it), then smoke()
set -o pipefail
O=gpurun_out/r06_h; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -25 | tee $O/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
