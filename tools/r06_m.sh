#!/bin/bash
set -o pipefail
O=gpurun_out/r06_m; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or block_ks" --durations=5 2>&1 | tail -12 | tee $O/pytest.log
