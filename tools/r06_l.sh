#!/bin/bash
set -o pipefail
O=gpurun_out/r06_l; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_upload.py tests/test_gpu_bench_contract.py -m gpu -x -q --durations=5 2>&1 | tail -15 | tee $O/pytest.log
