#!/bin/bash
set -o pipefail
O=gpurun_out/r06_k; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_ks_dense.py "tests/test_gpu_big_k.py" -m gpu -x -q --durations=5 2>&1 | tail -15 | tee $O/pytest.log
