#!/bin/bash
set -o pipefail
O=gpurun_out/r06_n; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_multirank.py tests/test_gpu_comm_selftest.py tests/test_gpu_bench_contract.py -m gpu -x -q --durations=5 2>&1 | tail -12 | tee $O/pytest.log
grep -q "failed" $O/pytest.log && exit 1
bash tools/r06_steppmc.sh
