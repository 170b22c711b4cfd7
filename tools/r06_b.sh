#!/bin/bash
# round 6, second GPU call: the new tests (configs 4 / 5 at their own size, bench contract incl. the c4 / c5 flows, communicator self-test, multirank)
set -o pipefail
O=gpurun_out/r06_b; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1100 python3 -m pytest tests/test_gpu_comm_selftest.py tests/test_gpu_multirank.py tests/test_gpu_bench_contract.py "tests/test_gpu_full_size.py::test_config4_at_its_own_size" "tests/test_gpu_full_size.py::test_config5_at_its_own_size" -m gpu -x -q --durations=15 2>&1 | tail -40 | tee $O/pytest.log
