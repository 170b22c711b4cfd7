#!/bin/bash
# forced 1-rank RCCL communicator inside bench.py's process (torch's bundled RCCL)
set -o pipefail
O=gpurun_out/r06_s; mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q -k "one_rank_rccl or single_gpu_line" 2>&1 | tail -15 | tee $O/pytest.log
ISLE_FORCE_COMM=1 ISLE_COMM_SELFTEST=1 timeout -k 10 300 python3 bench.py --workload tiny --steps 1 --warmup 1 > $O/bench_forced.json 2> $O/bench_forced.err
grep -i "rccl\|self-test\|rank 0" $O/bench_forced.err | head -10
