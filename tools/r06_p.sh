#!/bin/bash
# round 6: the tagged exchange of the persistent tridiagonalisation against the sharded barrier (EVD probe), and the forms' bit equality
set -o pipefail
O=gpurun_out/r06_p; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for v in sharded tagged sharded tagged; do
  case $v in sharded) unset ISLE_TD_BAR;; tagged) export ISLE_TD_BAR=tagged;; esac
  echo "== $v" | tee -a $O/evd.log
  timeout -k 10 200 python3 tools/evd_probe.py 400 1000 2000 2>&1 | grep -v amdgpu | tee -a $O/evd.log || exit 1
done
unset ISLE_TD_BAR
timeout -k 10 300 python3 -m pytest tests/test_gpu_big_k.py -m gpu -x -q -k "barriers or small_evd" 2>&1 | tail -5 | tee $O/pytest.log
