#!/bin/bash
# round 6: the eigenvectors' back-transformation by blocks of four reflectors (compact WY) against the sequential form
set -o pipefail
O=gpurun_out/r06_aa; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python3 -m pytest tests/test_gpu_big_k.py tests/test_gpu_ks_dense.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
for v in "" seq ""; do
  echo "== ISLE_TD_BACK=$v" | tee -a $O/evd_probe.log
  ISLE_TD_BACK=$v timeout -k 10 300 python3 tools/evd_probe.py 400 1000 2010 2>&1 | grep -v amdgpu.ids | tee -a $O/evd_probe.log
done
