#!/bin/bash
# round 6: the three grid barriers of the persistent tridiagonalisation against each other (EVD probe, n = 2000 and 400)
# (the switch was ISLE_TD_FLATBAR when this ran; it is ISLE_TD_BAR = flat | hier since, the sharded form being the default)
set -o pipefail
O=gpurun_out/r06_j; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for v in hier sharded flat hier sharded; do
  case $v in hier) unset ISLE_TD_FLATBAR;; sharded) export ISLE_TD_FLATBAR=sharded;; flat) export ISLE_TD_FLATBAR=1;; esac
  echo "== $v" | tee -a $O/evd.log
  timeout -k 10 200 python3 tools/evd_probe.py 400 2000 2>&1 | grep -v amdgpu | tee -a $O/evd.log || exit 1
done
