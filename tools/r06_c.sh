#!/bin/bash
# round 6, third GPU call: timing-only build for the Gram-apply pricing (a fifth fewer slots), configs 4 / 5 at their own size (tests, then bench lines)
set -o pipefail
O=gpurun_out/r06_c; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for wl in c3full c3shard; do
  echo "== gram probe $wl: regular build" | tee -a $O/gram_ab256.log
  GRAM_PROBE_WORKLOAD=$wl GRAM_PROBE_ROUNDS=2 timeout -k 10 300 python3 tools/gram_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/gram_ab256.log || exit 1
  echo "== gram probe $wl: timing-only build GL_ABLATE=256 (every fifth (band, group) of a wave skipped with its ids; results wrong by construction)" | tee -a $O/gram_ab256.log
  ISLE_HIP_LIB=$PWD/tools/variants/libisle_ab256.so GRAM_PROBE_WORKLOAD=$wl GRAM_PROBE_ROUNDS=2 timeout -k 10 300 python3 tools/gram_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/gram_ab256.log || exit 1
done
echo "== configs 4 / 5 at their own size"
timeout -k 10 900 python3 -m pytest "tests/test_gpu_full_size.py::test_config4_at_its_own_size" "tests/test_gpu_full_size.py::test_config5_at_its_own_size" -m gpu -x -q --durations=5 2>&1 | tail -30 | tee $O/pytest.log
