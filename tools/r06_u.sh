#!/bin/bash
# update_mfma_k: one or two sets of 16 columns in flight per wave (variant library against the in-tree build)
set -o pipefail
bash tools/r06_t.sh um2 > /dev/null || exit 1
ISLE_HIP_LIB=$PWD/tools/variants/libisle_um1.so bash tools/r06_t.sh um1 > /dev/null || exit 1
for t in um2 um1; do echo "== $t"; grep -A2 "^update_mfma_k" gpurun_out/r06_t/$t/ortho_by_width.txt; grep -A1 "^vtf_mfma_k" gpurun_out/r06_t/$t/ortho_by_width.txt; python3 -c "
import json;d=json.loads(open('gpurun_out/r06_t/$t/b.json').read().strip().splitlines()[-1]);print('   ms_per_step',d['ms_per_step'],'ortho',d['device_ms_per_step']['ortho'])"; done
