#!/usr/bin/env python3
"""A small hot-path pass with ISLE_ROCTX=1 for `rocprofv3 --marker-trace --kernel-trace`: which roctx ranges (kernel families) appear."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ISLE_ROCTX"] = "1"
from isle_amd import HotPath  # noqa: E402
from tools.synth import make_B  # noqa: E402

k = 20
B = make_B(2000, 5000, k, 0)
hp = HotPath(0)
hp.upload_csc(B["V"], B["vals"], B["rows"], B["offs"])
hp.compute_block_ks(k, allow_noconv=True)
g = hp.kmeans_init_on_projected_space(k, rng_seed=1)
lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
hp.run_lloyds(k)
hp.close()
print("done")
