#!/usr/bin/env python3
"""The orthogonalisation kernels at the shape a rank of N sees under the row-sharded step (n = V / N rows, basis widths 10 ... 2010), timed on
one GPU: the block Krylov-Schur solver on a dense symmetric n x n operator (isle_hip_block_ks_dense) runs vtf_mfma_k / update_mfma_k on n-row
panels.  Prints the device time of the `ortho` family per solve.  usage: ortho_slice_probe.py [n ...]   (default 100000/8 and 100000/4)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isle_amd import HotPath  # noqa: E402


def main():
    ns = [int(x) for x in sys.argv[1:]] or [12500, 25000]
    nev, blk = 1000, 10
    ncv = 2 * nev + blk
    hp = HotPath(0)
    for n in ns:
        rng = np.random.default_rng(n)
        t0 = time.time()
        X = rng.standard_normal((n, 2200)).astype(np.float32)
        w = (1.0 / np.arange(1, 2201) ** 0.5).astype(np.float32)
        A = (X * w) @ X.T
        A = 0.5 * (A + A.T)
        print("n = %d: operator built in %.1f s" % (n, time.time() - t0), flush=True)
        for rep in range(2):
            hp.timing_enable(1)
            hp.timing_reset()
            r = hp.block_ks_dense(A, nev, blk=blk, ncv=ncv, maxit=1, allow_noconv=True)
            hp.synchronize()
            t = hp.timing_get()
            hp.timing_enable(0)
            print("   n = %6d  rep %d: ortho %.2f ms in %d launches, qr %.2f ms, applies %s" % (n, rep, t["ortho"][0], t["ortho"][1], t["qr"][0], r.get("applies")), flush=True)
        del A, X


if __name__ == "__main__":
    main()
