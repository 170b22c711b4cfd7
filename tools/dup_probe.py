"""Duplicate documents: how often do two IDENTICAL columns of B end on different centres?

The reference adds a document's products in one fixed order, so identical documents get bit-identical distances and `isamin` sends both to the
same centre (src/sparseMatrix.cpp:1868-1870).  Here the bank-aware placement of the id streams makes a lane's summation order depend on the
lane, so identical documents may project to different bits and split when two centres are tied to rounding.

    python tools/dup_probe.py V D k seed [fraction]     ->  one JSON line (committed under profiles/)

A `fraction` (default 0.01) of the columns of the thresholded corpus is overwritten by copies of other columns; the whole hot path runs with
its default routes; reported: pairs whose two members differ after Lloyd in span(U) and after Lloyd on B."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def with_duplicates(B, fraction, seed=3):
    """B with `fraction` of its columns replaced by copies of other columns: (B', src, dst)."""
    D = B["D"]
    rng = np.random.default_rng(seed)
    n = max(1, int(D * fraction))
    pick = rng.choice(D, 2 * n, replace=False)
    src, dst = pick[:n], pick[n:]
    cols = np.arange(D, dtype=np.int64)
    cols[dst] = src
    offs = B["offs"]
    lens = (offs[cols + 1] - offs[cols]).astype(np.int64)
    so = np.zeros(D + 1, np.int64)
    np.cumsum(lens, out=so[1:])
    idx = np.repeat(offs[cols] - so[:-1], lens) + np.arange(so[-1], dtype=np.int64)
    return dict(V=B["V"], D=D, vals=B["vals"][idx], rows=B["rows"][idx], offs=so), src, dst


def run(hp, Bd, src, dst, k):
    hp.upload_csc(Bd["V"], Bd["vals"], Bd["rows"], Bd["offs"])
    hp.compute_block_ks(k, seed=1, allow_noconv=True)
    g = hp.kmeans_init_on_projected_space(k, rng_seed=1)
    lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
    hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
    ls = hp.run_lloyds(k, fetch_centers=False)
    return dict(pairs=int(len(src)), split_after_lloyd_in_span_U=int((lp["assign"][src] != lp["assign"][dst]).sum()),
                split_after_lloyd_on_B=int((ls["assign"][src] != ls["assign"][dst]).sum()), iters=[lp["iters"], ls["iters"]])


if __name__ == "__main__":
    from isle_amd import HotPath
    from tools.synth import make_B
    V, D, k, seed = (int(x) for x in sys.argv[1:5])
    fraction = float(sys.argv[5]) if len(sys.argv) > 5 else 0.01
    B = make_B(V, D, k, seed)
    Bd, src, dst = with_duplicates(B, fraction)
    hp = HotPath(0)
    res = run(hp, Bd, src, dst, k)
    hp.close()
    res.update(shape=[V, B["D"], k], seed=seed, duplicated_fraction=fraction, operator_form="LDS-banded (bank-aware placement on)")
    print(json.dumps(res))
