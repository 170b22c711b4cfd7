"""Exercises every RCCL call site of the sharded path on ONE GPU: a 1-rank communicator (ISLE_FORCE_COMM=1) makes
all-reduce / all-gather the identity, so the result must equal the communicator-free run.  (Real multi-GPU runs
are the driver's; this catches wrong datatypes, counts, in-place misuse and stream-ordering bugs.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, json, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from conftest import corpus
from isle_amd import HotPath
B = corpus(2000, 5000, 20, 0)
k = 20
hp = HotPath(0)
import os
if os.environ.get("ISLE_FORCE_COMM"):
    hp.comm_init(1, 0, HotPath.comm_unique_id())
from tools.synth import Corpus
cc = Corpus(1500, 4000, 10, 5)
cnt, rows, offs = cc.A()
hp.upload_counts(1500, cnt, rows, offs, doc_offset=0, docs_global=4000)
ti = hp.threshold(10)   # all-reduce of corpus statistics + histogram, all-gather of surviving column counts
Bd = hp.get_B()
thr_sig = [Bd["D"], Bd["nnz"], int(Bd["rows"].astype(np.int64).sum()), float(Bd["vals"].astype(np.float64).sum()),
           int(Bd["original_cols"].astype(np.int64).sum()), ti["entries_above_threshold"], hp.doc_offset, hp.D_global]
hp.upload_csc(B["V"], B["vals"], B["rows"], B["offs"], doc_offset=0, docs_global=B["D"])
X = np.random.default_rng(0).standard_normal((B["V"], 10)).astype(np.float32)
Z = hp.gram_apply(X)
r = hp.compute_block_ks(k, allow_noconv=True)
U = B["oracle"].block_ks(k)["U"]
hp.set_U(U)
ko = B["oracle"].kmeanspp(U, k, seed=3)
g = hp.kmeans_init_on_projected_space(k, inject_seeds=ko["seeds"])
free = hp.kmeans_init_on_projected_space(k, rng_seed=9)
lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
ls = hp.run_lloyds(k)
print(json.dumps(dict(fro=hp.frobenius(), thr=thr_sig, Z=float(np.abs(Z).sum()), ev=r["evals"].tolist(), seeds=free["seeds"].tolist(),
                      res=g["residual"], lp_it=lp["iters"], lp_assign=lp["assign"].tolist(), ls_it=ls["iters"],
                      ls_assign=ls["assign"].tolist(), cen=float(np.abs(ls["centers"]).sum()))))
''' % (ROOT, ROOT)


def run(force):
    env = dict(os.environ)
    env.pop("ISLE_FORCE_COMM", None)
    if force:
        env["ISLE_FORCE_COMM"] = "1"
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"fro"')]
    assert lines, (r.stdout[-1500:], r.stderr[-1500:])
    return json.loads(lines[-1])


def test_one_rank_communicator_is_the_identity():
    a, b = run(False), run(True)
    assert abs(a["fro"] - b["fro"]) <= 1e-6 * a["fro"]
    assert a["thr"] == b["thr"]
    assert abs(a["Z"] - b["Z"]) <= 1e-5 * a["Z"]
    assert np.allclose(a["ev"], b["ev"], rtol=1e-5)
    assert a["seeds"] == b["seeds"]  # same host RNG, same D^2 prefix sums -> same draws
    assert abs(a["res"] - b["res"]) <= 1e-5 * abs(a["res"])
    assert a["lp_it"] == b["lp_it"] and a["ls_it"] == b["ls_it"]
    assert (np.array(a["lp_assign"]) == np.array(b["lp_assign"])).mean() >= 0.999
    assert (np.array(a["ls_assign"]) == np.array(b["ls_assign"])).mean() >= 0.999
    assert abs(a["cen"] - b["cen"]) <= 1e-4 * a["cen"]


STALL = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from conftest import corpus
from isle_amd import HotPath
from isle_amd._lib import IsleHipError
B = corpus(2000, 5000, 20, 0)
hp = HotPath(0)
hp.comm_init(1, 0, HotPath.comm_unique_id())
hp.upload_csc(B["V"], B["vals"], B["rows"], B["offs"], doc_offset=0, docs_global=B["D"])
X = np.random.default_rng(0).standard_normal((B["V"], 10)).astype(np.float32)
try:
    hp.gram_apply(X)
    print("NO ERROR")
except IsleHipError as e:
    print("ERROR %%s" %% e)
''' % (ROOT, ROOT)


def test_a_collective_that_does_not_complete_becomes_an_error():
    """The watchdog of the RCCL collectives (api.cpp wd_*): ISLE_TEST_STALL_MS queues a kernel that spins for 4 s ahead of the all-reduce of
    the Gram apply, ISLE_COMM_TIMEOUT_S=1 lets the watchdog give up after 1 s without a completed collective: the communicator is aborted
    and the call returns ISLE_E_COMM (-5) instead of blocking; with the default timeout the same stall is simply waited for."""
    env = dict(os.environ, ISLE_FORCE_COMM="1", ISLE_TEST_STALL_MS="4000", ISLE_COMM_TIMEOUT_S="1")
    r = subprocess.run([sys.executable, "-c", STALL], capture_output=True, text=True, env=env, timeout=300)
    assert "ERROR isle_hip error -5" in r.stdout and "did not complete within 1 s" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
    assert "aborting the communicator" in r.stderr
    env["ISLE_COMM_TIMEOUT_S"] = "300"
    env["ISLE_TEST_STALL_MS"] = "1500"
    r = subprocess.run([sys.executable, "-c", STALL], capture_output=True, text=True, env=env, timeout=300)
    assert "NO ERROR" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_comm_selftest_over_rccl_at_world_size_one():
    """ISLE_COMM_SELFTEST=1 on the forced 1-rank RCCL communicator: every (operation, datatype, size class) the step issues runs once over
    RCCL itself — ncclAllReduce sum / max and ncclAllGather on f32 / f64 / i32 / u32 / u64 from one element to 16 M — and is checked against
    its closed-form result; the rank logs its device and PCI bus id.  (With more than one rank the self-test runs by default.)"""
    env = dict(os.environ, ISLE_FORCE_COMM="1", ISLE_COMM_SELFTEST="1")
    env.pop("ISLE_TEST_STALL_MS", None)
    r = subprocess.run([sys.executable, "-c", STALL], capture_output=True, text=True, env=env, timeout=300)
    assert "NO ERROR" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
    assert "rank 0 of 1: device 0 (PCI " in r.stderr and "RCCL " in r.stderr, r.stderr[-1500:]
    assert "communicator self-test: 72 collectives" in r.stderr and "correct in" in r.stderr
