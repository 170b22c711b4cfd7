"""The sharded (N > 1) path with 2 and 3 ranks on ONE GPU.

RCCL refuses two ranks on one device, so the ranks use the rehearsal transport of include/isle_hip.h
(isle_hip_comm_init_host): every collective of the library is staged through host memory and completed by gloo.  Everything
else is the product path: document shards from isle_hip_plan_shards, per-rank operator build, all-reduce of Z per operator
application, replicated eigensolver, k-means++ over per-shard D^2 prefix sums, centroid sums, stop rules.  The run with N
ranks must reproduce the single-rank run (same U and seeds injected), and the replicated results must be bit-identical on
all ranks (that is what keeps the ranks' control flow — restarts, rank decisions, stop rules — in step)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
os.environ["OMP_NUM_THREADS"] = "2"
from conftest import corpus
from isle_amd import HotPath
from tools.synth import Corpus

rccl = os.environ.get("ISLE_TEST_RCCL") == "1"   # one GPU per rank and the product's own transport (needs `world` GPUs)
hp = HotPath(rank if rccl else 0)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%d" %% port, rank=rank, world_size=world)
    if rccl:
        uid = [HotPath.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        hp.comm_init(world, rank, uid[0])
    else:
        hp.comm_init_host(world, rank, HotPath.gloo_exchange(dist, world, rank))
res = {}

# ---- thresholding on document shards: global statistics, B placed in the global column numbering
Vt, Dt, kt = 1500, 4000, 10
per = Dt // world
n_loc = per if rank < world - 1 else Dt - per * (world - 1)
cc = Corpus(Vt, n_loc, kt, 5, doc_base=rank * per)
cnt, rows, offs = cc.A()
hp.upload_counts(Vt, cnt, rows, offs, doc_offset=rank * per, docs_global=Dt)
ti = hp.threshold(kt)
Bd = hp.get_B()
res.update(thr_rows=Bd["rows"], thr_vals=Bd["vals"], thr_offs=Bd["offs"], thr_cols=Bd["original_cols"],
           thr_meta=np.array([hp.doc_offset, hp.D_global, ti["entries_above_threshold"]], np.int64))

# ---- the same with importance sampling of documents (BASELINE configs[3]): keys by global document number, one pivot for the corpus
hp.upload_counts(Vt, cnt, rows, offs, doc_offset=rank * per, docs_global=Dt)
ts = hp.threshold(kt, sample_rate=0.3, sample_seed=11)
Bs = hp.get_B()
res.update(ths_rows=Bs["rows"], ths_vals=Bs["vals"], ths_offs=Bs["offs"], ths_cols=Bs["original_cols"],
           ths_meta=np.array([hp.doc_offset, hp.D_global, ts["docs_kept"]], np.int64))

# ---- hot path on a shard of B
k = 20
B = corpus(3000, 9000, k, 3)
bounds = HotPath.plan_shards(B["offs"], world).astype(np.int64)
d0, d1 = int(bounds[rank]), int(bounds[rank + 1])
o0, o1 = int(B["offs"][d0]), int(B["offs"][d1])
hp.upload_csc(B["V"], B["vals"][o0:o1], B["rows"][o0:o1], B["offs"][d0:d1 + 1] - o0, doc_offset=d0, docs_global=B["D"])
res["range"] = np.array([d0, d1], np.int64)
res["fro"] = np.array([hp.frobenius()], np.float64)
X = np.random.default_rng(0).standard_normal((B["V"], 10)).astype(np.float32)
res["Z"] = hp.gram_apply(X)
X25 = np.random.default_rng(1).standard_normal((B["V"], 25)).astype(np.float32)
res["Z25"] = hp.gram_apply(X25)
r = hp.compute_block_ks(k, seed=4)
res["evals"] = r["evals"]; res["restarts"] = np.array([r["restarts"], r["nconv"]], np.int64)
res["U"] = hp.get_U(k)
upath = os.path.join(os.path.dirname(out), "U1.npy")
if world == 1:
    np.save(upath, res["U"])
U = np.load(upath)
hp.set_U(U)
free = hp.kmeans_init_on_projected_space(k, rng_seed=9)
res["free_seeds"] = free["seeds"]; res["free_res"] = np.array([free["residual"]], np.float64)
spath = os.path.join(os.path.dirname(out), "seeds1.npy")
if world == 1:
    np.save(spath, free["seeds"])
g = hp.kmeans_init_on_projected_space(k, inject_seeds=np.load(spath))
res["kmpp_C"] = g["C_lowd"]; res["kmpp_res"] = np.array([g["residual"]], np.float64); res["min_dist"] = hp.get_min_dist()
lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
res["lp_assign"] = lp["assign"]; res["lp_C"] = lp["C_lowd"]; res["lp_it"] = np.array([lp["iters"]])
hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
ls = hp.run_lloyds(k)
res["ls_assign"] = ls["assign"]; res["ls_cen"] = ls["centers"]; res["ls_it"] = np.array([ls["iters"]])
np.savez(out, **res)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
hp.close()
''' % (ROOT, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, tmp, env=None, tag=""):
    port = _free_port()
    procs = []
    for r in range(world):
        out = os.path.join(tmp, "w%d%s_r%d.npz" % (world, tag, r))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, str(r), str(world), str(port), out],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, **(env or {}))))
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=420)
            outs.append(o)
    finally:
        for p in procs:  # a rank that failed leaves the others waiting in a collective: end exactly those processes
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "world %d rank %d failed:\n%s" % (world, r, (outs[r] if r < len(outs) else "")[-3000:])
    _run.last_outputs = outs
    return [np.load(os.path.join(tmp, "w%d%s_r%d.npz" % (world, tag, r))) for r in range(world)]


@pytest.fixture(scope="module")
def single(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("multirank"))
    return tmp, _run(1, tmp)[0]


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("world,rowshard", [(2, "0"), (2, None), (3, None), (4, None), (2, "split"), (4, "split")])
def test_n_ranks_reproduce_one_rank(single, world, rowshard):
    """rowshard: ISLE_KS_ROWSHARD — unset (the default with several ranks since round 5) has rank r orthogonalise its row slice of the Krylov
    block (all-reduced coefficients, all-gathered block); "0" keeps the orthogonalisation replicated on every rank.  (This test shares one
    GPU through the host-staged transport; over RCCL a collective that never completes ends in ISLE_E_COMM: test_gpu_comm_selftest.py.)"""
    tmp, one = single
    # "split": the opt-in column split of the small EVD's eigenvector stage (ISLE_EVD_SPLIT=1) on top of the defaults
    env = {"ISLE_EVD_SPLIT": "1"} if rowshard == "split" else {"ISLE_KS_ROWSHARD": rowshard} if rowshard else None
    rs = _run(world, tmp, env=env, tag="rs" + str(rowshard))
    _compare_with_one_rank(rs, one, world)


def _compare_with_one_rank(rs, one, world):
    # replicated results: bit-identical on every rank
    for name in ("fro", "Z", "Z25", "evals", "restarts", "U", "free_seeds", "kmpp_C", "lp_C", "lp_it", "ls_cen", "ls_it", "thr_meta"):
        for r in rs[1:]:
            if name == "thr_meta":
                assert np.array_equal(r[name][1:], rs[0][name][1:])
            else:
                assert np.array_equal(r[name], rs[0][name]), "%s differs between ranks" % name
    # the shards tile the documents
    assert rs[0]["range"][0] == 0 and rs[-1]["range"][1] == one["range"][1]
    assert all(rs[i]["range"][1] == rs[i + 1]["range"][0] for i in range(world - 1))
    # thresholding: the shards' B, put side by side, is the single-rank B bit for bit
    assert np.array_equal(np.concatenate([r["thr_rows"] for r in rs]), one["thr_rows"])
    assert np.array_equal(np.concatenate([r["thr_vals"] for r in rs]), one["thr_vals"])
    assert np.array_equal(np.concatenate([r["thr_cols"] for r in rs]), one["thr_cols"])
    assert np.array_equal(np.concatenate([np.diff(r["thr_offs"]) for r in rs]), np.diff(one["thr_offs"]))
    assert rs[0]["thr_meta"][2] == one["thr_meta"][2]
    # ... and with importance sampling: the shards keep exactly the documents the single-rank run keeps
    assert rs[0]["ths_meta"][1] == one["ths_meta"][1] and sum(int(r["ths_meta"][2]) for r in rs) == int(one["ths_meta"][2])
    assert 0 < one["ths_meta"][2] < one["thr_cols"].shape[0]
    assert np.array_equal(np.concatenate([r["ths_cols"] for r in rs]), one["ths_cols"])
    assert np.array_equal(np.concatenate([r["ths_rows"] for r in rs]), one["ths_rows"])
    assert np.array_equal(np.concatenate([r["ths_vals"] for r in rs]), one["ths_vals"])
    assert np.array_equal(np.concatenate([np.diff(r["ths_offs"]) for r in rs]), np.diff(one["ths_offs"]))
    # operator and eigensolver
    a = rs[0]
    assert abs(a["fro"][0] - one["fro"][0]) <= 1e-6 * one["fro"][0]
    assert _rel(a["Z"], one["Z"]) <= 1e-5 and _rel(a["Z25"], one["Z25"]) <= 1e-5
    assert np.max(np.abs(a["evals"] - one["evals"]) / one["evals"]) <= 1e-5
    # k-means with the single-rank U and seeds injected
    assert _rel(a["kmpp_C"], one["kmpp_C"]) <= 1e-5
    assert abs(a["kmpp_res"][0] - one["kmpp_res"][0]) <= 1e-4 * one["kmpp_res"][0]
    md = np.concatenate([r["min_dist"] for r in rs])
    assert np.abs(md - one["min_dist"]).max() <= 1e-4 * one["min_dist"].max()
    assert a["lp_it"][0] == one["lp_it"][0] and a["ls_it"][0] == one["ls_it"][0]
    assert (np.concatenate([r["lp_assign"] for r in rs]) == one["lp_assign"]).mean() >= 0.999
    assert (np.concatenate([r["ls_assign"] for r in rs]) == one["ls_assign"]).mean() >= 0.999
    assert _rel(a["lp_C"], one["lp_C"]) <= 1e-4 and _rel(a["ls_cen"], one["ls_cen"]) <= 1e-4
    # free-running k-means++: the draws walk per-shard D^2 prefix sums joined by the all-gathered totals
    same = (a["free_seeds"] == one["free_seeds"]).mean()
    assert same >= 0.9 or abs(a["free_res"][0] - one["free_res"][0]) <= 0.2 * one["free_res"][0], same


def test_one_rank_bailing_out_of_the_persistent_evd_takes_all_ranks_along(single):
    """The persistent tridiagonalisation falls back to the launch chain when its grid barrier times out — a decision that depends
    on timing, while the two forms round differently.  ISLE_TD_FORCE_BAIL_RANK=1 makes rank 1 (only) report that time-out in
    every small EVD: the flag is all-reduced, so rank 0 follows it to the chain, the replicated Ritz data stay bit-identical,
    the ranks' restart decisions and collective sequences stay in step, and the run completes with the single-rank results."""
    tmp, one = single
    rs = _run(2, tmp, env={"ISLE_TD_FORCE_BAIL_RANK": "1"}, tag="bail")
    for name in ("evals", "restarts", "U", "Z", "lp_C", "ls_cen", "ls_it"):
        assert np.array_equal(rs[1][name], rs[0][name]), "%s differs between ranks" % name
    assert np.max(np.abs(rs[0]["evals"] - one["evals"]) / one["evals"]) <= 1e-5
    assert rs[0]["lp_it"][0] == one["lp_it"][0] and rs[0]["ls_it"][0] == one["ls_it"][0]
    assert (np.concatenate([r["ls_assign"] for r in rs]) == one["ls_assign"]).mean() >= 0.999


def test_comm_selftest_through_the_host_transport(single):
    """ISLE_COMM_SELFTEST=1: behind the communicator's creation every rank runs sum / max / all-gather of every (datatype, size class) the
    step issues on patterns with known results and logs rank, device and PCI bus id (api.cpp comm_selftest; default on over RCCL with
    more than one rank).  Here through the host-staged transport at two ranks: the same call sites, the same checks, and the run behind it
    still reproduces the single-rank one."""
    tmp, one = single
    rs = _run(2, tmp, env={"ISLE_COMM_SELFTEST": "1"}, tag="selftest")
    for r, o in enumerate(_run.last_outputs):
        assert "rank %d of 2" % r in o and "communicator self-test: 72 collectives" in o and "correct in" in o, o[-1500:]
        assert "host-staged test transport" in o
    assert np.array_equal(rs[0]["U"], rs[1]["U"])
    assert np.max(np.abs(rs[0]["evals"] - one["evals"]) / one["evals"]) <= 1e-5


@pytest.mark.parametrize("rowshard", [None, "0"])
def test_two_ranks_over_rccl(single, rowshard):
    """The same comparison with the product's own transport: two ranks, one GPU each, RCCL all-reduce / all-gather (the communicator
    self-test runs first, by default).  Needs two GPUs — the pool's boxes have one, where this test is skipped; until a box with two has
    run it, the row-sharded orthogonalisation and the split EVD (the multi-rank defaults) are verified over the host-staged transport only
    (DESIGN.md section 6 states that)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    tmp, one = single
    env = {"ISLE_TEST_RCCL": "1", "ISLE_COMM_TIMEOUT_S": "60"}
    if rowshard:
        env["ISLE_KS_ROWSHARD"] = rowshard
    rs = _run(2, tmp, env=env, tag="rccl" + str(rowshard))
    _compare_with_one_rank(rs, one, 2)
    assert all("communicator self-test: 72 collectives" in o for o in _run.last_outputs)
