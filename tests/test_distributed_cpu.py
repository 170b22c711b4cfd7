"""world_size-2 gloo tests (CPU) of the N > 1 path's host side: column sharding + globally consistent
thresholding, and the sharded form of the Gram apply  Z = sum_g B_g (B_g^T X)  that the RCCL all-reduce in
libisle_hip.so implements (checked here with the oracle as the per-shard operator and gloo as the all-reduce)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch
    import torch.distributed as dist
    from tools.synth import Corpus
    from oracle.oracle import OracleCsc
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    V, D, k, seed = 800, 3000, 8, 21
    per = D // world

    def allreduce(a):
        dist.all_reduce(torch.from_numpy(a))
        return a

    B = Corpus(V, per, k, seed, doc_base=rank * per).threshold(k, allreduce=allreduce)
    m = OracleCsc(V, B["D"], B["vals"], B["rows"], B["offs"])
    X = np.random.default_rng(5).standard_normal((V, 10)).astype(np.float32)
    Z = np.ascontiguousarray(m.gram_apply(X))
    allreduce(Z)
    fro = np.array([m.frobenius()], np.float64)
    allreduce(fro)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), Z=Z, zetas=B["zetas"], D=B["D"], nnz=B["nnz"], fro=fro,
             original_cols=B["original_cols"] + rank * per)
    dist.barrier()
    dist.destroy_process_group()


def test_two_shards_equal_one(tmp_path):
    import torch.multiprocessing as mp
    from tools.synth import Corpus
    from oracle.oracle import OracleCsc
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    V, D, k, seed = 800, 3000, 8, 21
    B = Corpus(V, D, k, seed).threshold(k)
    # thresholds use GLOBAL statistics, so the sharded corpus equals the monolithic one column for column
    assert np.array_equal(r0["zetas"], B["zetas"]) and np.array_equal(r1["zetas"], B["zetas"])
    assert int(r0["D"]) + int(r1["D"]) == B["D"] and int(r0["nnz"]) + int(r1["nnz"]) == B["nnz"]
    assert np.array_equal(np.concatenate([r0["original_cols"], r1["original_cols"]]), B["original_cols"])
    m = OracleCsc(V, B["D"], B["vals"], B["rows"], B["offs"])
    X = np.random.default_rng(5).standard_normal((V, 10)).astype(np.float32)
    Z = m.gram_apply(X)
    # both ranks hold the same all-reduced Z, equal to the unsharded product up to fp32 re-association
    assert np.array_equal(r0["Z"], r1["Z"])
    assert np.linalg.norm(r0["Z"] - Z) <= 1e-5 * np.linalg.norm(Z)
    assert abs(float(r0["fro"][0]) - m.frobenius()) <= 1e-5 * m.frobenius()
