"""bench.py's output contract: stdout is exactly ONE line, a JSON object with the agreed keys — also when a process group exists
(Gloo reports its connections on file descriptor 1; bench.py sends everything but the result to stderr).  Tiny workload; the
2-rank run shares GPU 0 through the rehearsal transport (ISLE_BENCH_REHEARSE=1)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILIES = ["gram", "ortho", "project", "kmpp", "lloyd_proj", "sparse", "rotate", "lift"]
KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline"]


def _check(stdout, n_gpus, steps, scaling="weak"):
    lines = stdout.splitlines()
    assert len(lines) == 1, stdout[-2000:]
    d = json.loads(lines[0])
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["unit"] == "docs/sec" and d["higher_is_better"] is True
    assert d["scaling"] == scaling and d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # every kernel family with a SURVEY 8(d) figure carries its own roofline: algorithmic bytes / flops x the counts the run executed
    fam = d["roofline_by_family"]
    for f in FAMILIES:
        assert f in fam, (f, sorted(fam))
        v = fam[f]
        assert v["bound"] in ("hbm", "mfma") and v["unit"] == ("GB/s" if v["bound"] == "hbm" else "TFLOP/s")
        assert v["device_ms_per_step"] > 0 and v["achieved"] >= 0 and abs(v["frac"] - v["achieved"] / v["peak"]) < 2e-3
    assert abs(fam["gram"]["frac"] - r["frac"]) < 0.05 + 0.5 * r["frac"]  # two measurements of one kernel (timed region / the extra pass)
    return d


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _check(r.stdout, 1, 2)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "docs/sec"


def test_single_gpu_line_over_a_one_rank_rccl_communicator():
    """bench.py's process maps torch's bundled librccl.so / libamdhip64.so before libisle_hip.so is loaded, so the library's RCCL calls bind
    to THAT build (an N > 1 run of the script does the same).  A forced 1-rank communicator with the communicator self-test runs every
    collective of the step through it on one GPU; the line must equal the communicator-free one."""
    def run(extra):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "1", "--warmup", "1"],
                           capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        return _check(r.stdout, 1, 1), r.stderr
    plain, _ = run({})
    forced, err = run({"ISLE_FORCE_COMM": "1", "ISLE_COMM_SELFTEST": "1"})
    assert "communicator self-test: 72 collectives" in err and "correct in" in err, err[-1500:]
    assert "RCCL " in err
    a, b = plain["accuracy"], forced["accuracy"]
    assert b["gate"]["passed"] and a["gate"]["passed"]
    assert b["planted_topic_agreement"] == a["planted_topic_agreement"]
    assert b["sigma_rel_err_bound"] == pytest.approx(a["sigma_rel_err_bound"], rel=0.5, abs=1e-7)


def test_two_rank_line_through_the_launcher():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, ISLE_BENCH_REHEARSE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "1",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _check(r.stdout, 2, 1)
    assert "REHEARSAL" in d["config"]["parallelism"]


def test_plain_gpus_2_rehearsal():
    """The driver's own form, `python bench.py --gpus 2 ...` with no launcher: the script starts its two ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["ISLE_BENCH_REHEARSE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _check(r.stdout, 2, 1)
    assert "REHEARSAL" in d["config"]["parallelism"]
    assert "launcher: started 2 ranks" in r.stderr


def test_plain_gpus_2_without_two_gpus_fails_loudly():
    """One GPU on the box and no rehearsal flag: no line, a non-zero status and the reason on stderr — never a silent one-rank run."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "ISLE_BENCH_REHEARSE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "needs 2 GPUs" in r.stderr


def test_config4_flow_line():
    """`--workload c4`'s flow at a size a test can run: A on the device, importance sampling there (checked against the CPU port bit for
    bit inside bench.py — a difference ends the run), the hot path on the kept documents, docs/sec quoted on the documents of A."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c4small", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _check(r.stdout, 1, 1)
    st = d["other_stages"]["input"]
    assert all(st["identical_to_cpu_port"].values()) and st["docs_A"] == 200_000 and 20_000 <= st["docs_kept"] <= 20_001
    assert "sample=1 sample_rate=0.10" in d["config"]["workload"]
    assert abs(d["value"] - 200_000 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]  # the metric counts the documents of the input


def test_config5_flow_line():
    """`--workload c5`'s flow: device thresholding (sample = 0), hot path, then catchwords + topic model + edge topics as an other_stages leg
    whose pair selection and edge columns bench.py holds against the CPU restatement."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c5small", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _check(r.stdout, 1, 1)
    e = d["other_stages"]["edge_topics"]
    assert e["pairs_identical_to_oracle"] and 0 < e["num_edge_topics"] <= 300 and e["edge_columns_worst_rel_err"] <= 1e-5
    assert all(d["other_stages"]["input"]["identical_to_cpu_port"].values())


def test_config4_flow_on_two_ranks():
    """The same flow on document shards (two ranks sharing GPU 0 through the rehearsal transport): every rank uploads its share of A, the
    thresholds come from all-reduced statistics, the sampling pivot from the all-gathered keys — the shards keep what the one-rank run keeps."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["ISLE_BENCH_REHEARSE"] = "1"
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c4small", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c4small", "--steps", "1", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert one.returncode == 0 and two.returncode == 0, (one.stderr[-1500:], two.stderr[-3000:])
    d1, d2 = _check(one.stdout, 1, 1), _check(two.stdout, 2, 1, scaling="strong")  # one corpus split over the ranks
    assert "REHEARSAL" in d2["config"]["parallelism"]
    k1 = d1["other_stages"]["input"]["docs_kept"]
    assert "B keeps %d of the 200000 documents" % k1 in d2["config"]["workload"]  # the ranks' kept documents add up to the one-rank count
    assert d2["accuracy"]["gate"]["passed"] and d2["config"]["kmeans"]["nonempty_clusters"] == 100
