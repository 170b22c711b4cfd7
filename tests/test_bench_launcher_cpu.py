"""bench.py's own launcher (`python bench.py --gpus N` without a launcher environment) — what can be held without a GPU: the script starts N
rank processes itself, ranks that find fewer than N GPUs end with the reason and a non-zero status instead of a silent one-rank run, and no
rank outlives a launcher that is told to stop."""
import os
import re
import signal
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCHER_ENV_KEYS = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "ISLE_BENCH_REHEARSE")


def _gpus():
    import torch
    return torch.cuda.device_count()  # counts, does not initialise a device


def test_plain_gpus_2_without_gpus_fails_loudly_and_prints_no_line():
    if _gpus() >= 2:
        pytest.skip("a box with two GPUs would run the benchmark: covered by tests/test_gpu_bench_contract.py")
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV_KEYS}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "launcher: started 2 ranks" in r.stderr and "needs 2 GPUs" in r.stderr


def test_a_launcher_environment_that_disagrees_with_gpus_is_an_error():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "tiny"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and r.stdout.strip() == "" and "WORLD_SIZE is 1" in r.stderr


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    try:  # a zombie still answers signal 0: it is gone for our purpose (the launcher waits for its children, so this should not occur)
        with open("/proc/%d/stat" % pid) as f:
            return f.read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def test_sigterm_to_the_launcher_leaves_no_rank_behind():
    """A signal that reaches only the launcher PID (a driver's timeout): every rank it started is ended by PID and the status says so.
    ISLE_BENCH_HOLD_S makes the ranks wait before they look for GPUs, so the signal finds them alive on any box."""
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV_KEYS}
    env["ISLE_BENCH_HOLD_S"] = "120"
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "1", "--warmup", "1"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)
    line, t0 = "", time.time()
    while "launcher: started 2 ranks" not in line and time.time() - t0 < 60:
        line = p.stderr.readline()
    m = re.search(r"pids \[(\d+), (\d+)\]", line)
    assert m, line
    pids = [int(m.group(1)), int(m.group(2))]
    assert all(_alive(q) for q in pids)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err[-500:])
    assert out.strip() == "" and "ending all ranks" in err
    assert not any(_alive(q) for q in pids)


def test_overall_deadline_ends_ranks_that_never_exit():
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV_KEYS}
    env.update(ISLE_BENCH_HOLD_S="120", ISLE_BENCH_DEADLINE_S="3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode == 124 and r.stdout.strip() == "" and "overall deadline" in r.stderr
    m = re.search(r"pids \[(\d+), (\d+)\]", r.stderr)
    assert m and not any(_alive(int(q)) for q in m.groups())
