"""bench.py's own launcher (`python bench.py --gpus N` without a launcher environment) — what can be held without a GPU: the script starts N
rank processes itself, and ranks that find fewer than N GPUs end with the reason and a non-zero status instead of a silent one-rank run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_gpus_2_without_gpus_fails_loudly_and_prints_no_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "ISLE_BENCH_REHEARSE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    import torch
    if torch.cuda.device_count() >= 2:
        return  # a box with two GPUs runs the benchmark: covered by tests/test_gpu_bench_contract.py
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "launcher: started 2 ranks" in r.stderr and "needs 2 GPUs" in r.stderr


def test_a_launcher_environment_that_disagrees_with_gpus_is_an_error():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "tiny"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and r.stdout.strip() == "" and "WORLD_SIZE is 1" in r.stderr
