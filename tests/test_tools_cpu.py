"""CPU tests of the measurement helpers under tools/ (no GPU, no reference)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gap_report_attributes_idle_time_to_kernel_pairs(tmp_path):
    """tools/gap_report.py: idle time between consecutive kernels of a rocprofv3 kernel trace, by (previous -> following) kernel,
    inside a time window — the tool that found the host-side stalls of round 2."""
    rows = [("a_k(int)", 0, 10_000), ("b_k(float*)", 30_000, 40_000), ("a_k(int)", 41_000, 50_000), ("b_k(float*)", 70_000, 80_000),
            ("c_k()", 400_000_000, 400_010_000)]  # the last gap (0.4 s) is a phase boundary and must be ignored
    p = tmp_path / "trace.csv"
    with open(p, "w") as f:
        f.write('"Kind","Kernel_Name","Start_Timestamp","End_Timestamp"\n')
        for name, s, e in rows:
            f.write('"KERNEL_DISPATCH","%s",%d,%d\n' % (name, s, e))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gap_report.py"), str(p), "4"], capture_output=True, text=True, check=True).stdout
    lines = out.splitlines()
    assert lines[0].startswith("kernels 5, busy 0.0 ms, idle (gaps < 0.2 s) 0.0 ms") or "kernels 5" in lines[0]
    pair = [ln for ln in lines if "a_k -> b_k" in ln]
    assert len(pair) == 1 and " 2 x" in pair[0] and "20.0 us" in pair[0]  # two gaps of 20 us each
    assert not any("b_k -> a_k" in ln for ln in lines)  # 1 us: under the 4-us threshold
    assert not any("c_k" in ln for ln in lines[1:])
    # window: only the first pair of kernels
    out2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gap_report.py"), str(p), "4", "0", "0.045"], capture_output=True, text=True,
                          check=True).stdout
    assert "kernels 3" in out2.splitlines()[0] and " 1 x" in [ln for ln in out2.splitlines() if "a_k -> b_k" in ln][0]
