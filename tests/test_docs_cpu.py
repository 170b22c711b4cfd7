"""DESIGN.md is a generated file (tools/make_design.py: docs/design_src/ + two tables).  Held here: it stays a document a reader can get
through (<= 60 KB), its kernel table has a row for every kernel among the top 40 of the round's config-3 profile, and it is what the generator
produces from the committed sources (nobody edits the output by hand)."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C3 = os.path.join(ROOT, "profiles", "r06_v_c3full_kernel_stats.csv")
SHARD = os.path.join(ROOT, "profiles", "r06_v_c3shard_kernel_stats.csv")


def test_design_is_short_and_covers_the_profiles_top_40():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_kernel_table as mk  # noqa: F401  (import runs nothing: the module prints only as a script)
    design = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read()
    assert len(design.encode()) <= 60 * 1024
    sec = design[design.index("## 4. Kernels"):design.index("## 5. Host control flow")]
    rows = list(csv.DictReader(open(C3)))
    agg = {}
    for r in rows:
        k = mk.short(r["Name"])
        agg[k] = agg.get(k, 0.0) + float(r["TotalDurationNs"])
    for k in sorted(agg, key=lambda x: -agg[x])[:40]:
        name = k.replace("isle_gemm3::", "").replace("isle_gemm::", "")[:70]
        assert "`%s`" % name in sec, name
        assert mk.info(k)[0] != "?", k  # every one of them has a family and a description, not a blank


def test_design_is_the_generators_output(tmp_path):
    cur = open(os.path.join(ROOT, "DESIGN.md"), "rb").read()
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_design.py"), C3, "5", SHARD, "5"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-1000:]
        assert open(os.path.join(ROOT, "DESIGN.md"), "rb").read() == cur
    finally:
        open(os.path.join(ROOT, "DESIGN.md"), "wb").write(cur)
