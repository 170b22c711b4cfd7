"""ISLETrain CLI contract and host pre-stages (tdf reader, CSC build, thresholding) — CPU only."""
import os
import subprocess

import numpy as np

from tools.synth import Corpus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "isle_amd", "host")


def write_tdf(path, counts, rows, offs, shuffle_seed=None, style="plain"):
    doc = np.repeat(np.arange(len(offs) - 1), np.diff(offs))
    order = np.arange(len(doc))
    if shuffle_seed is not None:
        order = np.random.default_rng(shuffle_seed).permutation(len(doc))
    with open(path, "w", newline="") as f:
        for n, i in enumerate(order):
            sep = [" ", "\t", "  ", " \t "][n % 4] if style == "messy" else " "
            eol = "\r\n" if (style == "messy" and n % 3 == 0) else "\n"
            if n == len(order) - 1 and style == "messy":
                eol = ""  # last newline optional (README.md:41-45)
            f.write("%d%s%d%s%d%s" % (doc[i] + 1, sep, rows[i] + 1, sep, int(counts[i]), eol))
    return len(doc)


def read_dump(path):
    raw = open(path, "rb").read()
    V, D, nnz, above = np.frombuffer(raw, np.uint64, 4)
    off = 32
    vals = np.frombuffer(raw, np.float32, int(nnz), off); off += 4 * int(nnz)
    rows = np.frombuffer(raw, np.uint64, int(nnz), off); off += 8 * int(nnz)
    offs = np.frombuffer(raw, np.int64, int(D) + 1, off); off += 8 * (int(D) + 1)
    oc = np.frombuffer(raw, np.uint64, int(D), off); off += 8 * int(D)
    zetas = np.frombuffer(raw, np.float32, int(V), off)
    return dict(V=int(V), D=int(D), nnz=int(nnz), above=int(above), vals=vals, rows=rows, offs=offs, original_cols=oc, zetas=zetas)


def test_usage_message_and_exit_code():
    r = subprocess.run([os.path.join(HOST, "ISLETrain"), "only", "three", "args"], capture_output=True, text=True)
    assert r.returncode == 255  # exit(-1), drivers/ISLETrain.cpp:9-16
    assert "Incorrect usage of ISLETrain" in r.stdout and "<max_edge_topics>" in r.stdout


def test_isleinfer_usage_message_and_exit_code():
    # drivers/ISLEInfer.cpp:11-20: anything but 11 arguments prints the usage text and exits with -1
    exe = os.path.join(ROOT, "isle_amd", "host", "ISLEInfer")
    r = subprocess.run([exe, "a", "b"], capture_output=True, text=True)
    assert r.returncode == 255
    assert "Incorrect usage of ISLEInfer" in r.stdout


def test_prestage_matches_input_tool(tmp_path):
    V, D, k = 400, 1500, 6
    c = Corpus(V, D, k, seed=8, L0=50.0)
    counts, rows, offs = c.A()
    ref = c.threshold(k)
    for style, shuffle in (("plain", None), ("messy", 3)):
        tdf = str(tmp_path / ("c_%s.tdf" % style))
        n = write_tdf(tdf, counts, rows, offs, shuffle_seed=shuffle, style=style)  # unsorted lines: the trainer sorts (trainer.cpp:237)
        out = str(tmp_path / "B.bin")
        r = subprocess.run([os.path.join(HOST, "prestage_dump"), tdf, str(V), str(D), str(n), str(k), "0", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        B = read_dump(out)
        assert B["D"] == ref["D"] and B["nnz"] == ref["nnz"] and B["above"] == ref["nnz"]
        assert np.array_equal(B["zetas"], ref["zetas"])
        assert np.array_equal(B["rows"].astype(np.uint32), ref["rows"]) and np.array_equal(B["offs"], ref["offs"])
        assert np.array_equal(B["vals"], ref["vals"]) and np.array_equal(B["original_cols"], ref["original_cols"])


def test_tdf_entry_count_must_match(tmp_path):
    tdf = str(tmp_path / "t.tdf")
    open(tdf, "w").write("1 1 2\n1 2 1\n2 1 3\n")
    r = subprocess.run([os.path.join(HOST, "prestage_dump"), tdf, "5", "2", "4", "1", "0", str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "max_entries" in r.stderr  # include/utils.h:227


def test_isletrain_refuses_numbers_its_types_cannot_hold():
    # every count is checked against the type the trainer takes it as, before anything touches a file or the device
    exe = os.path.join(HOST, "ISLETrain")
    base = ["none.tdf", "none.vocab", "/tmp", "100", "10", "50", "3", "0", "0", "0.1", "0", "5"]
    for pos, bad, what in ((3, "5000000000", "<vocab_size>"), (11, "3000000000", "<max_edge_topics>"), (5, "18446744073709551615", "<max_entries>"),
                           (6, "-3", "negative"), (4, "12x", "not a number")):
        a = list(base)
        a[pos] = bad
        r = subprocess.run([exe] + a, capture_output=True, text=True)
        assert r.returncode == 1 and what in r.stderr, (bad, r.stderr)
