#!/usr/bin/env python3
"""Assembles DESIGN.md from docs/design_src/*.md (prose, one file per section) and two generated tables: section 4's kernel table
(tools/make_kernel_table.py over the round's rocprofv3 summaries) and section 8's switch table (the library's own table: isle_amd/csrc/api.cpp).
usage: make_design.py <c3full kernel_stats.csv> <steps> <c3shard kernel_stats.csv> <steps>"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "docs", "design_src")


def switches():
    txt = open(os.path.join(ROOT, "isle_amd", "csrc", "api.cpp")).read()
    body = txt[txt.index("const IsleKnobInfo isle_knob_table[KN_COUNT] = {"):]
    body = body[:body.index("};")]
    rows = re.findall(r'\{"(ISLE_[A-Z0-9_]+)",\s*"([a-z ]+)",\s*"((?:[^"\\]|\\.)*)"\}', body)
    out = ["| switch | kind | effect |", "|---|---|---|"]
    for name, kind, what in rows:
        what = what.replace('\\"', '"').replace("|", "\\|")
        if len(what) > 100:  # the full text: isle_hip_switch_info / api.cpp
            cut = what[:100]
            what = cut[:cut.rfind(" ")] + " …"
        out.append("| `%s` | %s | %s |" % (name, kind, what))
    return "\n".join(out)


def main():
    a, sa, b, sb = sys.argv[1:5]
    ktab = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_kernel_table.py"), a, sa, b, sb, "40"], capture_output=True, text=True, check=True).stdout
    parts = []
    for fn in sorted(os.listdir(SRC)):
        if not fn.endswith(".md"):
            continue
        t = open(os.path.join(SRC, fn)).read()
        t = t.replace("<<KERNEL_TABLE>>", ktab.strip()).replace("<<SWITCH_TABLE>>", switches())
        parts.append(t.rstrip() + "\n")
    doc = "\n".join(parts)
    open(os.path.join(ROOT, "DESIGN.md"), "w").write(doc)
    print("DESIGN.md: %d bytes" % len(doc.encode()))


if __name__ == "__main__":
    main()
