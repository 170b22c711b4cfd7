#!/usr/bin/env python3
"""Build-time check of a code-generation property the merged-stream forms of gl_apply_k (gram_lds.hip) rely on.

Their id ring is four 64-bit VGPR pairs loaded by inline-asm global_load_dwordx2 with hand-placed s_waitcnt vmcnt(3): the
compiler does not know that those registers are written asynchronously, so it must never touch them outside the asm blocks
(a copy, a spill or a phi move would read a register whose load is still in flight).  This script compiles gram_lds.hip to
gfx950 assembly and verifies, for every merged instantiation, that after the first ring load no compiler-generated
instruction names a ring register.  usage: check_id_ring.py  (exit 0 = property holds)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check(asm_text):
    names = re.findall(r'^(_ZN\S*gl_apply_kILi(\d)ELb(\d)ELi([12])E\S*):', asm_text, re.M)
    report, bad = [], 0
    for full, lpe, half, m in names:
        i = asm_text.index('\n' + full + ':')
        lines = asm_text[i:asm_text.index('s_endpgm', i)].split('\n')
        ring, first = [], None
        for n, ln in enumerate(lines):
            if ';;#ASMSTART' in ln:
                mm = re.match(r'\s+global_load_dwordx2 v\[(\d+):(\d+)\]', lines[n + 1])
                if mm:
                    ring.append((int(mm.group(1)), int(mm.group(2))))
                    if first is None:
                        first = n
        regs = set()
        for a, b in set(ring):
            regs |= {a, b}
        issues, inasm = [], False
        for n, ln in enumerate(lines):
            if ';;#ASMSTART' in ln:
                inasm = True
                continue
            if ';;#ASMEND' in ln:
                inasm = False
                continue
            if inasm or first is None or n < first or ln.strip().startswith(';') or not ln.startswith('\t'):
                continue
            used = set()
            for a, b, c in re.findall(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', ln):
                used |= {int(c)} if c else set(range(int(a), int(b) + 1))
            if used & regs:
                issues.append((n, ln.strip()))
        report.append((int(lpe), int(half), int(m), sorted(set(ring)), issues))
        bad += len(issues) + (0 if len(set(ring)) == 4 else 1)
    return report, bad


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "gl.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-S",
                               "--cuda-device-only", "-w", "-o", out, os.path.join(ROOT, "isle_amd", "csrc", "gram_lds.hip")])
        report, bad = check(open(out).read())
    for lpe, half, m, ring, issues in report:
        print("gl_apply_k<%d,%d,%d> ring %s: %d foreign uses" % (lpe, half, m, ring, len(issues)))
        for n, l in issues[:8]:
            print("    line %d: %s" % (n, l))
    if not report:
        print("no instantiation of gl_apply_k found")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
