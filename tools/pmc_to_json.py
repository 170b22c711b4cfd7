#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/pmc_probe.py -> one entry of profiles/pmc_traffic.json, keyed by the source of
the Gram-apply kernel it was collected on (bench.py gram_kernel_sha16: the entry is reported as `roofline.traffic` only while the key matches).

usage: pmc_to_json.py <workload> <fetch counter_collection.csv> <write counter_collection.csv> <nnz> [note]
Bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KB and FETCH_SIZE counts half of the bytes on gfx950 (MI355X_MICROARCH.md,
HBM / rocprofv3 section; calibrated in the same pass on sumsq_k, which reads exactly 4 * nnz bytes)."""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def per_dispatch(path, counter):
    acc, disp = defaultdict(float), defaultdict(set)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            name = next((k for k in ("gl_apply_k", "gl_pack_scale_k", "gl_reduce_cm_k", "sumsq_k") if k in r["Kernel_Name"]), None)
            if name is None:
                continue
            acc[name] += float(r["Counter_Value"])
            disp[name].add(r["Dispatch_Id"])
    return {k: (v / max(len(disp[k]), 1), len(disp[k])) for k, v in acc.items()}


wl, fcsv, wcsv, nnz = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
note = sys.argv[5] if len(sys.argv) > 5 else ""
F, W = per_dispatch(fcsv, "FETCH_SIZE"), per_dispatch(wcsv, "WRITE_SIZE")


def nbytes(kernel):
    f = [v for k, v in F.items() if k.startswith(kernel)]
    w = [v for k, v in W.items() if k.startswith(kernel)]
    # several instantiations under one prefix (gl_apply_k<...>): dispatch-weighted mean
    fm = sum(a * n for a, n in f) / max(sum(n for _, n in f), 1)
    wm = sum(a * n for a, n in w) / max(sum(n for _, n in w), 1)
    return (2.0 * fm + wm) * 1024.0, sum(n for _, n in f)


cal, _ = nbytes("sumsq_k")
pack, _ = nbytes("gl_pack_scale_k")
app, napp = nbytes("gl_apply_k")
red, _ = nbytes("gl_reduce_cm_k")
total = pack + 2.0 * app + red
ent = {"gram_apply_hbm_bytes_per_launch": int(round(total)),
       "breakdown_bytes": {"gl_pack_scale_k": int(round(pack)), "gl_apply_k pass 1 + pass 2 (one kernel name: mean of both x 2)": int(round(2 * app)),
                           "gl_reduce_cm_k": int(round(red))},
       "calibration": {"kernel": "sumsq_k", "reads_bytes": 4 * nnz, "counters_say_bytes": int(round(cal)), "ratio": round(cal / (4.0 * nnz), 4)},
       "kernel_source_sha16": bench.gram_kernel_sha16(), "gl_apply_k_dispatches_counted": napp,
       "note": note or "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) over tools/pmc_probe.py %s" % wl}
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
d = json.load(open(path))
d[wl] = ent
json.dump(d, open(path, "w"), indent=2)
print(wl, json.dumps(ent))
