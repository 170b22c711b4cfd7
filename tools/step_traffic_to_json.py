#!/usr/bin/env python3
"""profiles/r06_steppmc/{FETCH,WRITE}_SIZE_by_kernel.csv (tools/r06_steppmc.sh: counters over one config-3 step, one row per kernel) ->
profiles/r06_step_traffic_by_family.json: HBM bytes of a whole step by kernel family, keyed by the library's source hash (bench.py reports it as
`step_traffic` while the key matches).  Bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE counts half the bytes on gfx950)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import make_kernel_table as mk  # noqa: E402

D = os.path.join(ROOT, "profiles", "r06_steppmc")
F = {r["kernel"]: (int(r["dispatches"]), float(r["FETCH_SIZE_KB_sum"])) for r in csv.DictReader(open(os.path.join(D, "FETCH_SIZE_by_kernel.csv")))}
W = {r["kernel"]: (int(r["dispatches"]), float(r["WRITE_SIZE_KB_sum"])) for r in csv.DictReader(open(os.path.join(D, "WRITE_SIZE_by_kernel.csv")))}
FAMILY = {"gram / project / kmpp": "gl_apply", "project / kmpp / sparse": "gl_apply", "kmpp / movers": "gl_apply", "project": "project", "project / kmpp": "project",
          "ortho": "ortho", "ortho / qr": "ortho", "qr": "qr", "evd": "evd", "rotate / lift": "rotate+lift", "dense": "dense", "lloyd_proj": "lloyd_proj",
          "lloyd_proj / sparse": "lloyd_proj", "sparse": "sparse", "sparse_update": "sparse", "kmpp": "kmpp", "kmeans": "kmpp", "op_build": "op_build",
          "gram": "gram", "upload": "upload", "frobenius": "other", "sort": "other", "runtime": "runtime", "?": "other"}


def fam(k):
    if k.startswith("isle_gemm::gemm_f32_k<isle_gemm::Cfg<2, 1"):
        return "kmpp"
    f = mk.info(k)[0]
    return FAMILY.get(f, f)


tot, per = {}, {}
for k in set(F) | set(W):
    b = (2 * F.get(k, (0, 0))[1] + W.get(k, (0, 0))[1]) * 1024
    f = fam(k)
    tot[f] = tot.get(f, 0) + b
    per.setdefault(f, []).append((b, k, F.get(k, W.get(k))[0]))
pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["c3full"]
gram_apply = 300 * pmc["breakdown_bytes"]["gl_apply_k pass 1 + pass 2 (one kernel name: mean of both x 2)"]
gl = tot.pop("gl_apply")
tot["gram"] = tot.get("gram", 0) + gram_apply
tot["wide_thin_products (projection + k-means++ + movers: gl_apply_k panels)"] = gl - gram_apply
out = {"workload": "config 3 on one GPU, one step (tools/pmc_step.py under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; tools/r06_steppmc.sh)",
       "bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024": True,
       "library_sources_sha16": bench.library_sources_sha16(),
       "hbm_bytes_per_step_by_family": {k: int(v) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])},
       "total_bytes_per_step": int(sum(tot.values())),
       "largest_kernels_per_family": {f: [{"kernel": k, "dispatches": n, "GB": round(b / 1e9, 2)} for b, k, n in sorted(v, reverse=True)[:4]] for f, v in per.items()},
       "note": "the Gram applications' share of gl_apply_k (600 of its dispatches) is taken from the per-application pass (profiles/pmc_traffic.json c3full); the "
               "remaining dispatches of the same kernel are the 100 projection panels, the 128 thin products of the k-means++ rounds and the movers"}
json.dump(out, open(os.path.join(ROOT, "profiles", "r06_step_traffic_by_family.json"), "w"), indent=1)
for k, v in out["hbm_bytes_per_step_by_family"].items():
    print("%-80s %8.1f GB" % (k, v / 1e9))
print("total %.2f TB per step; sources %s" % (out["total_bytes_per_step"] / 1e12, out["library_sources_sha16"]))
