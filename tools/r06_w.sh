#!/bin/bash
# orthogonalisation kernels at a rank's row slice (n = 12 500): kernel durations by basis width from a trace of tools/ortho_slice_probe.py
set -o pipefail
O=gpurun_out/r06_w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/p -o s -- python3 tools/ortho_slice_probe.py ${1:-12500} > $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
grep "rep" $O/probe.log
f=$(find $O/p -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/by_width.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
for name in ("vtf_mfma_k","update_mfma_k","vtf_reduce_k","pqr_","gemm_f32"):
    sel=[r for r in rows if name in r["Kernel_Name"]]
    sel=sel[len(sel)//2:]
    if not sel: continue
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in sel]
    print("%-16s calls %5d  total %8.2f ms  avg %6.1f us  min %5.1f  max %6.1f   grids %s .. %s" % (name, len(sel), sum(d)/1e3, sum(d)/len(d), min(d), max(d),
          (int(sel[0]["Grid_Size_X"])//int(sel[0]["Workgroup_Size_X"]), sel[0]["Grid_Size_Y"], sel[0]["Workgroup_Size_X"]), (int(sel[-1]["Grid_Size_X"])//int(sel[-1]["Workgroup_Size_X"]), sel[-1]["Grid_Size_Y"])))
    n=len(d)
    print("      by decile of the solve: " + " ".join("%.1f" % (sum(d[i*n//10:(i+1)*n//10])/max(1,len(d[i*n//10:(i+1)*n//10]))) for i in range(10)))
# gaps between consecutive kernels of the second solve
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
half=rows[len(rows)//2:]
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in half)/1e6
span=(int(half[-1]["End_Timestamp"])-int(half[0]["Start_Timestamp"]))/1e6
print("second half of the trace: %d kernels, busy %.1f ms of %.1f ms" % (len(half), busy, span))
PY
find $O -name "*kernel_trace.csv" -delete
