#!/bin/bash
# orthogonalisation kernels by basis width: durations from a kernel trace of one C3-shard step, binned by grid size
set -o pipefail
TAG=${1:-default}; O=gpurun_out/r06_t/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/p -o s -- python3 bench.py --workload c3shard --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
f=$(find $O/p -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/ortho_by_width.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
V=100000
for name in ("vtf_mfma_k","update_mfma_k","vtf_reduce_k"):
    sel=[r for r in rows if name in r["Kernel_Name"]]
    sel=sel[len(sel)//2:]   # the timed step (second half)
    bins=collections.OrderedDict()
    for i,r in enumerate(sel):
        dur=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
        gx,gy,wx=int(r["Grid_Size_X"]),int(r["Grid_Size_Y"]),int(r["Workgroup_Size_X"])
        key=(gx//wx, gy)
        bins.setdefault(key,[]).append(dur)
    print(name, "calls", len(sel))
    ks=sorted(bins.keys(), key=lambda k:(k[1],k[0]))
    tot=sum(sum(v) for v in bins.values()); print('   total %.2f ms per step' % (tot/1e3))
    for k in ks[::max(1,len(ks)//8)]:
        v=bins[k]; print("   grid %s: n %d  avg %.1f us  min %.1f" % (k, len(v), sum(v)/len(v), min(v)))
PY
find $O -name "*kernel_trace.csv" -delete
