#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_wait
mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-44)
  timeout -k 10 280 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc_$tag -o x --output-format csv -- python3 tools/pmc_probe.py c3shard > $OUT/pmc_$tag.log 2>&1
  echo "rc=$? $set"
done
python3 - <<'P'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r03_wait/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for (kn, cn), (n, v) in sorted(agg.items()):
        if "gl_apply" in kn:
            print("  %-62s %-26s calls %4d  per call %.4g" % (kn, cn, n, v / n))
P
