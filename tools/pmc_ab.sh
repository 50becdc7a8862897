#!/bin/bash
# Instruction-count PMC pass of tools/scene_bench.py for several libraries (ab/librtmi_<tag>.so), same box.
# Usage (inside gpurun): bash tools/pmc_ab.sh "<scene_bench args>" tag1 tag2 ...
set -u
cd "$(dirname "$0")/.."
REPO=$PWD
ARGS=$1; shift
export TMPDIR=/tmp PYTHONPATH=$REPO
cd /tmp
for t in "$@"; do
  OUT=$REPO/gpurun_out/pmcab_$t
  rm -rf $OUT; mkdir -p $OUT
  export RTMI_LIB=$REPO/ab/librtmi_$t.so
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $OUT/p1 -- python3 $REPO/tools/scene_bench.py $ARGS > $OUT/p1.log 2>&1 || { echo "pass failed"; tail -5 $OUT/p1.log; exit 1; }
  python3 - $OUT $t <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
s = defaultdict(float); n = defaultdict(set)
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_render" in row["Kernel_Name"]:
            s[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]].add(row["Dispatch_Id"])
per = {k: s[k] / max(len(n[k]), 1) for k in s}
print(tag, " ".join(f"{k[3:]}={per[k]:.4g}" for k in sorted(per)), f"lane_ratio={per['SQ_THREAD_CYCLES_VALU'] / (per['SQ_ACTIVE_INST_VALU'] * 64):.3f}")
PY
  grep -h "kernel" $OUT/p1.log | cut -c1-110
done
