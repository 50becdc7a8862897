#!/bin/bash
# Memory-pipe counters of the path kernel for one scene (texture addresser / L1 / LDS): is a BVH walk held by
# issue, by the vector L1's address rate, or by latency?  Usage (inside gpurun): bash tools/pmc_mem.sh <tag> <scene>
set -u
cd "$(dirname "$0")/.."
REPO=$PWD
TAG=$1; SCENE=$2
OUT=$REPO/gpurun_out/pmcmem_$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$REPO
cd /tmp
i=1
for P in "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum GRBM_GUI_ACTIVE" \
         "SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $REPO/tools/scene_bench.py $SCENE > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
  i=$((i+1))
done
python3 - $OUT $SCENE <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, scene = sys.argv[1], sys.argv[2]
s = defaultdict(float); n = defaultdict(set)
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_render" in row["Kernel_Name"]:
            s[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]].add(row["Dispatch_Id"])
line = [l for l in open(os.path.join(out, "p1.log")) if l.startswith(scene)]
print(line[-1].strip() if line else scene)
for k in sorted(s):
    print(f"  {k:40s} per launch {s[k] / max(len(n[k]), 1):.6g}")
PY
