#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then PMC passes (one rocprofv3 run each).
# Usage (inside gpurun): bash tools/profile.sh <tag> [bench args...]
set -u
cd "$(dirname "$0")/.."
REPO=$PWD
TAG=${1:-r01}; shift || true
ARGS=${@:-"--steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
PASS1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM"
PASS2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
PASS3="TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
PASS4="FETCH_SIZE"
PASS5="WRITE_SIZE"
i=1
for P in "$PASS1" "$PASS2" "$PASS3" "$PASS4" "$PASS5"; do
  timeout -s KILL 300 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -- python3 $REPO/bench.py $ARGS > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.log; exit 1; }
  i=$((i+1))
done
python3 $REPO/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
