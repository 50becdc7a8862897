#!/bin/bash
# Same-box A/B: runs tools/scene_bench.py once per library named (ab/librtmi_<tag>.so), twice round.
# Usage (on the GPU box): bash tools/ab_run.sh "<scene_bench args>" tag1 tag2 ...
ARGS=$1; shift
export PYTHONPATH=.
for rep in 1 2; do
  for t in "$@"; do
    echo "== $t"
    RTMI_LIB=ab/librtmi_$t.so timeout -k 10 300 python -u tools/scene_bench.py $ARGS | cut -c1-120 || exit 1
  done
done
