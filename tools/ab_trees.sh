#!/bin/bash
# Same-box A/B of two whole trees: this one and a second checkout with its own built library (e.g. an earlier
# commit extracted with `git archive <commit> raytracing_amd tools include | tar -x -C ab/<name>` and built there).
# Each tree runs its own tools/scene_bench.py with its own package, alternately, twice round.
# Usage (on the GPU box): bash tools/ab_trees.sh ab/<name> "<scene_bench args>"
OTHER=$1; ARGS=$2
ROOT=$(pwd)
for rep in 1 2; do
  echo "== this tree"
  (cd "$ROOT" && PYTHONPATH=. timeout -k 10 400 python -u tools/scene_bench.py $ARGS | cut -c1-120) || exit 1
  echo "== $OTHER"
  (cd "$ROOT/$OTHER" && PYTHONPATH=. timeout -k 10 400 python -u tools/scene_bench.py $ARGS | cut -c1-120) || exit 1
done
