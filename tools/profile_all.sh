#!/bin/bash
# One GPU call: headline profile (tools/profile.sh) and the PMC summary of every other world.
# Usage (inside gpurun): bash tools/profile_all.sh <tag>
set -u
cd "$(dirname "$0")/.."
TAG=$1
bash tools/profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1 || { tail -5 gpurun_out/profile_$TAG.log; exit 1; }
for S in cornell-smoke random-scene two-spheres two-perlin-spheres earth simple-light final-scene; do
  bash tools/pmc_scene.sh ${TAG}_$S $S > gpurun_out/pmc_${TAG}_$S.txt 2>&1 || { tail -5 gpurun_out/pmc_${TAG}_$S.txt; exit 1; }
  echo "done $S"
done
