#!/bin/bash
# Builds ab/librtmi_<tag>.so with extra compiler flags, for same-box A/B runs (RTMI_LIB=ab/librtmi_<tag>.so).
# Usage: bash tools/ab_build.sh <tag> [-DNAME=VALUE ...]
set -e
cd "$(dirname "$0")/../raytracing_amd/csrc"
TAG=$1; shift
mkdir -p ../../ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -Wno-unused-result -fno-gpu-rdc \
  "$@" -shared -o ../../ab/librtmi_$TAG.so rt_api.hip rt_kernels.hip rt_bvh.hip rt_multi.hip rt_jpeg.cpp rt_png.cpp rt_gifbmp.cpp -ldl -lpthread
echo built ab/librtmi_$TAG.so
