#!/bin/bash
# Builds the sanitizer targets of this directory into gpurun_tmp/ (git-ignored; it travels with a gpurun call):
#   librtmi_asan.so  the library with its HOST code under AddressSanitizer and UBSan (the GPU code is not instrumented)
#   scene_fuzz       mutated scene files against rt_scene_create_from_memory (+ a render and queries for some)
#   api_walk         every entry point of include/rtmi.h with ordinary arguments
#   jpeg_fuzz        mutated JPEG files against rt_image_decode_jpeg (host only: runs without a GPU)
#   png_fuzz         mutated PNG files (chunk bodies edited, CRCs made right again) against rt_image_decode_png (host only)
#   image_fuzz       blindly mutated files of any of the four formats against rt_image_decode (host only; GIF and BMP have no checksums)
# and writes the reference's worlds as scene files beside them. Usage: bash tools/fuzz/build.sh
# Then, on the GPU box (gpurun): cd gpurun_tmp && ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 ./api_walk *.rtscn -- *.jpg *.png *.gif *.bmp
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_tmp
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined"
(cd raytracing_amd/csrc && /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fno-gpu-rdc -w $SAN \
   -shared -o ../../gpurun_tmp/librtmi_asan.so rt_api.hip rt_kernels.hip rt_bvh.hip rt_multi.hip rt_jpeg.cpp rt_png.cpp rt_gifbmp.cpp -ldl -lpthread)
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address,undefined tools/fuzz/scene_fuzz.cpp -Lgpurun_tmp -lrtmi_asan \
   -Wl,-rpath,'$ORIGIN' -o gpurun_tmp/scene_fuzz
/opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -w $SAN tools/fuzz/api_walk.cpp -Lgpurun_tmp -lrtmi_asan \
   -Wl,-rpath,'$ORIGIN' -o gpurun_tmp/api_walk
g++ -O1 -g -std=c++17 -fsanitize=address,undefined tools/fuzz/jpeg_fuzz.cpp raytracing_amd/csrc/rt_jpeg.cpp -o gpurun_tmp/jpeg_fuzz
g++ -O1 -g -std=c++17 -fsanitize=address,undefined tools/fuzz/png_fuzz.cpp raytracing_amd/csrc/rt_png.cpp raytracing_amd/csrc/rt_jpeg.cpp raytracing_amd/csrc/rt_gifbmp.cpp -o gpurun_tmp/png_fuzz
g++ -O1 -g -std=c++17 -fsanitize=address,undefined tools/fuzz/image_fuzz.cpp raytracing_amd/csrc/rt_jpeg.cpp raytracing_amd/csrc/rt_png.cpp raytracing_amd/csrc/rt_gifbmp.cpp -o gpurun_tmp/image_fuzz
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from raytracing_amd.core import setup
from raytracing_amd import scenefile
for name in ("cornell-box", "cornell-smoke", "final-scene", "two-perlin-spheres", "random-scene", "earth"):
    cfg, flat = setup(name, 2)
    open(f"gpurun_tmp/{name}.rtscn", "wb").write(scenefile.dumps(flat))
PY
cp raytracing_amd/assets/earthmap.jpg raytracing_amd/assets/stones.jpg raytracing_amd/assets/map2.png gpurun_tmp/
python3 - <<'PY'
import numpy as np
from PIL import Image
with np.load("raytracing_amd/assets/stones_rgb8.npz") as z:
    rgb = z["rgb8"]
Image.fromarray(rgb).save("gpurun_tmp/stones_rgb.png")
Image.fromarray(rgb).quantize(200).save("gpurun_tmp/stones_palette.png")
Image.fromarray(rgb).convert("LA").save("gpurun_tmp/stones_grey_alpha.png")
Image.fromarray(rgb).quantize(256).save("gpurun_tmp/stones.gif")
Image.fromarray(rgb).save("gpurun_tmp/stones.bmp")
PY
echo built gpurun_tmp/
