// Mutation fuzz of rt_scene_create_from_memory with the library's HOST code under AddressSanitizer / UBSan
// (needs a GPU: a scene that passes validation is flattened, its trees are built and its tables uploaded).
//   hipcc -O1 -g -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fno-gpu-rdc -Xarch_host -fsanitize=address \
//     -Xarch_host -fsanitize=undefined -shared -o librtmi_asan.so raytracing_amd/csrc/{rt_api,rt_kernels,rt_bvh,rt_multi}.hip \
//     raytracing_amd/csrc/rt_jpeg.cpp raytracing_amd/csrc/rt_png.cpp raytracing_amd/csrc/rt_gifbmp.cpp -ldl -lpthread
//   /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address,undefined tools/fuzz/scene_fuzz.cpp -L. -lrtmi_asan \
//     -Wl,-rpath,'$ORIGIN' -o scene_fuzz
//   ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 ./scene_fuzz <mutations per file> <seed> <scene files written by rt_scene_write>
// Every fourth world that loads also renders 16 x 16 pixels and answers 64 closest-hit queries.
// One to four bytes overwritten per mutation (random values, 0x00, 0xff, 0x7f, 0x80), two in three of them in the first
// 4 KiB, where the header and the index arrays are; every result is handed over from an exact-size heap copy.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <random>
#include <vector>
#include "../../include/rtmi.h"
static std::vector<uint8_t> slurp(const char* p) {
  FILE* f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
  std::vector<uint8_t> v; uint8_t buf[65536]; size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f); return v;
}
int main(int argc, char** argv) {
  const int per_file = atoi(argv[1]);
  std::mt19937_64 rng(atoi(argv[2]));
  rt_ctx* ctx = nullptr;
  if (rt_init(0, &ctx) != RT_OK) { fprintf(stderr, "rt_init: %s\n", rt_last_error()); return 3; }
  long loaded = 0, refused = 0, render_refused = 0;
  for (int a = 3; a < argc; ++a) {
    const std::vector<uint8_t> base = slurp(argv[a]);
    for (int it = 0; it < per_file; ++it) {
      std::vector<uint8_t> d = base;
      if (it > 0) {
        const size_t span = it % 3 == 0 ? d.size() : (d.size() < 4096 ? d.size() : 4096);
        const size_t at = 8 + rng() % (span - 8);
        const int width = 1 + (int)(rng() % 4), kind = (int)(rng() % 5);
        for (int j = 0; j < width && at + j < d.size(); ++j)
          d[at + j] = kind == 0 ? (uint8_t)rng() : kind == 1 ? 0x00 : kind == 2 ? 0xff : kind == 3 ? 0x7f : 0x80;
        if (it % 11 == 0) d.resize(rng() % (d.size() + 1));  // and some are cut short
      }
      void* heap = nullptr;
      if (posix_memalign(&heap, 8, d.size() ? d.size() : 8)) return 4;
      memcpy(heap, d.data(), d.size());
      rt_scene* sc = nullptr;
      const int rc = rt_scene_create_from_memory(ctx, heap, d.size(), &sc);
      if (rc == RT_OK) {
        ++loaded;
        if (loaded % 4 == 1) {  // every fourth world that loads also answers rays and renders a few pixels
          rt_camera cam;
          memset(&cam, 0, sizeof cam);
          double* c = reinterpret_cast<double*>(&cam);  // origin, lower-left corner, horizontal, vertical, u, v, w, lens radius, time0, time1
          const double look[24] = {278, 278, -800, -86, -86, 200, 728, 0, 0, 0, 728, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0.05, 0, 1};
          memcpy(c, look, sizeof look);
          rt_params prm;
          memset(&prm, 0, sizeof prm);
          prm.width = 16; prm.height = 16; prm.spp = 2; prm.max_depth = 6; prm.shard_count = 1; prm.seed = (uint64_t)it;
          prm.background[0] = 0.7; prm.background[1] = 0.8; prm.background[2] = 1.0;
          std::vector<float> sum(16 * 16 * 3);
          std::vector<uint32_t> argb(16 * 16);
          rt_stats st;
          if (rt_render(ctx, sc, &cam, &prm, sum.data(), argb.data(), &st) != RT_OK) ++render_refused;
          std::vector<float> o(64 * 3), dir(64 * 3), tm(64);
          for (auto& x : o) x = (float)((double)(rng() % 2000) * 0.5 - 200.0);
          for (auto& x : dir) x = (float)((double)(rng() % 2001) * 0.001 - 1.0);
          for (auto& x : tm) x = (float)((double)(rng() % 1000) * 0.001);
          std::vector<rt_hit> hits(64);
          if (rt_hit_batch(ctx, sc, 64, o.data(), dir.data(), tm.data(), 0.001f, 1e30f, nullptr, hits.data()) != RT_OK) ++render_refused;
        }
        rt_scene_destroy(sc);
      } else {
        ++refused;
      }
      free(heap);
    }
    printf("%s: loaded %ld, refused %ld, render or query refused %ld so far\n", argv[a], loaded, refused, render_refused);
    fflush(stdout);
  }
  rt_shutdown(ctx);
  fflush(stdout);
  _exit(0);  // (ROCm's sanitizer runtime trips over its own device allocator when the process unwinds normally)
}
