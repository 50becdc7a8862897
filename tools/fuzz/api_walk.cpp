// Every entry point of include/rtmi.h called once or more with ordinary arguments, the library's HOST code under
// AddressSanitizer / UBSan (the GPU code is not instrumented): what the Python tests cannot do, because a sanitized
// library wants its runtime loaded before the interpreter. Needs a GPU.
//   hipcc -O1 -g -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fno-gpu-rdc -Xarch_host -fsanitize=address \
//     -Xarch_host -fsanitize=undefined -shared -o librtmi_asan.so raytracing_amd/csrc/{rt_api,rt_kernels,rt_bvh,rt_multi}.hip \
//     raytracing_amd/csrc/rt_jpeg.cpp raytracing_amd/csrc/rt_png.cpp raytracing_amd/csrc/rt_gifbmp.cpp -ldl -lpthread
//   hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined \
//     tools/fuzz/api_walk.cpp -L. -lrtmi_asan -Wl,-rpath,'$ORIGIN' -o api_walk
//   ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 ./api_walk <scene files written by rt_scene_write> -- <JPEG, PNG, GIF and BMP files>
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../include/rtmi.h"
extern "C" int __lsan_do_recoverable_leak_check();

#define OK(call)                                                                                   \
  do {                                                                                             \
    const int rc_ = (call);                                                                        \
    if (rc_ != RT_OK) { fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, rt_last_error()); exit(1); } \
  } while (0)
#define MOK(call)                                                                                  \
  do {                                                                                             \
    const int rc_ = (call);                                                                        \
    if (rc_ != RT_OK) { fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, rt_multi_last_error()); exit(1); } \
  } while (0)
#define HIP(call)                                                                                  \
  do {                                                                                             \
    const hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); exit(1); } \
  } while (0)

static std::vector<uint8_t> slurp(const char* p) {
  FILE* f = fopen(p, "rb");
  if (!f) { perror(p); exit(2); }
  std::vector<uint8_t> v;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}

static rt_camera a_camera() {  // from [278 278 -800] towards +z, a small lens, the shutter open over [0, 1]
  rt_camera cam;
  const double look[24] = {278, 278, -800, -86, -86, 200, 728, 0, 0, 0, 728, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0.05, 0, 1};
  memcpy(&cam, look, sizeof look);
  return cam;
}
static rt_params some_params(int w, int h, int spp) {
  rt_params p;
  memset(&p, 0, sizeof p);
  p.width = w; p.height = h; p.spp = spp; p.max_depth = 8; p.shard_count = 1; p.seed = 0x5EED0001u;
  p.background[0] = 0.7; p.background[1] = 0.8; p.background[2] = 1.0;
  return p;
}

static void walk_scene(rt_ctx* ctx, rt_scene* sc, std::mt19937_64& rng) {
  rt_scene_info info;
  OK(rt_scene_get_info(sc, &info));
  int32_t variant = -1, height = 0, n_nodes = 0;
  uint32_t mask = 0;
  uint64_t digest = 0;
  OK(rt_scene_get_variants(sc, &variant, &mask));
  OK(rt_scene_bvh_digest(sc, &digest, &n_nodes, &height));
  const rt_camera cam = a_camera();
  const int W = 44, H = 27;  // ragged against the 8 x 8 tiles
  std::vector<float> sum((size_t)W * H * 3);
  std::vector<uint32_t> argb((size_t)W * H);
  rt_stats st;
  rt_params p = some_params(W, H, 3);
  OK(rt_render(ctx, sc, &cam, &p, sum.data(), argb.data(), &st));
  OK(rt_render(ctx, sc, &cam, &p, nullptr, argb.data(), nullptr));
  p.flags = RT_FLAG_COUNTERS;
  OK(rt_render(ctx, sc, &cam, &p, sum.data(), nullptr, &st));
  p.flags = 0; p.shard_index = 1; p.shard_count = 3; p.sample_offset = 5;
  OK(rt_render(ctx, sc, &cam, &p, sum.data(), argb.data(), &st));
  p = some_params(W, H, 40);  // more than 32 samples: chunks of several samples
  OK(rt_render(ctx, sc, &cam, &p, sum.data(), argb.data(), &st));
  p = some_params(W, H, 2);
  float* d_sum = nullptr;
  uint32_t* d_argb = nullptr;
  hipStream_t stream;
  HIP(hipMalloc(&d_sum, sum.size() * sizeof(float)));
  HIP(hipMalloc(&d_argb, argb.size() * sizeof(uint32_t)));
  HIP(hipStreamCreate(&stream));
  OK(rt_render_device(ctx, sc, &cam, &p, d_sum, d_argb, &st, stream));
  OK(rt_render_device(ctx, sc, &cam, &p, d_sum, nullptr, nullptr, stream));
  OK(rt_tonemap_device(ctx, d_sum, W * H, p.spp, d_argb, stream));
  OK(rt_render_device_noalloc(ctx, sc, &cam, &p, d_sum, d_argb, stream));
  HIP(hipStreamSynchronize(stream));
  HIP(hipMemcpy(argb.data(), d_argb, argb.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIP(hipStreamDestroy(stream));
  HIP(hipFree(d_sum));
  HIP(hipFree(d_argb));
  const int N = 77;
  std::vector<float> o((size_t)N * 3), d((size_t)N * 3), tm((size_t)N), xi((size_t)N * (size_t)(info.n_media ? info.n_media : 1));
  for (auto& x : o) x = (float)((double)(rng() % 2000) * 0.5 - 200.0);
  for (auto& x : d) x = (float)((double)(rng() % 2001) * 0.001 - 1.0);
  for (auto& x : tm) x = (float)((double)(rng() % 1000) * 0.001);
  for (auto& x : xi) x = (float)((double)(1 + rng() % 999) * 0.001);
  std::vector<rt_hit> hits((size_t)N);
  OK(rt_hit_batch(ctx, sc, N, o.data(), d.data(), tm.data(), 0.001f, 1e30f, info.n_media ? xi.data() : nullptr, hits.data()));
  OK(rt_hit_batch(ctx, sc, N, o.data(), d.data(), nullptr, 0.001f, 500.0f, nullptr, hits.data()));
  OK(rt_hit_batch_variant(ctx, sc, RT_VARIANT_RENDER, N, o.data(), d.data(), tm.data(), 0.001f, 1e30f, nullptr, hits.data()));
  for (int v = 0; v < RT_N_VARIANTS; ++v)
    if (mask & (1u << v)) OK(rt_hit_batch_variant(ctx, sc, v, N, o.data(), d.data(), tm.data(), 0.001f, 1e30f, nullptr, hits.data()));
  std::vector<int32_t> px((size_t)N), smp((size_t)N), bnc((size_t)N);
  for (int i = 0; i < N; ++i) { px[(size_t)i] = (int32_t)(rng() % (uint64_t)(W * H)); smp[(size_t)i] = (int32_t)(rng() % 100); bnc[(size_t)i] = (int32_t)(rng() % 8); }
  std::vector<rt_scatter> sca((size_t)N);
  OK(rt_scatter_batch(ctx, sc, 99, N, o.data(), d.data(), tm.data(), px.data(), smp.data(), bnc.data(), sca.data()));
  std::vector<float> rgb((size_t)N * 3);
  OK(rt_sample_batch(ctx, sc, &cam, &p, N, px.data(), smp.data(), rgb.data()));
  printf("  variant %d of mask 0x%x, %d prims, %d tree nodes (height %d, digest %016llx), %d media, %d bytes of LDS\n", variant, mask,
         info.n_prims_flat, n_nodes, height, (unsigned long long)digest, info.n_media, info.lds_bytes);
}

int main(int argc, char** argv) {
  std::mt19937_64 rng(5);
  printf("ABI %d\n", rt_abi_version());
  rt_ctx* ctx = nullptr;
  OK(rt_init(0, &ctx));
  rt_device_info di;
  OK(rt_get_device_info(ctx, &di));
  printf("device: %d CUs, %d kHz, %d bytes of LDS per workgroup\n", di.n_cu, di.clock_khz, di.lds_bytes_max);
  int a = 1;
  for (; a < argc && strcmp(argv[a], "--") != 0; ++a) {
    printf("%s\n", argv[a]);
    rt_scene* sc = nullptr;
    OK(rt_scene_create_from_file(ctx, argv[a], &sc));
    walk_scene(ctx, sc, rng);
    rt_scene_destroy(sc);
    const std::vector<uint8_t> blob = slurp(argv[a]);
    std::vector<uint8_t> odd(blob.size() + 1);  // an unaligned copy: the library must realign it
    memcpy(odd.data() + 1, blob.data(), blob.size());
    OK(rt_scene_create_from_memory(ctx, odd.data() + 1, blob.size(), &sc));
    rt_scene_destroy(sc);
  }
  // a world described by hand: a few hundred spheres in a list (one tree), a light, a checkered ground
  {
    const int NS = 300;
    std::vector<int32_t> prim_type, prim_mat, node_kind, node_i, children;
    std::vector<double> prim_d, node_d;
    for (int i = 0; i < NS + 1; ++i) {
      prim_type.push_back(RT_PRIM_SPHERE);
      double rec[RT_PRIM_STRIDE] = {0};
      if (i == NS) { rec[0] = 0; rec[1] = -1000; rec[2] = 0; rec[3] = 1000; }
      else { rec[0] = (double)(i % 20) * 30.0; rec[1] = 10.0; rec[2] = (double)(i / 20) * 30.0; rec[3] = 10.0; }
      prim_d.insert(prim_d.end(), rec, rec + RT_PRIM_STRIDE);
      prim_mat.push_back(i == NS ? 1 : (i % 7 == 0 ? 2 : 0));
      node_kind.push_back(RT_NODE_PRIM);
      const int32_t ni[RT_NODE_ISTRIDE] = {i, 0, 0, 0};
      node_i.insert(node_i.end(), ni, ni + RT_NODE_ISTRIDE);
      node_d.insert(node_d.end(), RT_NODE_DSTRIDE, 0.0);
      children.push_back(i);
    }
    node_kind.push_back(RT_NODE_LIST);
    const int32_t root_i[RT_NODE_ISTRIDE] = {0, NS + 1, 0, 0};
    node_i.insert(node_i.end(), root_i, root_i + RT_NODE_ISTRIDE);
    node_d.insert(node_d.end(), RT_NODE_DSTRIDE, 0.0);
    const int32_t mat_type[3] = {RT_MAT_LAMBERTIAN, RT_MAT_LAMBERTIAN, RT_MAT_DIFFUSE_LIGHT};
    const double mat_d[3 * RT_MAT_DSTRIDE] = {0};
    const int32_t mat_tex[3] = {0, 2, 1};
    const int32_t tex_type[3] = {RT_TEX_SOLID, RT_TEX_SOLID, RT_TEX_CHECKER};
    const double tex_d[3 * RT_TEX_DSTRIDE] = {0.6, 0.5, 0.4, 0, 7, 7, 7, 0, 0, 0, 0, 0};
    const int32_t tex_i[3 * RT_TEX_ISTRIDE] = {0, 0, 0, 0, 0, 1};
    rt_scene_desc desc;
    memset(&desc, 0, sizeof desc);
    desc.abi_version = RTMI_ABI_VERSION;
    desc.n_prims = NS + 1; desc.prim_type = prim_type.data(); desc.prim_d = prim_d.data(); desc.prim_mat = prim_mat.data();
    desc.n_nodes = NS + 2; desc.node_kind = node_kind.data(); desc.node_i = node_i.data(); desc.node_d = node_d.data();
    desc.n_children = NS + 1; desc.node_children = children.data(); desc.root = NS + 1;
    desc.n_mats = 3; desc.mat_type = mat_type; desc.mat_d = mat_d; desc.mat_tex = mat_tex;
    desc.n_tex = 3; desc.tex_type = tex_type; desc.tex_d = tex_d; desc.tex_i = tex_i;
    printf("a world described by hand\n");
    for (int builder : {RT_BVH_DEFAULT, RT_BVH_DEVICE_SAH, RT_BVH_HOST_SAH, RT_BVH_DEVICE_LBVH}) {
      rt_scene* sc = nullptr;
      OK(rt_scene_create_with(ctx, &desc, builder, &sc));
      walk_scene(ctx, sc, rng);
      rt_scene_destroy(sc);
    }
    const uint64_t need = rt_scene_file_size(&desc);
    std::vector<uint64_t> file((size_t)(need + 7) / 8);
    uint64_t written = 0;
    OK(rt_scene_write(&desc, file.data(), need, &written));
    rt_scene* sc = nullptr;
    OK(rt_scene_create_from_memory(ctx, file.data(), written, &sc));
    rt_scene_destroy(sc);
    OK(rt_scene_create(ctx, &desc, &sc));
    rt_scene_destroy(sc);
    // the same world on "three GPUs" (this one, three times): both splits
    const int32_t ids[3] = {0, 0, 0};
    rt_multi* m = nullptr;
    MOK(rt_multi_init(ids, 3, &m));
    printf("rt_multi over %d devices\n", rt_multi_device_count(m));
    rt_multi_scene* ms = nullptr;
    MOK(rt_multi_scene_create(m, &desc, &ms));
    const rt_camera cam = a_camera();
    rt_params p = some_params(52, 36, 7);
    std::vector<float> sum((size_t)52 * 36 * 3);
    std::vector<uint32_t> argb((size_t)52 * 36);
    rt_stats st;
    MOK(rt_multi_render(m, ms, &cam, &p, RT_SPLIT_TILES, sum.data(), argb.data(), &st));
    MOK(rt_multi_render(m, ms, &cam, &p, RT_SPLIT_SAMPLES, nullptr, argb.data(), nullptr));
    rt_multi_scene_destroy(ms);
    rt_multi_shutdown(m);
    const int rccl = rt_multi_selftest_rccl(0);
    printf("rt_multi_selftest_rccl: %d%s%s\n", rccl, rccl ? " " : "", rccl ? rt_multi_last_error() : "");
  }
  for (++a; a < argc; ++a) {
    const std::vector<uint8_t> file = slurp(argv[a]);
    int32_t w = 0, h = 0;
    // by content (JPEG, PNG, GIF or BMP), then once more through the decoder the first bytes name
    OK(rt_image_decode(file.data(), file.size(), &w, &h, nullptr, 0));
    std::vector<uint8_t> rgb((size_t)w * (size_t)h * 3), again(rgb.size());
    OK(rt_image_decode(file.data(), file.size(), &w, &h, rgb.data(), rgb.size()));
    const uint8_t c0 = file.empty() ? 0 : file[0];
    const char* kind = c0 == 0x89 ? "PNG" : c0 == 'G' ? "GIF" : c0 == 'B' ? "BMP" : "JPEG";
    OK((c0 == 0x89 ? rt_image_decode_png : c0 == 'G' ? rt_image_decode_gif : c0 == 'B' ? rt_image_decode_bmp : rt_image_decode_jpeg)(
        file.data(), file.size(), &w, &h, again.data(), again.size()));
    if (rgb != again) { printf("%s: the two entry points disagree\n", argv[a]); return 1; }
    printf("%s: %d x %d (%s)\n", argv[a], w, h, kind);
  }
  // what must be refused is refused
  {
    rt_scene* sc = nullptr;
    if (rt_scene_create(ctx, nullptr, &sc) == RT_OK) { fprintf(stderr, "a NULL description was accepted\n"); return 1; }
    if (rt_scene_create_from_file(ctx, "/nonexistent/file", &sc) == RT_OK) { fprintf(stderr, "a missing file was accepted\n"); return 1; }
  }
  // Worlds come and go without taking device memory with them. (The runtime's allocator keeps some of what it is handed
  // back - free memory sinks by some tens of MB over the first hundred rounds and then stays - so what is checked is that
  // the second half of the rounds loses far less than the first: a leak per round would lose as much.)
  if (argc > 1 && strcmp(argv[1], "--") != 0) {
    size_t free0 = 0, free_half = 0, free1 = 0, total = 0;
    rt_scene* sc = nullptr;
    OK(rt_scene_create_from_file(ctx, argv[1], &sc));  // (first use pays for what the runtime keeps: modules, queues)
    rt_scene_destroy(sc);
    HIP(hipDeviceSynchronize());
    HIP(hipMemGetInfo(&free0, &total));
    const int rounds = getenv("RT_ROUNDS") ? atoi(getenv("RT_ROUNDS")) : 100;
    for (int round = 0; round < rounds; ++round) {
      for (int b = 1; b < argc && strcmp(argv[b], "--") != 0; ++b) {
        OK(rt_scene_create_from_file(ctx, argv[b], &sc));
        const rt_camera cam = a_camera();
        rt_params p = some_params(40, 24, 2);
        std::vector<uint32_t> argb(40 * 24);
        OK(rt_render(ctx, sc, &cam, &p, nullptr, argb.data(), nullptr));
        rt_scene_destroy(sc);
      }
      if (round == rounds / 2) {
        HIP(hipDeviceSynchronize());
        HIP(hipMemGetInfo(&free_half, &total));
      }
    }
    HIP(hipDeviceSynchronize());
    HIP(hipMemGetInfo(&free1, &total));
    const long long first = (long long)free0 - (long long)free_half, second = (long long)free_half - (long long)free1;
    printf("device memory free over %d rounds of create-render-destroy of every scene: first half lost %lld bytes, second half %lld\n",
           rounds, first, second);
    if (second > first / 2 + (16 << 20)) { fprintf(stderr, "device memory is leaking: %lld bytes in the second half of the rounds\n", second); return 1; }
  }
  rt_shutdown(ctx);
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
  // host allocations still reachable from nowhere (the runtime's own are reported too: read the frames)
  if (getenv("RT_LEAK_CHECK")) __lsan_do_recoverable_leak_check();
#endif
#endif
  printf("done\n");
  fflush(stdout);
  _exit(0);  // (ROCm's sanitizer runtime trips over its own device allocator when the process unwinds normally)
}
