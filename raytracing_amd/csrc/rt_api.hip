// rt_api.hip — the C ABI of include/rtmi.h: validation, the host-side rewrite of the flat scene
// description into the device layout of rt_scene.h, uploads, and the render / parity entry
// points. No C++ type, exception or abort crosses this boundary.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <new>
#include <set>
#include <string>
#include <utility>
#include <array>
#include <functional>
#include <vector>

#include "../../include/rtmi.h"
#include "rt_launch.h"
#include "rt_scene.h"

using namespace rt;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

}  // namespace
namespace rt {
// for the library's other translation units: sets what rt_last_error() returns on this thread
int set_error(int code, const char* message) {
  g_err = message;
  return code;
}
}  // namespace rt
namespace {

// Runs f(); a C++ exception never leaves the library.
template <class F>
int guarded(const char* what, F&& f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return fail(RT_ERR_OOM, "out of host memory in %s", what);
  } catch (const std::exception& e) {
    return fail(RT_ERR_ARG, "%s: %s", what, e.what());
  } catch (...) {
    return fail(RT_ERR_ARG, "%s: unknown failure", what);
  }
}

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP,   \
                                      "%s failed: %s", #expr, hipGetErrorString(e_));        \
  } while (0)

}  // namespace

struct rt_ctx {
  int device = 0;
  int n_cu = 0;
  int clock_khz = 0;
  int wall_khz = 0;  // rate of the constant timer (s_memrealtime)
  size_t lds_max = 65536;  // LDS a workgroup may ask for on this device
  unsigned char* scratch = nullptr;  // this device's copy of the file-scope scratch array (rt_render_device_noalloc)
  // reusable device workspace
  uint32_t* d_queue = nullptr;              // the heads of the work-item queue (RT_QUEUE_HEADS lines)
  unsigned long long* d_stats = nullptr;    // RT_STATS_WORDS words
  float* d_partial = nullptr;               // n_chunks * W*H*3 partial sums
  size_t partial_bytes = 0;
  void* d_io = nullptr;                     // staging for the host-buffer entry points
  size_t io_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

struct rt_scene {
  rt_ctx* ctx = nullptr;
  DScene S{};
  std::vector<void*> allocs;
  rt_scene_info info{};
  int n_media = 0;
  bool has_time_bounds = false;
  float time_lo = 0.f, time_hi = 1.f;
  int blocks_per_cu[2] = {0, 0};  // occupancy of k_render<false>, k_render<true>
  bool pool = false;              // the world can render on k_render_pool (RT_FLAG_POOL_PATHS)
  int pool_blocks_per_cu[2] = {0, 0};
  const DBvhNode* d_tree = nullptr;  // the unpacked node records as the builder wrote them
  int n_tree_nodes = 0, tree_height = 0;
  int n_shadowed_spheres = 0;     // duplicates left out of their groups' tested ranges (exact for t-min >= 0)
};

namespace {

int ensure_io(rt_ctx* ctx, size_t bytes) {
  if (ctx->io_bytes >= bytes) return RT_OK;
  if (ctx->d_io) hipFree(ctx->d_io);
  ctx->d_io = nullptr;
  ctx->io_bytes = 0;
  HIPCHK(hipMalloc(&ctx->d_io, bytes));
  ctx->io_bytes = bytes;
  return RT_OK;
}

// ----------------------------------------------------------------------------- validation
bool finite_all(const double* p, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (!std::isfinite(p[i])) return false;
  return true;
}

int validate_desc(const rt_scene_desc* d) {
  if (!d) return fail(RT_ERR_ARG, "scene description is NULL");
  if (d->abi_version != RTMI_ABI_VERSION)
    return fail(RT_ERR_ARG, "abi_version %d, library speaks %d", d->abi_version, RTMI_ABI_VERSION);
  if (d->n_prims < 0 || d->n_nodes <= 0 || d->n_children < 0 || d->n_mats < 0 || d->n_tex < 0 ||
      d->n_perlin < 0 || d->n_images < 0 || d->n_media < 0)
    return fail(RT_ERR_ARG, "negative or empty count in scene description");
  if (d->root < 0 || d->root >= d->n_nodes) return fail(RT_ERR_ARG, "root node %d out of range", d->root);
  if ((d->n_prims && (!d->prim_type || !d->prim_d || !d->prim_mat)) || !d->node_kind || !d->node_i || !d->node_d ||
      (d->n_children && !d->node_children) || (d->n_mats && (!d->mat_type || !d->mat_d || !d->mat_tex)) ||
      (d->n_tex && (!d->tex_type || !d->tex_d || !d->tex_i)) || (d->n_perlin && (!d->perlin_ranvec || !d->perlin_perm)) ||
      (d->n_images && (!d->img_w || !d->img_h || !d->img_rgb8)))
    return fail(RT_ERR_ARG, "NULL array with a non-zero count in scene description");
  for (int i = 0; i < d->n_prims; ++i) {
    int t = d->prim_type[i];
    if (t < RT_PRIM_SPHERE || t > RT_PRIM_YZ_RECT) return fail(RT_ERR_ARG, "prim %d: unknown type %d", i, t);
    if (d->prim_mat[i] < 0 || d->prim_mat[i] >= d->n_mats) return fail(RT_ERR_ARG, "prim %d: material out of range", i);
    if (!finite_all(d->prim_d + (size_t)i * RT_PRIM_STRIDE, RT_PRIM_STRIDE))
      return fail(RT_ERR_ARG, "prim %d: non-finite parameter", i);
  }
  for (int i = 0; i < d->n_mats; ++i) {
    int t = d->mat_type[i];
    if (t < RT_MAT_LAMBERTIAN || t > RT_MAT_ISOTROPIC) return fail(RT_ERR_ARG, "material %d: unknown type %d", i, t);
    bool textured = t == RT_MAT_LAMBERTIAN || t == RT_MAT_DIFFUSE_LIGHT || t == RT_MAT_ISOTROPIC;
    // metal and dielectric carry no texture: -1, or any valid index (it is never read)
    if (d->mat_tex[i] >= d->n_tex || d->mat_tex[i] < (textured ? 0 : -1))
      return fail(RT_ERR_ARG, "material %d: texture out of range", i);
  }
  for (int i = 0; i < d->n_tex; ++i) {
    int t = d->tex_type[i];
    const int32_t* ti = d->tex_i + (size_t)i * RT_TEX_ISTRIDE;
    if (t < RT_TEX_SOLID || t > RT_TEX_IMAGE) return fail(RT_ERR_ARG, "texture %d: unknown type %d", i, t);
    if (t == RT_TEX_CHECKER && (ti[0] < 0 || ti[0] >= d->n_tex || ti[1] < 0 || ti[1] >= d->n_tex))
      return fail(RT_ERR_ARG, "texture %d: checker branch out of range", i);
    if (t == RT_TEX_NOISE && (ti[0] < 0 || ti[0] >= d->n_perlin)) return fail(RT_ERR_ARG, "texture %d: perlin id out of range", i);
    if (t == RT_TEX_IMAGE && ti[0] >= d->n_images) return fail(RT_ERR_ARG, "texture %d: image id out of range", i);
  }
  // checker textures must bottom out within the depth the kernels walk
  for (int i = 0; i < d->n_tex; ++i) {
    std::vector<std::pair<int, int>> st{{i, 0}};
    while (!st.empty()) {
      auto [t, depth] = st.back();
      st.pop_back();
      if (d->tex_type[t] != RT_TEX_CHECKER) continue;
      if (depth >= 15) return fail(RT_ERR_UNSUPPORTED, "texture %d: checker nesting deeper than 15 (or cyclic)", i);
      st.push_back({d->tex_i[(size_t)t * RT_TEX_ISTRIDE], depth + 1});
      st.push_back({d->tex_i[(size_t)t * RT_TEX_ISTRIDE + 1], depth + 1});
    }
  }
  for (int i = 0; i < d->n_perlin * 3 * 256; ++i)
    if (d->perlin_perm[i] < 0 || d->perlin_perm[i] > 255) return fail(RT_ERR_ARG, "perlin permutation entry out of 0..255");
  for (int i = 0; i < d->n_images; ++i)
    if (d->img_w[i] <= 0 || d->img_h[i] <= 0 || !d->img_rgb8[i]) return fail(RT_ERR_ARG, "image %d: bad size or NULL data", i);
  std::vector<char> medium_seen((size_t)d->n_media, 0);
  for (int i = 0; i < d->n_nodes; ++i) {
    const int32_t* ni = d->node_i + (size_t)i * RT_NODE_ISTRIDE;
    auto node_ok = [&](int n) { return n >= 0 && n < d->n_nodes; };
    switch (d->node_kind[i]) {
      case RT_NODE_PRIM:
        if (ni[0] < 0 || ni[0] >= d->n_prims) return fail(RT_ERR_ARG, "node %d: prim out of range", i);
        break;
      case RT_NODE_LIST:
        if (ni[0] < 0 || ni[1] < 0 || (long long)ni[0] + ni[1] > d->n_children)
          return fail(RT_ERR_ARG, "node %d: child range out of bounds", i);
        for (int c = 0; c < ni[1]; ++c)
          if (!node_ok(d->node_children[ni[0] + c])) return fail(RT_ERR_ARG, "node %d: child out of range", i);
        break;
      case RT_NODE_BVH:
        if (!node_ok(ni[0]) || !node_ok(ni[1])) return fail(RT_ERR_ARG, "node %d: bvh child out of range", i);
        break;
      case RT_NODE_TRANSLATE:
      case RT_NODE_ROTATE_Y:
        if (!node_ok(ni[0])) return fail(RT_ERR_ARG, "node %d: child out of range", i);
        if (!finite_all(d->node_d + (size_t)i * RT_NODE_DSTRIDE, 3)) return fail(RT_ERR_ARG, "node %d: non-finite transform", i);
        break;
      case RT_NODE_MEDIUM:
        if (!node_ok(ni[0])) return fail(RT_ERR_ARG, "node %d: boundary out of range", i);
        if (ni[1] < 0 || ni[1] >= d->n_mats) return fail(RT_ERR_ARG, "node %d: phase material out of range", i);
        if (ni[2] < 0 || ni[2] >= d->n_media || medium_seen[ni[2]]++) return fail(RT_ERR_ARG, "node %d: medium index invalid or repeated", i);
        break;
      default:
        return fail(RT_ERR_ARG, "node %d: unknown kind %d", i, d->node_kind[i]);
    }
  }
  return RT_OK;
}

// ------------------------------------------------------------------------------ flattening
struct Sink {  // leaves in scan order with the chain each was reached under, duplicates dropped
  std::vector<std::pair<int, int>> seq;  // (chain, primitive)
  std::set<std::pair<int, int>> seen;
  std::set<std::pair<int, int>> walked;  // (node, chain) already expanded into this sink
  void add(int chain, int prim) {
    if (!seen.insert({chain, prim}).second) return;
    seq.push_back({chain, prim});
  }
};

struct HostMedium {
  int outer_chain = 0, phase_mat = 0;
  float neg_inv_density = 0.f;
  Sink boundary;
  bool present = false;
};

#define RT_MAX_TREE_DEPTH 2048
struct Flattener {
  const rt_scene_desc* d;
  std::vector<std::vector<DChainOp>> chains{{}};  // chain 0 = identity
  Sink surface;
  std::vector<HostMedium> media;
  std::vector<char> on_stack;
  int err = RT_OK;

  int intern(const std::vector<DChainOp>& ops) {
    for (size_t i = 0; i < chains.size(); ++i) {
      if (chains[i].size() != ops.size()) continue;
      if (ops.empty() || std::memcmp(chains[i].data(), ops.data(), ops.size() * sizeof(DChainOp)) == 0) return (int)i;
    }
    chains.push_back(ops);
    return (int)chains.size() - 1;
  }

  // The walk runs through the description BACKWARDS (last child first) and the sinks are turned round at the end:
  // a leaf then stands where the scan meets it for the LAST time. That is the occurrence that decides an exact
  // tie - a rect accepts t = closest-so-far (rectangles.clj), so of two coinciding rects the one the scan tests
  // later wins, and a wall listed twice wins from its second place. A subtree reached again under the same
  // chain (going backwards: reached earlier in the scan) adds nothing, so it is not expanded again: a
  // description that shares subtrees (the reference's bvhnode does, base.clj:143-160) costs what it holds, not
  // what it unfolds to. Depth is bounded because the walk recurses on the host stack.
  void walk(int node, int chain, Sink& sink, bool in_boundary, int depth = 0) {
    if (err) return;
    if (on_stack[node]) { err = fail(RT_ERR_ARG, "node %d: the hittable tree has a cycle", node); return; }
    if (depth > RT_MAX_TREE_DEPTH) { err = fail(RT_ERR_UNSUPPORTED, "the hittable tree is more than %d levels deep", RT_MAX_TREE_DEPTH); return; }
    if (!sink.walked.insert({node, chain}).second) return;
    on_stack[node] = 1;
    const int32_t* ni = d->node_i + (size_t)node * RT_NODE_ISTRIDE;
    const double* nd = d->node_d + (size_t)node * RT_NODE_DSTRIDE;
    switch (d->node_kind[node]) {
      case RT_NODE_PRIM:
        sink.add(chain, ni[0]);
        break;
      case RT_NODE_LIST:
        for (int c = ni[1]; c-- > 0 && !err;) walk(d->node_children[ni[0] + c], chain, sink, in_boundary, depth + 1);
        break;
      case RT_NODE_BVH:
        walk(ni[1], chain, sink, in_boundary, depth + 1);
        walk(ni[0], chain, sink, in_boundary, depth + 1);
        break;
      case RT_NODE_TRANSLATE:
      case RT_NODE_ROTATE_Y: {
        std::vector<DChainOp> ops = chains[chain];
        if (ops.size() >= RT_MAX_CHAIN_OPS) {
          err = fail(RT_ERR_UNSUPPORTED, "node %d: more than %d nested translate/rotate-y wrappers", node, RT_MAX_CHAIN_OPS);
          break;
        }
        DChainOp op;
        std::memset(&op, 0, sizeof op);
        op.kind = d->node_kind[node] == RT_NODE_TRANSLATE ? DOP_TRANSLATE : DOP_ROTATE_Y;
        op.da = nd[0]; op.db = nd[1]; op.dc = op.kind == DOP_TRANSLATE ? nd[2] : 0.0;
        op.a = (float)op.da; op.b = (float)op.db; op.c = (float)op.dc;
        ops.push_back(op);
        walk(ni[0], intern(ops), sink, in_boundary, depth + 1);
        break;
      }
      case RT_NODE_MEDIUM: {
        if (in_boundary) {
          err = fail(RT_ERR_UNSUPPORTED, "node %d: a constant medium inside another medium's boundary", node);
          break;
        }
        HostMedium& M = media[ni[2]];
        M.present = true;
        M.outer_chain = chain;
        M.phase_mat = ni[1];
        M.neg_inv_density = (float)nd[0];
        walk(ni[0], chain, M.boundary, true, depth + 1);
        break;
      }
    }
    on_stack[node] = 0;
  }
};

template <class T>
int upload(rt_scene* sc, const std::vector<T>& v, const T** out) {
  *out = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, bytes));
  sc->allocs.push_back(p);
  if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = static_cast<const T*>(p);
  return RT_OK;
}

bool tex_needs_uv(const rt_scene_desc* d, int tex, int depth = 0) {
  if (tex < 0 || depth > 16) return false;
  int t = d->tex_type[tex];
  if (t == RT_TEX_IMAGE) return true;
  if (t == RT_TEX_CHECKER)
    return tex_needs_uv(d, d->tex_i[(size_t)tex * RT_TEX_ISTRIDE], depth + 1) ||
           tex_needs_uv(d, d->tex_i[(size_t)tex * RT_TEX_ISTRIDE + 1], depth + 1);
  return false;
}

void fill_camera(const rt_camera* c, const rt_params* p, DCamera& o) {
  o.wm1 = (double)(p->width - 1); o.hm1 = (double)(p->height - 1);
  o.inv_wm1 = 1.0 / o.wm1; o.inv_hm1 = 1.0 / o.hm1;
  for (int a = 0; a < 3; ++a) {
    o.origin[a] = c->origin[a]; o.llc[a] = c->lower_left_corner[a];
    o.horizontal[a] = c->horizontal[a]; o.vertical[a] = c->vertical[a];
    o.u[a] = c->u[a]; o.v[a] = c->v[a];
  }
  o.lens_radius = c->lens_radius; o.time0 = c->time0; o.time1 = c->time1;
  o.pinhole = (c->lens_radius == 0.0 && p->width > 1 && p->height > 1) ? 1 : 0;
  for (int a = 0; a < 3; ++a) {
    o.dC[a] = c->lower_left_corner[a] - c->origin[a];
    o.dH[a] = c->horizontal[a] / o.wm1;
    o.dV[a] = c->vertical[a] / o.hm1;
    o.fO[a] = (float)c->origin[a];
  }
  o.pad_ = 0;
}

int fill_render(const rt_scene* sc, const rt_camera* cam, const rt_params* p, DRender& R) {
  if (!cam || !p) return fail(RT_ERR_ARG, "camera or params is NULL");
  if (p->width <= 0 || p->height <= 0 || p->width > 65535 || p->height > 65535 || (long long)p->width * p->height > (1ll << 28))
    return fail(RT_ERR_ARG, "image size %dx%d out of range", p->width, p->height);
  if (p->spp <= 0 || p->sample_offset < 0 || p->max_depth < 0 || p->max_depth > (1 << 26))
    return fail(RT_ERR_ARG, "spp %d / sample_offset %d / max_depth %d out of range", p->spp, p->sample_offset, p->max_depth);
  int sc_count = p->shard_count <= 0 ? 1 : p->shard_count;
  if (p->shard_index < 0 || p->shard_index >= sc_count) return fail(RT_ERR_ARG, "shard %d of %d", p->shard_index, sc_count);
  if (sc->has_time_bounds) {
    double lo = std::min(cam->time0, cam->time1), hi = std::max(cam->time0, cam->time1);
    if (lo < sc->time_lo || hi > sc->time_hi)
      return fail(RT_ERR_ARG, "camera shutter [%g, %g] leaves the interval [%g, %g] moving spheres are bounded for",
                  lo, hi, sc->time_lo, sc->time_hi);
  }
  R.width = p->width; R.height = p->height;
  R.spp = p->spp; R.sample_offset = p->sample_offset; R.max_depth = p->max_depth;
  R.shard_index = p->shard_index; R.shard_count = sc_count;
  R.tiles_x = (p->width + RT_TILE - 1) / RT_TILE;
  R.tiles_y = (p->height + RT_TILE - 1) / RT_TILE;
  int n_tiles = R.tiles_x * R.tiles_y;
  // every full group of sc_count tiles holds one tile of this shard; the last, partial group may (shard_tile)
  {
    const int full = n_tiles / sc_count;
    R.n_owned_tiles = full + (shard_tile(full, R.shard_index, sc_count, R.tiles_x) < n_tiles ? 1 : 0);
  }
  // A pixel's samples are split into work items ("chunks") by a schedule that depends on spp
  // alone, so the FP32 summation order of a pixel does not depend on how the image is sharded.
  // 64 equal chunks carry 80 % of the samples, then five tiers shrink down to single samples. The
  // queue hands items out in chunk order, so the last items are one path long and the waves drain
  // together. Items must also be short against the job: waves of one SIMD do not progress at the
  // same rate, and a slow wave is still inside the item it took early when the fast ones have
  // emptied the queue (with 24 chunks of 33 samples an eighth-of-the-image shard lost 12 %).
  {  // (fixed arrays: this runs on the entry point that may not allocate)
    int sizes[RT_MAX_CHUNKS + 64];
    int n_sizes = 0;
    const int spp = p->spp;
    if (spp <= 128) {  // too few samples for tiers: up to 32 equal chunks
      const int n = std::min(spp, 32), sz = (spp + n - 1) / n;
      for (int left = spp; left > 0; left -= sz) sizes[n_sizes++] = std::min(sz, left);
    } else {
      const int tier_count[5] = {8, 8, 8, 8, 16};
      const int tier_permille[5] = {12, 6, 3, 2, 1};  // of spp, per chunk
      int tail[48];
      int n_tail = 0, tail_first = 0, tail_sum = 0;
      for (int t = 0; t < 5; ++t)
        for (int k = 0; k < tier_count[t]; ++k) {
          int sz = std::max(1, (int)((long long)spp * tier_permille[t] / 1000));
          tail[n_tail++] = sz;
          tail_sum += sz;
        }
      int rem = spp - tail_sum;  // samples of the equal chunks
      while (rem <= 0) { tail_sum -= tail[tail_first++]; rem = spp - tail_sum; }
      const int big = (rem + 63) / 64;
      for (int left = rem; left > 0; left -= big) sizes[n_sizes++] = std::min(big, left);
      for (int k = tail_first; k < n_tail; ++k) sizes[n_sizes++] = tail[k];
    }
    if (n_sizes > RT_MAX_CHUNKS) return fail(RT_ERR_ARG, "internal: %d chunks", n_sizes);
    R.n_chunks = n_sizes;
    int at = 0;
    for (int c = 0; c < R.n_chunks; ++c) { R.chunk_start[c] = at; at += sizes[c]; }
    for (int c = R.n_chunks; c <= RT_MAX_CHUNKS; ++c) R.chunk_start[c] = at;  // == spp
  }
  long long n_items = (long long)R.n_owned_tiles * 64 * R.n_chunks;
  if (n_items > 0x7fffffffll - 4096) return fail(RT_ERR_ARG, "too many work items (%lld)", n_items);
  R.n_items = (int)n_items;
  for (int a = 0; a < 3; ++a) R.bg[a] = (float)p->background[a];
  R.seed_lo = (uint32_t)p->seed; R.seed_hi = (uint32_t)(p->seed >> 32);
  return RT_OK;
}

int init_ctx(rt_ctx* ctx) {
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, ctx->device));
  ctx->n_cu = prop.multiProcessorCount;
  ctx->clock_khz = prop.clockRate;
  int wall_khz = 0;
  if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, ctx->device) == hipSuccess) ctx->wall_khz = wall_khz;
  int lds_max = 0;
  HIPCHK(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device));
  ctx->lds_max = (size_t)lds_max;
  HIPCHK(hipMalloc(&ctx->d_queue, RT_QUEUE_HEADS * RT_QUEUE_STRIDE * 4));
  HIPCHK(hipMalloc(&ctx->d_stats, sizeof(unsigned long long) * RT_STATS_WORDS));
  HIPCHK(hipEventCreate(&ctx->ev0));
  HIPCHK(hipEventCreate(&ctx->ev1));
  return RT_OK;
}

}  // namespace

// ========================================================================== the C ABI
extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }
int rt_abi_version(void) { return RTMI_ABI_VERSION; }

static int lookup_scratch(rt_ctx* ctx);

int rt_init(int32_t device_id, rt_ctx** out) {
  if (!out) return fail(RT_ERR_ARG, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(RT_ERR_HIP, "no HIP device available (%s); this library has no CPU path",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= n) return fail(RT_ERR_ARG, "device %d out of range (have %d)", device_id, n);
  HIPCHK(hipSetDevice(device_id));
  rt_ctx* ctx = new (std::nothrow) rt_ctx();
  if (!ctx) return fail(RT_ERR_OOM, "out of host memory");
  ctx->device = device_id;
  int rc = init_ctx(ctx);
  if (!rc) rc = lookup_scratch(ctx);
  if (rc) {  // whatever was allocated so far goes back; the message of the failing call stays
    rt_shutdown(ctx);
    return rc;
  }
  *out = ctx;
  return RT_OK;
}

void rt_shutdown(rt_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipFree(ctx->d_queue);
  hipFree(ctx->d_stats);
  hipFree(ctx->d_partial);
  hipFree(ctx->d_io);
  if (ctx->ev0) hipEventDestroy(ctx->ev0);
  if (ctx->ev1) hipEventDestroy(ctx->ev1);
  delete ctx;
}

void rt_scene_destroy(rt_scene* sc) {
  if (!sc) return;
  if (sc->ctx) hipSetDevice(sc->ctx->device);
  for (void* p : sc->allocs) hipFree(p);
  delete sc;
}

// May throw (std::vector, std::map, std::function allocate): only called through guarded().
static int scene_create_impl(rt_ctx* ctx, const rt_scene_desc* d, int bvh_builder, rt_scene** out) {
  if (!ctx || !out) return fail(RT_ERR_ARG, "ctx or out is NULL");
  *out = nullptr;
  if (bvh_builder < RT_BVH_DEFAULT || bvh_builder > RT_BVH_DEVICE_LBVH) return fail(RT_ERR_ARG, "unknown BVH builder %d", bvh_builder);
  int rc = validate_desc(d);
  if (rc) return rc;
  HIPCHK(hipSetDevice(ctx->device));

  Flattener F;
  F.d = d;
  F.media.resize((size_t)d->n_media);
  F.on_stack.assign((size_t)d->n_nodes, 0);
  F.walk(d->root, 0, F.surface, false);
  if (F.err) return F.err;
  std::reverse(F.surface.seq.begin(), F.surface.seq.end());
  for (HostMedium& M : F.media) std::reverse(M.boundary.seq.begin(), M.boundary.seq.end());

  // device primitive records, one contiguous run per group
  std::vector<DPrim> prims;
  std::vector<DGroup> groups;
  bool any_moving = false;
  double tlo = 0.0, thi = 1.0;
  uint32_t features = 0;  // F_* mask
  std::vector<DGroup64> groups64;
  std::vector<float> brute_rects;
  std::vector<float> brute_boxes;
  // the chain's ray side collapsed to one rigid motion about y: o' = Ry(o) + b (see DGroup)
  auto chain_motion = [&](int chain, DGroup& G, DGroup64& H) {
    double c = 1.0, s = 0.0, b[3] = {0.0, 0.0, 0.0};
    for (const DChainOp& op : F.chains[chain]) {
      if (op.kind == DOP_TRANSLATE) {
        b[0] -= op.da; b[1] -= op.db; b[2] -= op.dc;
      } else {
        const double so = op.da, co = op.db;
        const double nc = co * c - so * s, ns = so * c + co * s;
        const double bx = co * b[0] - so * b[2], bz = so * b[0] + co * b[2];
        c = nc; s = ns; b[0] = bx; b[2] = bz;
      }
    }
    std::memset(&H, 0, sizeof H);
    H.rc = c; H.rs = s; H.bx = b[0]; H.by = b[1]; H.bz = b[2];
    G.rc = (float)c; G.rs = (float)s; G.bx = (float)b[0]; G.by = (float)b[1]; G.bz = (float)b[2];
  };
  // extent of a primitive's bounding box along its longest axis, and its centre
  auto prim_extent = [&](int pi, double c[3]) {
    const int t = d->prim_type[pi];
    const double* pd = d->prim_d + (size_t)pi * RT_PRIM_STRIDE;
    double lo[3], hi[3];
    if (t == RT_PRIM_SPHERE) {
      for (int a = 0; a < 3; ++a) { lo[a] = pd[a] - std::fabs(pd[3]); hi[a] = pd[a] + std::fabs(pd[3]); }
    } else if (t == RT_PRIM_MOVING_SPHERE) {
      for (int a = 0; a < 3; ++a) {
        lo[a] = std::min(pd[a], pd[3 + a]) - std::fabs(pd[8]);
        hi[a] = std::max(pd[a], pd[3 + a]) + std::fabs(pd[8]);
      }
    } else {
      const int ax_a = t == RT_PRIM_YZ_RECT ? 1 : 0, ax_b = t == RT_PRIM_XY_RECT ? 1 : 2, ax_k = 3 - ax_a - ax_b;
      lo[ax_a] = std::min(pd[0], pd[1]); hi[ax_a] = std::max(pd[0], pd[1]);
      lo[ax_b] = std::min(pd[2], pd[3]); hi[ax_b] = std::max(pd[2], pd[3]);
      lo[ax_k] = hi[ax_k] = pd[4];
    }
    double e = 0.0;
    for (int a = 0; a < 3; ++a) { c[a] = 0.5 * (lo[a] + hi[a]); e = std::max(e, hi[a] - lo[a]); }
    return e;
  };
  // A list entry is a primitive of the description and, for a sphere carried into world space (below), the
  // chain its record still goes through: primitive | record chain << 32.
  typedef long long Entry;
  auto e_prim = [](Entry e) { return (int)(e & 0xffffffffll); };
  auto e_chain = [](Entry e) { return (int)(e >> 32); };
  std::function<void(int, std::vector<Entry>)> emit_one;
  bool drop_shadowed_spheres = true;  // surface groups; switched off while media boundaries are emitted
  int n_shadowed_total = 0;
  // Spheres below a chain are intersected in WORLD space: a chain is a rigid motion, so the sphere carried
  // through its inverse meets the world ray at the same t. They join the groups (and trees) of the
  // chain-less primitives instead of forming groups of their own, each of which costs every ray an FP64
  // ray transform and a tree walk as long as its slowest lane's. Their records are still built in the
  // chain's space from the untouched primitive (prims_rec) and unwound through the chain: the primitive's
  // group is a header group that holds the chain's motion and is never tested.
  std::vector<DPrim> prims_rec;
  std::vector<std::pair<size_t, int>> world_spheres;  // (device primitive, record chain), headers assigned later
  bool spheres_to_world = true;
  const bool lid_boxes_as_leaves = RT_LIDBOX_LEAVES != 0;
  auto emit_groups = [&](const Sink& sink) {
    // Groups are tested one after the other and a rect accepts t = closest-so-far (rectangles.clj), so on an exact
    // tie the group tested LATER wins - as the later primitive does in the reference's scan. The groups therefore
    // follow the scan: a group is a run of consecutive leaves under one chain. A later run of a chain joins an
    // earlier one (fewer groups) only where that cannot reorder a tie: no group in between holds a rect of an
    // orientation the run holds too. (Two XZ rects at one height under different rotate-y wrappers tie exactly,
    // whatever the angles: the turn leaves y alone.) Spheres need `<` to displace a hit and never tie by accident.
    struct Run { int chain; std::vector<Entry> list; unsigned rect_types; };
    auto type_bit = [&](int pi) -> unsigned { int t = d->prim_type[pi]; return t >= RT_PRIM_XY_RECT ? 1u << (t - RT_PRIM_XY_RECT) : 0u; };
    std::vector<Run> runs;
    for (const auto& cp : sink.seq) {
      const bool to_world = spheres_to_world && cp.first != 0 && d->prim_type[cp.second] <= RT_PRIM_MOVING_SPHERE;
      const int chain = to_world ? 0 : cp.first;
      if (runs.empty() || runs.back().chain != chain) runs.push_back(Run{chain, {}, 0u});
      runs.back().list.push_back((Entry)cp.second | ((Entry)(to_world ? cp.first : 0) << 32));
      runs.back().rect_types |= type_bit(cp.second);
    }
    // merge pass: a run joins the nearest earlier run of its chain when nothing between them shares a rect orientation with it
    for (size_t i = 0; i < runs.size(); ++i) {
      if (runs[i].list.empty()) continue;
      unsigned between = 0;
      for (size_t j = i; j-- > 0;) {
        if (runs[j].list.empty()) continue;
        if (runs[j].chain == runs[i].chain) {
          if (!(between & runs[i].rect_types)) {
            runs[j].list.insert(runs[j].list.end(), runs[i].list.begin(), runs[i].list.end());
            runs[j].rect_types |= runs[i].rect_types;
            runs[i].list.clear();
          }
          break;
        }
        between |= runs[j].rect_types;
      }
    }
    for (const Run& run : runs) {
      if (run.list.empty()) continue;
      const int chain = run.chain;
      std::vector<Entry> list = run.list;
      // by type, stably: XY, XZ, YZ rects, then spheres (scan order kept within a type)
      auto type_rank = [&](Entry e) { int t = d->prim_type[e_prim(e)]; return t >= RT_PRIM_XY_RECT ? t - RT_PRIM_XY_RECT : 3; };
      std::stable_sort(list.begin(), list.end(), [&](Entry x, Entry y) { return type_rank(x) < type_rank(y); });
      // A BVH over Morton codes is only as good as its primitives are alike in size: one huge
      // primitive (the r = 1000 ground spheres of the random-spheres world) inflates the box of
      // every ancestor it has and stretches the Morton grid until the small primitives share a few
      // cells. Primitives much larger than the spread of the others' centres go into a brute-force
      // group of their own, scanned BEFORE the tree: the tree's leaf test applies the tie rule on
      // ranks explicitly, so an exact tie across the two groups still falls as in one scan.
      if ((int)list.size() > RT_BRUTE_MAX) {
        std::vector<double> ext(list.size());
        std::vector<std::array<double, 3>> ctr(list.size());
        for (size_t i = 0; i < list.size(); ++i) ext[i] = prim_extent(e_prim(list[i]), ctr[i].data());
        std::vector<double> sorted = ext;
        std::nth_element(sorted.begin(), sorted.begin() + (long)sorted.size() / 2, sorted.end());
        const double median = sorted[sorted.size() / 2];
        std::vector<Entry> huge, rest;
        for (size_t i = 0; i < list.size(); ++i)
          (ext[i] > 16.0 * median && median > 0.0 ? huge : rest).push_back(list[i]);
        if (!huge.empty() && (int)huge.size() <= RT_BRUTE_MAX && !rest.empty()) {
          emit_one(chain, huge);
          emit_one(chain, rest);
          continue;
        }
      }
      emit_one(chain, list);
    }
  };
  emit_one = [&](int chain, std::vector<Entry> list) {
    {
      // what the device tests: a sphere's centre(s) and radius as the description holds them, or carried
      // into world space through the inverse of its record chain's motion (o' = Ry(o) + b, see DGroup)
      auto sphere_fields = [&](Entry e, double out[9]) {
        const int pi = e_prim(e);
        const double* pd = d->prim_d + (size_t)pi * RT_PRIM_STRIDE;
        const int n = d->prim_type[pi] == RT_PRIM_SPHERE ? 4 : 9;
        for (int k = 0; k < n; ++k) out[k] = pd[k];
        if (e_chain(e) == 0) return;
        DGroup G0;
        DGroup64 H0;
        std::memset(&G0, 0, sizeof G0);
        chain_motion(e_chain(e), G0, H0);
        for (int c0 = 0; c0 < (n == 4 ? 3 : 6); c0 += 3) {
          const long double lx = (long double)pd[c0] - H0.bx, ly = (long double)pd[c0 + 1] - H0.by, lz = (long double)pd[c0 + 2] - H0.bz;
          out[c0] = (double)((long double)H0.rc * lx + (long double)H0.rs * lz);
          out[c0 + 1] = (double)ly;
          out[c0 + 2] = (double)((long double)H0.rc * lz - (long double)H0.rs * lx);
        }
      };
      DGroup G;
      DGroup64 H;
      std::memset(&G, 0, sizeof G);
      G.chain = chain; G.first_prim = (int)prims.size(); G.bvh_root = -1;
      chain_motion(chain, G, H);
      // A sphere with the centre and radius of an EARLIER sphere of the same scan can never be the
      // closest hit: it computes the same two roots, its range is the earlier one's or narrower
      // (closest-so-far only shrinks), and base.clj:242,248 need root < t-max strictly, so it loses the
      // tie where the earlier one hit and misses where that one missed. (The random-spheres world lists
      // two coincident r = 1000 ground spheres, scenes.clj:52,64.) Such spheres keep their records at
      // the end of the group but are left out of its tested range. That holds for t-min >= 0 only: with a
      // first root below zero the duplicate's second root (-half_b + root1) / a can come out SMALLER
      // than the first and win, so the boundary groups of media, which are asked over (-inf, inf)
      // (constant_medium.clj:12), keep their duplicates.
      size_t n_tested = list.size();
      if (drop_shadowed_spheres) {
        std::map<std::array<double, 4>, int> seen;
        std::vector<Entry> kept, shadowed;
        for (Entry e : list) {
          const int pi = e_prim(e);
          bool dup = false;
          if (d->prim_type[pi] == RT_PRIM_SPHERE) {
            double f[9];
            sphere_fields(e, f);  // (the fields the device tests: the same numbers give the same roots)
            dup = !seen.emplace(std::array<double, 4>{f[0], f[1], f[2], f[3]}, pi).second;
          }
          (dup ? shadowed : kept).push_back(e);
        }
        n_tested = kept.size();
        n_shadowed_total += (int)shadowed.size();
        list = kept;
        list.insert(list.end(), shadowed.begin(), shadowed.end());
      }
      // Groups that get a tree: six rects that are a box with its lid sunk below the rim (RT_PRIM_LIDBOX, rt_scene.h)
      // become ONE leaf. The rects are found by exact comparison of their fields, all of one material; they move
      // behind everything the group tests (their records stay, the leaf names the one that was hit), in the order
      // they had, so ties among them fall as before. A box is left alone when one of its rects lies in the plane of
      // a rect that stays in the tree and overlaps it: the later of the two in the scan must win that tie, and behind
      // the others the box's rect would always be the later one.
      struct LidBox { float b[6]; float lid; int pos[6]; int mat; };  // pos: places in `list` of the faces z0 z1 y0 x0 x1 and of the lid
      std::vector<LidBox> lidboxes;
      std::vector<char> absorbed(list.size(), 0);
      if ((int)n_tested > RT_BRUTE_MAX && lid_boxes_as_leaves) {
        auto RF = [&](int pos, int k) { return (float)d->prim_d[(size_t)e_prim(list[(size_t)pos]) * RT_PRIM_STRIDE + k]; };
        auto TY = [&](int pos) { return d->prim_type[e_prim(list[(size_t)pos])]; };
        auto MT = [&](int pos) { return d->prim_mat[e_prim(list[(size_t)pos])]; };
        typedef std::array<float, 6> Key;  // type, a0 a1 b0 b1, material
        std::map<Key, std::vector<int>> by_shape;
        for (int pos = 0; pos < (int)n_tested; ++pos)
          if (TY(pos) >= RT_PRIM_XY_RECT) by_shape[Key{(float)TY(pos), RF(pos, 0), RF(pos, 1), RF(pos, 2), RF(pos, 3), (float)MT(pos)}].push_back(pos);
        auto find_k = [&](int type, float a0, float a1, float b0, float b1, int mat, float k_lo, float k_hi, int avoid) {
          // a free rect of that shape whose plane lies in [k_lo, k_hi]
          auto it = by_shape.find(Key{(float)type, a0, a1, b0, b1, (float)mat});
          if (it == by_shape.end()) return -1;
          for (int pos : it->second)
            if (!absorbed[(size_t)pos] && pos != avoid && RF(pos, 4) >= k_lo && RF(pos, 4) <= k_hi) return pos;
          return -1;
        };
        for (int p = 0; p < (int)n_tested; ++p) {
          if (TY(p) != RT_PRIM_XY_RECT || absorbed[(size_t)p]) continue;
          const float x0 = RF(p, 0), x1 = RF(p, 1), y0 = RF(p, 2), y1 = RF(p, 3), za = RF(p, 4);
          const int mat = MT(p);
          if (!(x0 < x1 && y0 < y1)) continue;
          auto it = by_shape.find(Key{(float)RT_PRIM_XY_RECT, x0, x1, y0, y1, (float)mat});
          for (int q : it->second) {
            if (q == p || absorbed[(size_t)q] || RF(q, 4) == za) continue;
            const float z0 = std::min(za, RF(q, 4)), z1 = std::max(za, RF(q, 4));
            const int fz0 = RF(p, 4) == z0 ? p : q, fz1 = RF(p, 4) == z0 ? q : p;
            const int fx0 = find_k(RT_PRIM_YZ_RECT, y0, y1, z0, z1, mat, x0, x0, -1);
            const int fx1 = find_k(RT_PRIM_YZ_RECT, y0, y1, z0, z1, mat, x1, x1, fx0);
            const int fy0 = find_k(RT_PRIM_XZ_RECT, x0, x1, z0, z1, mat, y0, y0, -1);
            const int lid = find_k(RT_PRIM_XZ_RECT, x0, x1, z0, z1, mat, std::nextafter(y0, y1), y1, fy0);
            if (fx0 < 0 || fx1 < 0 || fy0 < 0 || lid < 0) continue;
            LidBox B{{x0, x1, y0, y1, z0, z1}, RF(lid, 4), {fz0, fz1, fy0, fx0, fx1, lid}, mat};
            for (int f : B.pos) absorbed[(size_t)f] = 1;
            lidboxes.push_back(B);
            break;
          }
        }
        // give up the boxes that share a plane with a rect that stays in the tree
        for (bool changed = true; changed;) {
          changed = false;
          for (size_t bi = 0; bi < lidboxes.size() && !changed; ++bi)
            for (int f : lidboxes[bi].pos) {
              auto planes_meet = [&](int a, int b) {
                return TY(a) == TY(b) && RF(a, 4) == RF(b, 4) && RF(a, 0) < RF(b, 1) && RF(b, 0) < RF(a, 1) && RF(a, 2) < RF(b, 3) && RF(b, 2) < RF(a, 3);
              };
              bool meets = false;
              for (int pos = 0; pos < (int)n_tested && !meets; ++pos)
                meets = !absorbed[(size_t)pos] && TY(pos) >= RT_PRIM_XY_RECT && planes_meet(f, pos);
              if (meets) {
                for (int g : lidboxes[bi].pos) absorbed[(size_t)g] = 0;
                lidboxes.erase(lidboxes.begin() + (long)bi);
                changed = true;
                break;
              }
            }
        }
      }
      if (!lidboxes.empty() && (int)(n_tested - 5 * lidboxes.size()) <= RT_BRUTE_MAX) {
        // too few leaves left for a tree: the brute-force loops know nothing of box leaves
        lidboxes.clear();
        std::fill(absorbed.begin(), absorbed.end(), 0);
      }
      // the order on the device: what the group tests (its rects by type, its spheres, its box leaves), then
      // what only keeps a record (shadowed spheres, the rects of the box leaves)
      std::vector<int> order;        // a place in `list`, or -1 - b for box leaf b
      std::vector<int> new_place(list.size(), -1);
      for (int pos = 0; pos < (int)n_tested; ++pos)
        if (!absorbed[(size_t)pos]) { new_place[(size_t)pos] = (int)order.size(); order.push_back(pos); }
      for (int b = 0; b < (int)lidboxes.size(); ++b) order.push_back(-1 - b);
      const size_t n_tested_dev = order.size();
      for (int pos = (int)n_tested; pos < (int)list.size(); ++pos) { new_place[(size_t)pos] = (int)order.size(); order.push_back(pos); }
      for (int pos = 0; pos < (int)n_tested; ++pos)
        if (absorbed[(size_t)pos]) { new_place[(size_t)pos] = (int)order.size(); order.push_back(pos); }
      size_t li = 0;
      for (int place : order) {
        const bool tested = li++ < n_tested_dev;
        if (place < 0) {  // a box leaf
          const LidBox& B = lidboxes[(size_t)(-1 - place)];
          DPrim P;
          std::memset(&P, 0, sizeof P);
          P.type = RT_PRIM_LIDBOX | ((int)groups.size() << 4);
          P.mat = B.mat;
          for (int k = 0; k < 6; ++k) P.f[k] = B.b[k];
          P.f[6] = B.lid;
          for (int k = 0; k < 6; ++k) {
            const int idx = G.first_prim + new_place[(size_t)B.pos[k]];
            std::memcpy(&P.f[7 + k], &idx, 4);
          }
          G.n_lidbox++;
          features |= F_LIDBOX;
          prims.push_back(P);
          prims_rec.push_back(P);
          continue;
        }
        const Entry e = list[(size_t)place];
        const int pi = e_prim(e);
        DPrim P;
        std::memset(&P, 0, sizeof P);
        const int ptype = d->prim_type[pi];
        P.type = ptype | ((int)groups.size() << 4);
        P.mat = d->prim_mat[pi];
        const double* pd = d->prim_d + (size_t)pi * RT_PRIM_STRIDE;
        if (ptype <= RT_PRIM_MOVING_SPHERE) { features |= F_SPHERES; G.n_sph += tested ? 1 : 0; }
        else features |= F_RECTS;
        if (ptype <= RT_PRIM_MOVING_SPHERE && chain != 0) G.flags |= DGROUP_SPHERES_UNDER_CHAIN;
        if (ptype == RT_PRIM_XY_RECT && tested) G.n_xy++;
        if (ptype == RT_PRIM_XZ_RECT && tested) G.n_xz++;
        if (ptype == RT_PRIM_YZ_RECT && tested) G.n_yz++;
        DPrim L = P;  // the record's copy: the primitive in its chain's space
        if (ptype == RT_PRIM_SPHERE) {
          for (int k = 0; k < 4; ++k) L.d[k] = pd[k];
          L.d[4] = 1.0 / pd[3];  // base.clj:244 divides by the radius: the correctly rounded quotient, formed once
          sphere_fields(e, P.d);
          P.d[4] = L.d[4];
        } else if (ptype == RT_PRIM_MOVING_SPHERE) {
          for (int k = 0; k < 9; ++k) L.d[k] = pd[k];
          sphere_fields(e, P.d);
          any_moving = true;
          features |= F_MOVING;
          tlo = std::min(tlo, std::min(pd[6], pd[7]));
          thi = std::max(thi, std::max(pd[6], pd[7]));
        } else {
          for (int k = 0; k < 5; ++k) P.f[k] = (float)pd[k];
          L = P;
        }
        if (e_chain(e) != 0) world_spheres.push_back({prims.size(), e_chain(e)});
        prims.push_back(P);
        prims_rec.push_back(L);
      }
      G.first_rect = (int)(brute_rects.size() / RT_PAIR_STRIDE);
      G.first_box = (int)(brute_boxes.size() / RT_BOX_STRIDE);
      if ((int)n_tested <= RT_BRUTE_MAX) {
        // Six rects that close an axis-aligned box with one material (what box.clj:7-20 builds) are
        // intersected as one slab test; they keep their primitive records and indices.
        const int o_xy = G.first_prim, o_xz = o_xy + G.n_xy, o_yz = o_xz + G.n_xz, o_end = o_yz + G.n_yz;
        std::vector<char> boxed((size_t)(o_end - G.first_prim), 0);
        auto F5 = [&](int i) { return prims[(size_t)i].f; };  // a0 a1 b0 b1 k
        auto used = [&](int i) -> char& { return boxed[(size_t)(i - G.first_prim)]; };
        auto same_face = [&](int i, int j) {
          const float *a = F5(i), *b = F5(j);
          return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3] && prims[(size_t)i].mat == prims[(size_t)j].mat;
        };
        struct BoxCand { float bounds[6]; int face[6]; bool open; };
        std::vector<BoxCand> cands;
        auto take = [&](float x0, float x1, float y0, float y1, float z0, float z1, const int face[6], bool open) {
          BoxCand c{{x0, x1, y0, y1, z0, z1}, {face[0], face[1], face[2], face[3], face[4], face[5]}, open};
          for (int i : c.face)
            if (i >= 0) used(i) = 1;
          cands.push_back(c);
        };
        for (int p = o_xy; p < o_xz; ++p) {
          for (int q = p + 1; q < o_xz && !used(p); ++q) {
            if (used(q) || !same_face(p, q) || F5(p)[4] == F5(q)[4]) continue;
            const float x0 = F5(p)[0], x1 = F5(p)[1], y0 = F5(p)[2], y1 = F5(p)[3];
            const float z0 = std::min(F5(p)[4], F5(q)[4]), z1 = std::max(F5(p)[4], F5(q)[4]);
            if (!(x0 < x1 && y0 < y1)) continue;
            // the XZ pair at y0, y1 over [x0,x1] x [z0,z1] and the YZ pair at x0, x1 over [y0,y1] x [z0,z1]
            auto find_pair = [&](int lo, int hi, float a0, float a1, float b0, float b1, float k0, float k1, int out[2]) {
              out[0] = out[1] = -1;
              for (int i = lo; i < hi; ++i) {
                if (used(i) || prims[(size_t)i].mat != prims[(size_t)p].mat) continue;
                const float* f = F5(i);
                if (f[0] != a0 || f[1] != a1 || f[2] != b0 || f[3] != b1) continue;
                if (f[4] == k0 && out[0] < 0) out[0] = i;
                else if (f[4] == k1 && out[1] < 0) out[1] = i;
              }
              return out[0] >= 0 && out[1] >= 0;
            };
            int fy[2], fx[2];
            if (!find_pair(o_xz, o_yz, x0, x1, z0, z1, y0, y1, fy)) continue;
            if (!find_pair(o_yz, o_end, y0, y1, z0, z1, x0, x1, fx)) continue;
            const int face[6] = {F5(p)[4] == z0 ? p : q, F5(p)[4] == z0 ? q : p, fy[0], fy[1], fx[0], fx[1]};
            take(x0, x1, y0, y1, z0, z1, face, false);
          }
        }
        // Five rects that are faces of ONE axis-aligned box, the room of the Cornell worlds
        // (scenes.clj:76-81), whatever their materials: the same slab test with the missing face
        // marked -1. Every XY rect left is tried as a z face; the z extent comes from an XZ or YZ
        // rect that spans the XY rect's x or y extent and starts or ends on its plane.
        for (int p = o_xy; p < o_xz; ++p) {
          if (used(p)) continue;
          const float x0 = F5(p)[0], x1 = F5(p)[1], y0 = F5(p)[2], y1 = F5(p)[3], kz = F5(p)[4];
          if (!(x0 < x1 && y0 < y1)) continue;
          std::vector<std::array<float, 2>> z_extents;
          auto offer = [&](int lo, int hi, float a0, float a1) {
            for (int i = lo; i < hi; ++i) {
              const float* f = F5(i);
              if (used(i) || f[0] != a0 || f[1] != a1 || !(f[2] < f[3]) || (f[2] != kz && f[3] != kz)) continue;
              z_extents.push_back({f[2], f[3]});
            }
          };
          offer(o_xz, o_yz, x0, x1);
          offer(o_yz, o_end, y0, y1);
          for (const auto& ze : z_extents) {
            const float z0 = ze[0], z1 = ze[1];
            auto find_face = [&](int lo, int hi, float a0, float a1, float b0, float b1, float k, int taken) {
              for (int i = lo; i < hi; ++i) {
                const float* f = F5(i);
                if (!used(i) && i != taken && f[0] == a0 && f[1] == a1 && f[2] == b0 && f[3] == b1 && f[4] == k) return i;
              }
              return -1;
            };
            int face[6];
            face[0] = find_face(o_xy, o_xz, x0, x1, y0, y1, z0, -1);
            face[1] = find_face(o_xy, o_xz, x0, x1, y0, y1, z1, face[0]);
            face[2] = find_face(o_xz, o_yz, x0, x1, z0, z1, y0, -1);
            face[3] = find_face(o_xz, o_yz, x0, x1, z0, z1, y1, face[2]);
            face[4] = find_face(o_yz, o_end, y0, y1, z0, z1, x0, -1);
            face[5] = find_face(o_yz, o_end, y0, y1, z0, z1, x1, face[4]);
            int n_faces = 0;
            for (int i : face) n_faces += i >= 0;
            if (n_faces != 5) continue;
            take(x0, x1, y0, y1, z0, z1, face, true);
            break;
          }
        }
        // The device tests loose rects first, then the closed boxes, then the open ones. That differs from
        // the scan only where two rects can tie: same orientation, same plane, overlapping. The later
        // one in the scan must win (rectangles.clj accepts t = t-max), so the earlier one has to be tested
        // first; a box whose face would be tested AFTER a coinciding later rect is given up (its rects
        // go back to the loose ones, which are tested in scan order and before every box).
        auto coincide = [&](int i, int j) {
          const float *a = F5(i), *b = F5(j);
          return a[4] == b[4] && a[0] < b[1] && b[0] < a[1] && a[2] < b[3] && b[2] < a[3];
        };
        for (bool changed = true; changed;) {
          changed = false;
          std::stable_partition(cands.begin(), cands.end(), [](const BoxCand& c) { return !c.open; });
          std::vector<int> place((size_t)(o_end - G.first_prim), 0);  // 0 loose, 1 + index of its box
          for (size_t c = 0; c < cands.size(); ++c)
            for (int i : cands[c].face)
              if (i >= 0) place[(size_t)(i - G.first_prim)] = 1 + (int)c;
          const int lo[3] = {o_xy, o_xz, o_yz}, hi[3] = {o_xz, o_yz, o_end};
          for (int t = 0; t < 3 && !changed; ++t)
            for (int i = lo[t]; i < hi[t] && !changed; ++i)
              for (int j = i + 1; j < hi[t] && !changed; ++j) {
                const int pi_ = place[(size_t)(i - G.first_prim)], pj = place[(size_t)(j - G.first_prim)];
                if (pi_ > pj && coincide(i, j)) {  // i is earlier in the scan and would be tested later
                  for (int f : cands[(size_t)pi_ - 1].face)
                    if (f >= 0) used(f) = 0;
                  cands.erase(cands.begin() + (pi_ - 1));
                  changed = true;
                }
              }
        }
        int n_box = 0, n_open = 0;
        for (const BoxCand& c : cands) {
          for (float v : c.bounds) brute_boxes.push_back(v);
          for (int i : c.face) {
            float bits;
            std::memcpy(&bits, &i, 4);
            brute_boxes.push_back(bits);
          }
          (c.open ? n_open : n_box) += 1;
        }
        G.flags |= n_open << DGROUP_OPEN_SHIFT;
        // pair records per orientation (see DScene.brute_rects) for the rects no box took
        int off = G.first_prim, loose_n[3] = {0, 0, 0}, ti = 0;
        for (int n_t : {G.n_xy, G.n_xz, G.n_yz}) {
          std::vector<int> loose;
          for (int i = off; i < off + n_t; ++i)
            if (!used(i)) loose.push_back(i);
          const int n_l = (int)loose.size(), odd = n_l & 1;
          for (int p = 0; p < (n_l + 1) / 2; ++p) {
            const int j1 = 2 * p + 1 - odd;
            const int i1 = loose[(size_t)j1], i0 = loose[(size_t)std::max(j1 - 1, 0)];
            const float* f0 = prims[(size_t)i0].f;
            const float* f1 = prims[(size_t)i1].f;
            // DPrim.f = a0 a1 b0 b1 k
            for (int k : {4, 0, 1, 2, 3}) { brute_rects.push_back(f0[k]); brute_rects.push_back(f1[k]); }
            float bits[2];
            std::memcpy(&bits[0], &i0, 4);
            std::memcpy(&bits[1], &i1, 4);
            brute_rects.push_back(bits[0]);
            brute_rects.push_back(bits[1]);
          }
          loose_n[ti++] = n_l;
          off += n_t;
        }
        G.loose = loose_n[0] | (loose_n[1] << 8) | (loose_n[2] << 16) | (n_box << 24);
      }
      groups.push_back(G);
      groups64.push_back(H);
    }
  };
  emit_groups(F.surface);
  int n_surface_groups = (int)groups.size();
  spheres_to_world = false;  // boundaries of media keep their spheres in the chain's space
  {  // one header group per chain that lent spheres to the world-space groups, behind the tested ones
    std::map<int, int> header;
    for (const auto& ws : world_spheres) {
      auto it = header.find(ws.second);
      if (it == header.end()) {
        DGroup G;
        DGroup64 H;
        std::memset(&G, 0, sizeof G);
        G.chain = ws.second; G.first_prim = (int)prims.size(); G.bvh_root = -1;
        G.flags = DGROUP_SPHERES_UNDER_CHAIN;
        chain_motion(ws.second, G, H);
        it = header.emplace(ws.second, (int)groups.size()).first;
        groups.push_back(G);
        groups64.push_back(H);
      }
      prims[ws.first].type = (prims[ws.first].type & 15) | (it->second << 4);
      prims_rec[ws.first].type = prims[ws.first].type;
    }
  }
  if (F.chains.size() > (1u << 20)) return fail(RT_ERR_UNSUPPORTED, "too many distinct instance chains");
  std::vector<DMedium> media((size_t)d->n_media);
  for (int m = 0; m < d->n_media; ++m) {
    HostMedium& M = F.media[m];
    DMedium dm;
    std::memset(&dm, 0, sizeof dm);
    dm.first_group = (int)groups.size();
    if (M.present) {
      drop_shadowed_spheres = false;
      emit_groups(M.boundary);
      drop_shadowed_spheres = true;
      dm.outer_chain = M.outer_chain;
      dm.phase_mat = M.phase_mat;
      dm.neg_inv_density = M.neg_inv_density;
    }
    dm.n_groups = (int)groups.size() - dm.first_group;  // 0 for a medium the root never reaches
    if (dm.n_groups > 0) {
      // the two boundary shapes medium_t answers from one intersection (every medium the reference
      // ships); anything else needs an instantiation with the general boundary loops
      const DGroup& B = groups[(size_t)dm.first_group];
      const int n_rects = B.n_xy + B.n_xz + B.n_yz;
      const bool one_sphere = dm.n_groups == 1 && B.n_sph == 1 && n_rects == 0;
      const bool one_box = dm.n_groups == 1 && B.n_sph == 0 && n_rects <= RT_BRUTE_MAX && B.loose == (1 << 24);
      if (!one_sphere && !one_box) features |= F_MEDIA_ANY;
    }
    {  // an empty group that carries the motion of the chain the medium itself lives under
      DGroup G;
      DGroup64 H;
      std::memset(&G, 0, sizeof G);
      G.chain = dm.outer_chain; G.first_prim = (int)prims.size(); G.bvh_root = -1;
      chain_motion(dm.outer_chain, G, H);
      dm.outer_group = (int)groups.size();
      groups.push_back(G);
      groups64.push_back(H);
    }
    media[m] = dm;
  }

  std::vector<DChainRec> chain_recs;
  struct ChainD { double A, B, t[3], A1, B1, s0, c0, s01, c01; };  // the same in FP64, for the per-rect constants of DPrimInfo
  std::vector<ChainD> chain_d;
  for (auto& c : F.chains) {
    // the record side, composed in FP64 from the innermost wrapper outwards (see DChainRec)
    DChainRec rec;
    std::memset(&rec, 0, sizeof rec);
    rec.n_ops = (int)c.size();
    rec.inner = c.size() >= 2 ? 1 : 0;
    double A = 1.0, B = 0.0, t[3] = {0.0, 0.0, 0.0};
    rec.A1 = 1.f; rec.B1 = 0.f;
    double A1d = 1.0, B1d = 0.0;
    for (size_t i = c.size(); i-- > 0;) {
      if (c[i].kind == DOP_TRANSLATE) {
        t[0] += c[i].da; t[1] += c[i].db; t[2] += c[i].dc;
      } else {
        const double so = c[i].da, co = c[i].db;
        const double nA = co * A + so * B, nB = co * B + so * A;
        const double nx = co * t[0] + so * t[2], nz = so * t[0] + co * t[2];
        A = nA; B = nB; t[0] = nx; t[2] = nz;
        ++rec.n_rotate;
      }
      if (i == 1) { rec.A1 = (float)A; rec.B1 = (float)B; A1d = A; B1d = B; }  // everything but the outermost
    }
    rec.A = (float)A; rec.B = (float)B; rec.tx = (float)t[0]; rec.ty = (float)t[1]; rec.tz = (float)t[2];
    // the ray side's turn (rotate.clj:14-24, angles add) after the outermost and after the two outermost
    double s0 = 0.0, c0 = 1.0;
    if (!c.empty() && c[0].kind == DOP_ROTATE_Y) { s0 = c[0].da; c0 = c[0].db; }
    double s01 = s0, c01 = c0;
    if (c.size() >= 2 && c[1].kind == DOP_ROTATE_Y) {
      const double so = c[1].da, co = c[1].db;
      c01 = co * c0 - so * s0; s01 = so * c0 + co * s0;
    }
    rec.s0 = (float)s0; rec.c0 = (float)c0; rec.s01 = (float)s01; rec.c01 = (float)c01;
    chain_recs.push_back(rec);
    chain_d.push_back(ChainD{A, B, {t[0], t[1], t[2]}, A1d, B1d, s0, c0, s01, c01});
  }

  std::vector<DMat> mats((size_t)d->n_mats);
  for (int i = 0; i < d->n_mats; ++i) {
    const double* md = d->mat_d + (size_t)i * RT_MAT_DSTRIDE;
    DMat M;
    std::memset(&M, 0, sizeof M);
    M.type = d->mat_type[i];
    M.tex = d->mat_tex[i];
    M.flags = tex_needs_uv(d, d->mat_tex[i]) ? DMAT_NEEDS_UV : 0;
    M.r = (float)md[0]; M.g = (float)md[1]; M.b = (float)md[2]; M.x = (float)md[3];
    const bool textured = M.type == RT_MAT_LAMBERTIAN || M.type == RT_MAT_DIFFUSE_LIGHT || M.type == RT_MAT_ISOTROPIC;
    if (textured) {
      if (d->tex_type[M.tex] == RT_TEX_SOLID) {
        M.flags |= DMAT_SOLID;
        const double* td = d->tex_d + (size_t)M.tex * RT_TEX_DSTRIDE;
        M.r = (float)td[0]; M.g = (float)td[1]; M.b = (float)td[2];
      } else {
        features |= F_TEXTURES;
      }
    }
    if (M.type == RT_MAT_METAL || M.type == RT_MAT_DIELECTRIC) features |= F_SPECULAR;
    if (M.type == RT_MAT_ISOTROPIC) features |= F_MEDIA;
    mats[i] = M;
  }
  if (features & F_TEXTURES)
    for (int i = 0; i < d->n_tex; ++i)
      features |= d->tex_type[i] == RT_TEX_NOISE ? F_NOISE : d->tex_type[i] == RT_TEX_IMAGE ? F_IMAGE : 0u;
  std::vector<DTex> texs((size_t)d->n_tex);
  for (int i = 0; i < d->n_tex; ++i) {
    DTex T;
    std::memset(&T, 0, sizeof T);
    T.type = d->tex_type[i];
    T.i0 = d->tex_i[(size_t)i * RT_TEX_ISTRIDE];
    T.i1 = d->tex_i[(size_t)i * RT_TEX_ISTRIDE + 1];
    T.r = (float)d->tex_d[(size_t)i * RT_TEX_DSTRIDE];
    T.g = (float)d->tex_d[(size_t)i * RT_TEX_DSTRIDE + 1];
    T.b = (float)d->tex_d[(size_t)i * RT_TEX_DSTRIDE + 2];
    texs[i] = T;
  }
  if (d->n_perlin > RT_MAX_PERLIN_LDS)
    return fail(RT_ERR_UNSUPPORTED, "%d Perlin table sets; at most %d fit the LDS budget", d->n_perlin, RT_MAX_PERLIN_LDS);
  std::vector<float> pvec((size_t)d->n_perlin * 256 * 4, 0.f);
  std::vector<uint8_t> pperm((size_t)d->n_perlin * 768, 0);
  for (int i = 0; i < d->n_perlin * 256; ++i)
    for (int k = 0; k < 3; ++k) pvec[(size_t)i * 4 + k] = (float)d->perlin_ranvec[(size_t)i * 3 + k];
  for (int i = 0; i < d->n_perlin * 768; ++i) pperm[i] = (uint8_t)d->perlin_perm[i];
  std::vector<DImage> images((size_t)d->n_images);
  std::vector<uint32_t> texels;
  for (int i = 0; i < d->n_images; ++i) {
    DImage I{d->img_w[i], d->img_h[i], (int)texels.size(), 0};
    size_t n = (size_t)I.w * I.h;
    if (texels.size() + n > 0x7fffffffull) return fail(RT_ERR_UNSUPPORTED, "image textures exceed 2^31 texels");
    const uint8_t* src = d->img_rgb8[i];
    for (size_t t = 0; t < n; ++t)
      texels.push_back((uint32_t)src[3 * t] | ((uint32_t)src[3 * t + 1] << 8) | ((uint32_t)src[3 * t + 2] << 16));
    images[i] = I;
  }
  for (DTex& T : texs)
    if (T.type == RT_TEX_IMAGE && T.i0 >= 0) { T.i1 = images[(size_t)T.i0].first_texel; T.w = images[(size_t)T.i0].w; T.h = images[(size_t)T.i0].h; }

  // per-primitive record / scatter info (one lookup for the winner of a traversal)
  std::vector<DPrimInfo> infos(prims.size());
  for (size_t i = 0; i < prims.size(); ++i) {
    const DPrim& P = prims[i];
    const DGroup& G = groups[(size_t)(P.type >> 4)];
    const DMat& M = mats[(size_t)P.mat];
    DPrimInfo I;
    std::memset(&I, 0, sizeof I);
    I.type_group = P.type; I.mat = P.mat; I.chain = G.chain; I.gflags = G.flags;
    if ((P.type & 15) >= RT_PRIM_XY_RECT) {
      I.a0 = P.f[0]; I.a1 = P.f[1]; I.b0 = P.f[2]; I.b1 = P.f[3]; I.k = P.f[4];
      I.ox = (P.type & 15) == RT_PRIM_YZ_RECT ? 1.f : 0.f;
      I.oy = (P.type & 15) == RT_PRIM_XZ_RECT ? 1.f : 0.f;
      I.oz = (P.type & 15) == RT_PRIM_XY_RECT ? 1.f : 0.f;
    }
    I.rc = G.rc; I.rs = G.rs; I.bx = G.bx; I.by = G.by; I.bz = G.bz;
    if ((P.type & 15) >= RT_PRIM_XY_RECT && (P.type & 15) <= RT_PRIM_YZ_RECT && G.chain > 0) {
      const ChainD& C = chain_d[(size_t)G.chain];
      const double o[3] = {I.ox, I.oy, I.oz};
      // the normal inside the outermost wrapper, then outside it (rotate.clj:37-42 composed: x' = A x + B z, z' = B x + A z)
      const double n1[3] = {C.A1 * o[0] + C.B1 * o[2], o[1], C.B1 * o[0] + C.A1 * o[2]};
      const double nn[3] = {C.c0 * n1[0] + C.s0 * n1[2], n1[1], C.s0 * n1[0] + C.c0 * n1[2]};
      // dot(turn(d), n) = dot(d, turn^T(n)), turn(d) = (c dx - s dz, dy, s dx + c dz) (rotate.clj:19-27)
      const double g1[3] = {C.c01 * n1[0] + C.s01 * n1[2], n1[1], -C.s01 * n1[0] + C.c01 * n1[2]};
      const double g0[3] = {C.c0 * nn[0] + C.s0 * nn[2], nn[1], -C.s0 * nn[0] + C.c0 * nn[2]};
      I.fA = (float)C.A; I.fB = (float)C.B; I.ftx = (float)C.t[0]; I.fty = (float)C.t[1]; I.ftz = (float)C.t[2];
      I.fnx = (float)nn[0]; I.fny = (float)nn[1]; I.fnz = (float)nn[2];
      I.g0x = (float)g0[0]; I.g0y = (float)g0[1]; I.g0z = (float)g0[2];
      I.g1x = (float)g1[0]; I.g1y = (float)g1[1]; I.g1z = (float)g1[2];
      I.finner = chain_recs[(size_t)G.chain].inner;
    }
    I.mtype = M.type; I.mflags = M.flags; I.mtex = M.tex; I.r = M.r; I.g = M.g; I.b = M.b; I.x = M.x;
    infos[i] = I;
  }

  rt_scene* sc = new (std::nothrow) rt_scene();
  if (!sc) return fail(RT_ERR_OOM, "out of host memory");
  sc->ctx = ctx;
  sc->n_media = d->n_media;
  sc->has_time_bounds = any_moving;
  sc->n_shadowed_spheres = n_shadowed_total;
  sc->time_lo = (float)tlo;
  sc->time_hi = (float)thi;
  struct Guard {
    rt_scene* s;
    ~Guard() { if (s) rt_scene_destroy(s); }
  } guard{sc};

  DScene& S = sc->S;
  const DPrim* d_prims = nullptr;
  const size_t n_real_prims = prims.size();
  {  // the brute-force loops read primitive records four at a time: keep that much slack
    DPrim zero;
    std::memset(&zero, 0, sizeof zero);
    for (int k = 0; k < 4; ++k) prims.push_back(zero);
  }
  if ((rc = upload(sc, prims, &d_prims))) return rc;
  prims.resize(n_real_prims);
  S.prims = d_prims;
  S.prims_rec = d_prims;
  if (!world_spheres.empty() && (rc = upload(sc, prims_rec, &S.prims_rec))) return rc;

  // BVHs for the large groups, built on the device: surface-area sweep up to RT_SWEEP_MAX primitives,
  // Morton order beyond (or whichever builder the caller names)
  int n_bvh_nodes = 0;
  auto group_size = [](const DGroup& G) { return G.n_xy + G.n_xz + G.n_yz + G.n_sph + G.n_lidbox; };
  for (auto& G : groups)
    if (group_size(G) > RT_BRUTE_MAX) n_bvh_nodes += group_size(G) - 1;
  DBvhNode* d_nodes = nullptr;
  {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, sizeof(DBvhNode) * (size_t)std::max(n_bvh_nodes, 1)));
    sc->allocs.push_back(p);
    HIPCHK(hipMemset(p, 0, sizeof(DBvhNode) * (size_t)std::max(n_bvh_nodes, 1)));
    d_nodes = static_cast<DBvhNode*>(p);
  }
  if (n_bvh_nodes && n_real_prims > (size_t)RT_LEAF_INDEX_MASK)
    return fail(RT_ERR_UNSUPPORTED, "%zu primitives; a world with trees holds at most %d", n_real_prims, RT_LEAF_INDEX_MASK);
  int node_base = 0, max_height = 0;
  for (auto& G : groups) {
    const int n_prims = group_size(G);
    if (n_prims <= RT_BRUTE_MAX) continue;
    int height = 0;
    int builder = bvh_builder;
    if (builder == RT_BVH_HOST_SAH && n_prims > RT_SAH_MAX) builder = RT_BVH_DEFAULT;
    if (builder == RT_BVH_DEFAULT) builder = n_prims <= RT_SWEEP_MAX ? RT_BVH_DEVICE_SAH : RT_BVH_DEVICE_LBVH;
    hipError_t e =
        builder == RT_BVH_HOST_SAH
            ? build_bvh_sah(prims.data(), G.first_prim, n_prims, d_nodes, node_base, sc->time_lo, sc->time_hi, &height, 0)
        : builder == RT_BVH_DEVICE_SAH
            ? build_bvh_sweep(d_prims, G.first_prim, n_prims, d_nodes, node_base, sc->time_lo, sc->time_hi, &height, 0)
            : build_bvh(d_prims, G.first_prim, n_prims, d_nodes, node_base, sc->time_lo, sc->time_hi, &height, 0);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "BVH build failed: %s", hipGetErrorString(e));
    G.bvh_root = node_base;
    node_base += n_prims - 1;
    max_height = std::max(max_height, height);
  }
  {  // the traversal reads the packed form
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, sizeof(DBvhNode16) * (size_t)std::max(n_bvh_nodes, 1)));
    sc->allocs.push_back(p);
    hipError_t e = pack_bvh_nodes(d_nodes, d_prims, static_cast<DBvhNode16*>(p), n_bvh_nodes, 0);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "BVH packing failed: %s", hipGetErrorString(e));
    S.nodes = static_cast<const DBvhNode16*>(p);
  }
  sc->d_tree = d_nodes;
  sc->n_tree_nodes = n_bvh_nodes;
  sc->tree_height = max_height;
  S.bvh_stack = n_bvh_nodes ? max_height + 1 : 0;
  S.n_bvh_nodes = n_bvh_nodes;
  // the second half of every packed node also lives in LDS when the trees are small enough for that not to
  // cost the kernel a resident workgroup (16 bytes per node and workgroup)
  S.xch_on_stacks = S.bvh_stack >= 4 ? 1 : 0;  // (256 words per wave = four levels of its 64 lanes' stacks)
  S.lds_nodes = n_bvh_nodes <= RT_LDS_NODE_MAX ? n_bvh_nodes : 0;
  {
    std::vector<int32_t> bvh_next(groups.size(), -1);
    int next = -1;
    for (int gi = n_surface_groups - 1; gi >= 0; --gi) {
      bvh_next[(size_t)gi] = next;
      if (groups[(size_t)gi].bvh_root >= 0) next = gi;
    }
    S.first_bvh_group = next;
    if ((rc = upload(sc, bvh_next, &S.bvh_next))) return rc;
  }
  if (n_bvh_nodes) features |= F_BVH;
  if (d->n_media) features |= F_MEDIA;
  if (chain_recs.size() > 1) features |= F_CHAINS;
  S.features = features;

  brute_rects.resize(brute_rects.size() + RT_PAIR_STRIDE * RT_PAIR_SLACK, 0.f);  // slack for batched reads
  if ((rc = upload(sc, brute_rects, &S.brute_rects))) return rc;
  brute_boxes.resize(brute_boxes.size() + RT_BOX_STRIDE, 0.f);  // slack
  if ((rc = upload(sc, brute_boxes, &S.brute_boxes))) return rc;
  if ((rc = upload(sc, infos, &S.prim_info))) return rc;
  S.n_prims = (int)prims.size();
  if ((rc = upload(sc, groups, &S.groups))) return rc;
  if ((rc = upload(sc, groups64, &S.groups64))) return rc;
  if ((rc = upload(sc, chain_recs, &S.chain_recs))) return rc;
  if ((rc = upload(sc, media, &S.media))) return rc;
  if ((rc = upload(sc, mats, &S.mats))) return rc;
  if ((rc = upload(sc, texs, &S.texs))) return rc;
  if ((rc = upload(sc, pvec, &S.perlin_vec))) return rc;
  if ((rc = upload(sc, pperm, &S.perlin_perm))) return rc;
  if ((rc = upload(sc, images, &S.images))) return rc;
  if ((rc = upload(sc, texels, &S.texels))) return rc;
  S.n_surface_groups = n_surface_groups;
  S.n_groups = (int)groups.size();
  S.n_chains = (int)chain_recs.size();
  S.n_media = d->n_media;
  S.n_perlin = d->n_perlin;

  // the largest instantiation's need bounds every kernel this scene may run on (hit and scatter hooks included)
  size_t lds = 0, lds_any = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    lds = shade_lds_bytes(S, RT_BLOCK);
    lds_any = lds;
    for (int v = 0; v < RT_N_VARIANTS; ++v)
      if (variant_covers(S, v)) lds_any = std::max(lds_any, variant_lds_bytes(S, RT_BLOCK, v));
    if (lds_any <= ctx->lds_max || S.lds_nodes == 0) break;
    S.lds_nodes = 0;  // a deep tree's stacks leave no room for the node halves: they are read from memory
  }
  if (lds_any > ctx->lds_max)
    return fail(RT_ERR_UNSUPPORTED, "scene needs %zu bytes of LDS per workgroup (BVH height %d); this device allows %zu",
                lds_any, max_height, ctx->lds_max);
  HIPCHK(allow_lds(lds_any));
  HIPCHK(render_occupancy(S, false, false, &sc->blocks_per_cu[0]));
  HIPCHK(render_occupancy(S, true, false, &sc->blocks_per_cu[1]));
  if (sc->blocks_per_cu[0] < 1 || sc->blocks_per_cu[1] < 1) return fail(RT_ERR_HIP, "path kernel does not fit a compute unit");
  // worlds with trees can render on the kernel that re-packs paths between the stages of a bounce (RT_FLAG_POOL_PATHS)
  sc->pool = pool_kernel_covers(S) && pool_lds_bytes(S, RT_BLOCK) <= ctx->lds_max;
  if (sc->pool) {
    HIPCHK(allow_lds(std::max(lds_any, pool_lds_bytes(S, RT_BLOCK))));
    HIPCHK(render_occupancy(S, false, true, &sc->pool_blocks_per_cu[0]));
    HIPCHK(render_occupancy(S, true, true, &sc->pool_blocks_per_cu[1]));
    if (sc->pool_blocks_per_cu[0] < 1 || sc->pool_blocks_per_cu[1] < 1) sc->pool = false;
  }

  sc->info.n_prims_flat = (int)prims.size();
  sc->info.n_groups = (int)groups.size();
  sc->info.n_chains = (int)chain_recs.size();
  sc->info.n_media = d->n_media;
  sc->info.n_bvh_nodes = n_bvh_nodes;
  sc->info.lds_bytes = (int)lds;
  guard.s = nullptr;
  *out = sc;
  return RT_OK;
}

int rt_scene_create(rt_ctx* ctx, const rt_scene_desc* d, rt_scene** out) {
  return guarded("rt_scene_create", [&] { return scene_create_impl(ctx, d, RT_BVH_DEFAULT, out); });
}
int rt_scene_bvh_digest(rt_scene* sc, uint64_t* digest, int32_t* n_nodes, int32_t* height) {
  return guarded("rt_scene_bvh_digest", [&]() -> int {
    if (!sc || !digest) return fail(RT_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(sc->ctx->device));
    std::vector<DBvhNode> nodes((size_t)sc->n_tree_nodes);
    if (sc->n_tree_nodes) HIPCHK(hipMemcpy(nodes.data(), sc->d_tree, sizeof(DBvhNode) * nodes.size(), hipMemcpyDeviceToHost));
    uint64_t h = 0xcbf29ce484222325ull;  // FNV-1a over the records' words
    for (const DBvhNode& N : nodes) {
      uint32_t w[14];
      memcpy(w + 0, N.lmin, 12); memcpy(w + 3, N.lmax, 12); memcpy(w + 6, N.rmin, 12); memcpy(w + 9, N.rmax, 12);
      memcpy(w + 12, &N.left, 4); memcpy(w + 13, &N.right, 4);
      for (uint32_t x : w) { h ^= x; h *= 0x100000001b3ull; }
    }
    *digest = h;
    if (n_nodes) *n_nodes = sc->n_tree_nodes;
    if (height) *height = sc->tree_height;
    return RT_OK;
  });
}

int rt_scene_create_with(rt_ctx* ctx, const rt_scene_desc* d, int32_t bvh_builder, rt_scene** out) {
  return guarded("rt_scene_create_with", [&] { return scene_create_impl(ctx, d, bvh_builder, out); });
}

int rt_scene_get_variants(const rt_scene* sc, int32_t* render_var, uint32_t* covering_mask) {
  if (!sc) return fail(RT_ERR_ARG, "scene is NULL");
  if (render_var) *render_var = render_variant(sc->S);
  if (covering_mask) {
    uint32_t m = 0;
    for (int v = 0; v < RT_N_VARIANTS; ++v)
      if (variant_covers(sc->S, v)) m |= 1u << v;
    *covering_mask = m;
  }
  return RT_OK;
}

int rt_get_device_info(const rt_ctx* ctx, rt_device_info* out) {
  if (!ctx || !out) return fail(RT_ERR_ARG, "ctx or out is NULL");
  out->n_cu = ctx->n_cu;
  out->clock_khz = ctx->clock_khz;
  out->lds_bytes_max = (int32_t)ctx->lds_max;
  out->timer_khz = ctx->wall_khz;
  return RT_OK;
}

int rt_scene_get_info(const rt_scene* sc, rt_scene_info* out) {
  if (!sc || !out) return fail(RT_ERR_ARG, "scene or out is NULL");
  *out = sc->info;
  return RT_OK;
}

// Enqueues one render on `stream`: clears the queue heads and counters, launches the path kernel over
// the persistent grid, resolves the partial sums. `queue`, `d_stats` and `partial` are the scratch
// buffers of this render (the context's own, or the file-scope array of the entry point that
// allocates nothing). Synchronises only when `stats` is asked for.
static int render_enqueue(rt_ctx* ctx, const rt_scene* sc, const DRender& R, const DCamera& C, bool count, bool pool, uint32_t* queue,
                          unsigned long long* d_stats, float* partial, float* d_rgb_sum, uint32_t* d_argb,
                          rt_stats* stats, hipStream_t stream) {
  HIPCHK(hipMemsetAsync(queue, 0, RT_QUEUE_HEADS * RT_QUEUE_STRIDE * 4, stream));
  HIPCHK(hipMemsetAsync(d_stats, 0, sizeof(unsigned long long) * RT_STATS_WORDS, stream));
  // persistent grid: as many workgroups as stay resident, never more than there is work for
  const int per_cu = pool ? sc->pool_blocks_per_cu[count ? 1 : 0] : sc->blocks_per_cu[count ? 1 : 0];
  int grid = ctx->n_cu * per_cu;
  int grid_needed = (R.n_items + RT_BLOCK - 1) / RT_BLOCK;
  grid = std::max(1, std::min(grid, grid_needed));
  if (stats) HIPCHK(hipEventRecord(ctx->ev0, stream));
  if (R.n_items > 0)
    HIPCHK(launch_render(sc->S, C, R, partial, queue, d_stats, count, pool, grid, stream));
  if (stats) HIPCHK(hipEventRecord(ctx->ev1, stream));
  if (d_rgb_sum || d_argb) HIPCHK(launch_resolve(R, partial, d_rgb_sum, d_argb, stream));
  if (stats) {
    unsigned long long h[RT_STATS_WORDS];
    HIPCHK(hipMemcpyAsync(h, d_stats, sizeof h, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    std::memset(stats, 0, sizeof *stats);
    stats->samples = h[0];
    stats->rays = h[1];
    for (int e = 0; e < RT_EV_COUNT; ++e) stats->events[e] = h[2 + e];
    stats->kernel_ms = ms;
    // cycle counter over constant-rate timer, both read by one wave at its start and at its end
    const unsigned long long cyc = h[RT_STATS_CLOCK + 1] - h[RT_STATS_CLOCK], tim = h[RT_STATS_CLOCK + 3] - h[RT_STATS_CLOCK + 2];
    if (R.n_items > 0 && tim > 0 && ctx->wall_khz > 0) stats->shader_mhz = (double)cyc / (double)tim * ctx->wall_khz * 1e-3;
  }
  return RT_OK;
}

static int render_device_impl(rt_ctx* ctx, const rt_scene* sc, const rt_camera* cam, const rt_params* p, float* d_rgb_sum,
                              uint32_t* d_argb, rt_stats* stats, void* hip_stream) {
  if (!ctx || !sc || sc->ctx != ctx) return fail(RT_ERR_ARG, "ctx/scene is NULL or the scene belongs to another context");
  DRender R;
  int rc = fill_render(sc, cam, p, R);
  if (rc) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  DCamera C;
  fill_camera(cam, p, C);
  const bool count = (p->flags & RT_FLAG_COUNTERS) != 0;
  const size_t need = (size_t)std::max(R.n_items, 1) * 3 * sizeof(float);  // one slot per work item
  if (ctx->partial_bytes < need) {
    // (the whole device: an earlier render on another stream may still be using the old slab)
    HIPCHK(hipDeviceSynchronize());
    if (ctx->d_partial) HIPCHK(hipFree(ctx->d_partial));
    ctx->d_partial = nullptr;
    ctx->partial_bytes = 0;
    HIPCHK(hipMalloc(&ctx->d_partial, need));
    ctx->partial_bytes = need;
  }
  return render_enqueue(ctx, sc, R, C, count, sc->pool && (p->flags & RT_FLAG_POOL_PATHS), ctx->d_queue, ctx->d_stats, ctx->d_partial, d_rgb_sum, d_argb, stats, stream);
}

// ---- the entry point that allocates nothing ------------------------------------------------
// Scratch of a render = the queue heads, the counters and the partial-sum slab (one 12-byte slot
// per work item). Here they live in this file-scope device array instead of in buffers the
// library allocates. Each GPU holds its own copy at its own address (looked up once per device in
// rt_init and kept in the context); every call on a GPU shares that GPU's copy.
#define RT_SCRATCH_BYTES (32u << 20)
#define RT_SCRATCH_QUEUE_OFF 0u       // RT_QUEUE_HEADS * RT_QUEUE_STRIDE * 4 = 1024 bytes
#define RT_SCRATCH_STATS_OFF 1024u    // RT_STATS_WORDS * 8 bytes
#define RT_SCRATCH_SLAB_OFF 4096u
static_assert(RT_QUEUE_HEADS * RT_QUEUE_STRIDE * 4 <= RT_SCRATCH_STATS_OFF, "queue heads overlap the counters");
static_assert(RT_SCRATCH_STATS_OFF + RT_STATS_WORDS * 8 <= RT_SCRATCH_SLAB_OFF, "counters overlap the slab");
static __device__ __align__(256) unsigned char scratch[RT_SCRATCH_BYTES];

// Called by rt_init with the context's device current.
static int lookup_scratch(rt_ctx* ctx) {
  void* ptr = nullptr;
  HIPCHK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
  if (!ptr) return fail(RT_ERR_HIP, "the scratch array has no address on device %d", ctx->device);
  ctx->scratch = static_cast<unsigned char*>(ptr);
  return RT_OK;
}

int rt_render_device_noalloc(rt_ctx* ctx, const rt_scene* sc, const rt_camera* cam, const rt_params* p, float* d_rgb_sum,
                             uint32_t* d_argb, void* hip_stream) {
  if (!ctx || !sc || sc->ctx != ctx) return fail(RT_ERR_ARG, "ctx/scene is NULL or the scene belongs to another context");
  if (!ctx->scratch) return fail(RT_ERR_HIP, "the scratch array was not found on device %d", ctx->device);
  DRender R;
  int rc = fill_render(sc, cam, p, R);
  if (rc) return rc;
  if (p->flags & RT_FLAG_COUNTERS) return fail(RT_ERR_UNSUPPORTED, "rt_render_device_noalloc returns no statistics: RT_FLAG_COUNTERS is refused");
  const unsigned long long need = (unsigned long long)R.n_items * 3ull * sizeof(float);
  const unsigned long long room = RT_SCRATCH_BYTES - RT_SCRATCH_SLAB_OFF;
  if (need > room)
    return fail(RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: %d work items (%d tiles x 64 pixels x %d sample chunks) need %llu "
                "bytes of partial sums, the fixed scratch holds %llu (at most %llu work items): render fewer pixels or samples "
                "per call, or use rt_render_device", R.n_items, R.n_owned_tiles, R.n_chunks, need, room, room / 12ull);
  HIPCHK(hipSetDevice(ctx->device));
  DCamera C;
  fill_camera(cam, p, C);
  unsigned char* base = ctx->scratch;
  return render_enqueue(ctx, sc, R, C, false, sc->pool && (p->flags & RT_FLAG_POOL_PATHS), reinterpret_cast<uint32_t*>(base + RT_SCRATCH_QUEUE_OFF),
                        reinterpret_cast<unsigned long long*>(base + RT_SCRATCH_STATS_OFF),
                        reinterpret_cast<float*>(base + RT_SCRATCH_SLAB_OFF), d_rgb_sum, d_argb, nullptr,
                        static_cast<hipStream_t>(hip_stream));
}

int rt_render_device(rt_ctx* ctx, const rt_scene* sc, const rt_camera* cam, const rt_params* p, float* d_rgb_sum,
                     uint32_t* d_argb, rt_stats* stats, void* hip_stream) {
  return guarded("rt_render_device", [&] { return render_device_impl(ctx, sc, cam, p, d_rgb_sum, d_argb, stats, hip_stream); });
}

int rt_render(rt_ctx* ctx, const rt_scene* sc, const rt_camera* cam, const rt_params* p, float* rgb_sum,
              uint32_t* argb, rt_stats* stats) {
  if (!ctx || !p) return fail(RT_ERR_ARG, "ctx or params is NULL");
  if (p->width <= 0 || p->height <= 0 || p->width > 65535 || p->height > 65535 || (long long)p->width * p->height > (1ll << 28))
    return fail(RT_ERR_ARG, "image size %dx%d out of range", p->width, p->height);
  HIPCHK(hipSetDevice(ctx->device));
  size_t npix = (size_t)p->width * p->height;
  size_t rgb_bytes = npix * 3 * sizeof(float), argb_bytes = npix * sizeof(uint32_t);
  int rc = ensure_io(ctx, rgb_bytes + argb_bytes);
  if (rc) return rc;
  float* d_rgb = static_cast<float*>(ctx->d_io);
  uint32_t* d_argb = reinterpret_cast<uint32_t*>(static_cast<char*>(ctx->d_io) + rgb_bytes);
  rc = rt_render_device(ctx, sc, cam, p, rgb_sum ? d_rgb : nullptr, argb ? d_argb : nullptr, stats, nullptr);
  if (rc) return rc;
  if (rgb_sum) HIPCHK(hipMemcpy(rgb_sum, d_rgb, rgb_bytes, hipMemcpyDeviceToHost));
  if (argb) HIPCHK(hipMemcpy(argb, d_argb, argb_bytes, hipMemcpyDeviceToHost));
  HIPCHK(hipDeviceSynchronize());
  return RT_OK;
}

int rt_tonemap_device(rt_ctx* ctx, const float* d_rgb_sum, int32_t n_pixels, int32_t spp_total, uint32_t* d_argb,
                      void* hip_stream) {
  if (!ctx || !d_rgb_sum || !d_argb || n_pixels <= 0 || spp_total <= 0) return fail(RT_ERR_ARG, "bad tonemap arguments");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(launch_tonemap(d_rgb_sum, n_pixels, spp_total, d_argb, static_cast<hipStream_t>(hip_stream)));
  return RT_OK;
}

int rt_hit_batch_variant(rt_ctx* ctx, const rt_scene* sc, int32_t variant, int32_t n, const float* orig3,
                         const float* dir3, const float* time, float t_min, float t_max, const float* medium_xi,
                         rt_hit* out) {
  if (!ctx || !sc || sc->ctx != ctx) return fail(RT_ERR_ARG, "ctx/scene is NULL or the scene belongs to another context");
  if (n < 0 || (n > 0 && (!orig3 || !dir3 || !out))) return fail(RT_ERR_ARG, "bad ray batch arguments");
  if (variant == RT_VARIANT_GENERAL) variant = 6;  // V_FULL, the instantiation that covers every world
  if (variant == RT_VARIANT_RENDER) variant = render_variant(sc->S);
  if (variant < 0 || variant >= RT_N_VARIANTS) return fail(RT_ERR_ARG, "unknown kernel instantiation %d", variant);
  if (!variant_covers(sc->S, variant))
    return fail(RT_ERR_UNSUPPORTED, "kernel instantiation %d does not cover this world (features 0x%x)", variant, sc->S.features);
  if (n == 0) return RT_OK;
  if (t_min < 0.f && sc->n_shadowed_spheres > 0)
    return fail(RT_ERR_UNSUPPORTED, "t_min < 0 with %d coincident duplicate sphere(s) in the world: they are left out of the "
                "scan, which is only exact for t_min >= 0", sc->n_shadowed_spheres);
  if (sc->has_time_bounds) {
    for (int i = 0; i < n; ++i) {
      float t = time ? time[i] : 0.f;
      if (!(t >= sc->time_lo && t <= sc->time_hi))
        return fail(RT_ERR_ARG, "ray %d: time %g outside [%g, %g]", i, t, sc->time_lo, sc->time_hi);
    }
  }
  HIPCHK(hipSetDevice(ctx->device));
  size_t b3 = (size_t)n * 3 * sizeof(float), b1 = (size_t)n * sizeof(float);
  size_t bxi = medium_xi ? (size_t)n * std::max(sc->n_media, 1) * sizeof(float) : 0;
  size_t bout = (size_t)n * sizeof(rt_hit);
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  int rc = ensure_io(ctx, al(b3) * 2 + al(b1) + al(bxi) + al(bout));
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->d_io);
  float* d_o = reinterpret_cast<float*>(base);
  float* d_d = reinterpret_cast<float*>(base + al(b3));
  float* d_t = reinterpret_cast<float*>(base + al(b3) * 2);
  float* d_xi = reinterpret_cast<float*>(base + al(b3) * 2 + al(b1));
  rt_hit* d_out = reinterpret_cast<rt_hit*>(base + al(b3) * 2 + al(b1) + al(bxi));
  HIPCHK(hipMemcpy(d_o, orig3, b3, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_d, dir3, b3, hipMemcpyHostToDevice));
  if (time) HIPCHK(hipMemcpy(d_t, time, b1, hipMemcpyHostToDevice));
  if (medium_xi && sc->n_media > 0) HIPCHK(hipMemcpy(d_xi, medium_xi, (size_t)n * sc->n_media * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(launch_hit_batch(sc->S, variant, n, d_o, d_d, time ? d_t : nullptr, t_min, t_max, medium_xi ? d_xi : nullptr, d_out, 0));
  HIPCHK(hipMemcpy(out, d_out, bout, hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_hit_batch(rt_ctx* ctx, const rt_scene* sc, int32_t n, const float* orig3, const float* dir3, const float* time,
                 float t_min, float t_max, const float* medium_xi, rt_hit* out) {
  return rt_hit_batch_variant(ctx, sc, RT_VARIANT_GENERAL, n, orig3, dir3, time, t_min, t_max, medium_xi, out);
}

int rt_scatter_batch(rt_ctx* ctx, const rt_scene* sc, uint64_t seed, int32_t n, const float* orig3, const float* dir3,
                     const float* time, const int32_t* px, const int32_t* smp, const int32_t* bounce, rt_scatter* out) {
  if (!ctx || !sc || sc->ctx != ctx) return fail(RT_ERR_ARG, "ctx/scene is NULL or the scene belongs to another context");
  if (n < 0 || (n > 0 && (!orig3 || !dir3 || !px || !smp || !bounce || !out))) return fail(RT_ERR_ARG, "bad scatter batch arguments");
  if (n == 0) return RT_OK;
  for (int i = 0; i < n; ++i) {
    if (px[i] < 0 || smp[i] < 0 || bounce[i] < 0 || bounce[i] >= (1 << 26))
      return fail(RT_ERR_ARG, "ray %d: pixel, sample or bounce index out of range", i);
    const float t = time ? time[i] : 0.f;
    if (sc->has_time_bounds && !(t >= sc->time_lo && t <= sc->time_hi))
      return fail(RT_ERR_ARG, "ray %d: time %g outside [%g, %g]", i, t, sc->time_lo, sc->time_hi);
  }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t b3 = (size_t)n * 3 * sizeof(float), b1 = (size_t)n * sizeof(float), bout = (size_t)n * sizeof(rt_scatter);
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  int rc = ensure_io(ctx, al(b3) * 2 + al(b1) * 4 + al(bout));
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->d_io);
  float* d_o = reinterpret_cast<float*>(base);
  float* d_d = reinterpret_cast<float*>(base + al(b3));
  float* d_t = reinterpret_cast<float*>(base + al(b3) * 2);
  int* d_px = reinterpret_cast<int*>(base + al(b3) * 2 + al(b1));
  int* d_smp = reinterpret_cast<int*>(base + al(b3) * 2 + al(b1) * 2);
  int* d_b = reinterpret_cast<int*>(base + al(b3) * 2 + al(b1) * 3);
  rt_scatter* d_out = reinterpret_cast<rt_scatter*>(base + al(b3) * 2 + al(b1) * 4);
  HIPCHK(hipMemcpy(d_o, orig3, b3, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_d, dir3, b3, hipMemcpyHostToDevice));
  if (time) HIPCHK(hipMemcpy(d_t, time, b1, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_px, px, b1, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_smp, smp, b1, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_b, bounce, b1, hipMemcpyHostToDevice));
  HIPCHK(launch_scatter_batch(sc->S, (uint32_t)seed, (uint32_t)(seed >> 32), n, d_o, d_d, time ? d_t : nullptr, d_px, d_smp,
                              d_b, d_out, 0));
  HIPCHK(hipMemcpy(out, d_out, bout, hipMemcpyDeviceToHost));
  return RT_OK;
}

static int sample_batch_impl(rt_ctx* ctx, const rt_scene* sc, const rt_camera* cam, const rt_params* p, int32_t n,
                             const int32_t* px, const int32_t* smp, float* out_rgb) {
  if (!ctx || !sc || sc->ctx != ctx) return fail(RT_ERR_ARG, "ctx/scene is NULL or the scene belongs to another context");
  DRender R;
  int rc = fill_render(sc, cam, p, R);
  if (rc) return rc;
  if (n < 0 || (n > 0 && (!px || !smp || !out_rgb))) return fail(RT_ERR_ARG, "bad sample batch arguments");
  if (n == 0) return RT_OK;
  for (int i = 0; i < n; ++i)
    if (px[i] < 0 || px[i] >= R.width * R.height || smp[i] < 0) return fail(RT_ERR_ARG, "sample %d: pixel or sample index out of range", i);
  HIPCHK(hipSetDevice(ctx->device));
  DCamera C;
  fill_camera(cam, p, C);
  size_t bi = (size_t)n * sizeof(int32_t), bo = (size_t)n * 3 * sizeof(float);
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  rc = ensure_io(ctx, al(bi) * 2 + al(bo));
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->d_io);
  int* d_px = reinterpret_cast<int*>(base);
  int* d_smp = reinterpret_cast<int*>(base + al(bi));
  float* d_out = reinterpret_cast<float*>(base + al(bi) * 2);
  HIPCHK(hipMemcpy(d_px, px, bi, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_smp, smp, bi, hipMemcpyHostToDevice));
  HIPCHK(launch_sample_batch(sc->S, C, R, n, d_px, d_smp, d_out, 0));
  HIPCHK(hipMemcpy(out_rgb, d_out, bo, hipMemcpyDeviceToHost));
  return RT_OK;
}

int rt_sample_batch(rt_ctx* ctx, const rt_scene* sc, const rt_camera* cam, const rt_params* p, int32_t n,
                    const int32_t* px, const int32_t* smp, float* out_rgb) {
  return guarded("rt_sample_batch", [&] { return sample_batch_impl(ctx, sc, cam, p, n, px, smp, out_rgb); });
}

// ------------------------------------------------------------------------------- scene files
// A scene file is the rt_scene_desc as one little-endian blob: a 64-byte header (magic, ABI version,
// the nine counts in struct order) and then every array of the struct in struct order, each padded
// to a multiple of 8 bytes, the images last (h*w*3 bytes each). The offsets follow from the counts,
// so a file can be mapped and pointed into without a parse step.
namespace {

constexpr char kSceneMagic[8] = {'R', 'T', 'M', 'I', 'S', 'C', 'N', '1'};
constexpr uint64_t kSceneHeader = 64;

struct SceneSpans {  // byte offset of every array, and the total
  uint64_t prim_type, prim_d, prim_mat, node_kind, node_i, node_d, node_children, mat_type, mat_d, mat_tex, tex_type,
      tex_d, tex_i, perlin_ranvec, perlin_perm, img_w, img_h, images, total;
};

uint64_t pad8(uint64_t x) { return (x + 7u) & ~(uint64_t)7u; }

// offsets up to the image sizes (`images`); `total` is filled once the sizes are known
SceneSpans scene_spans(const int32_t c[9]) {
  const uint64_t n_prims = c[0], n_nodes = c[1], n_children = c[2], n_mats = c[5], n_tex = c[6], n_perlin = c[7], n_images = c[8];
  SceneSpans s{};
  uint64_t o = kSceneHeader;
  auto take = [&](uint64_t& field, uint64_t bytes) { field = o; o += pad8(bytes); };
  take(s.prim_type, n_prims * 4);
  take(s.prim_d, n_prims * RT_PRIM_STRIDE * 8);
  take(s.prim_mat, n_prims * 4);
  take(s.node_kind, n_nodes * 4);
  take(s.node_i, n_nodes * RT_NODE_ISTRIDE * 4);
  take(s.node_d, n_nodes * RT_NODE_DSTRIDE * 8);
  take(s.node_children, n_children * 4);
  take(s.mat_type, n_mats * 4);
  take(s.mat_d, n_mats * RT_MAT_DSTRIDE * 8);
  take(s.mat_tex, n_mats * 4);
  take(s.tex_type, n_tex * 4);
  take(s.tex_d, n_tex * RT_TEX_DSTRIDE * 8);
  take(s.tex_i, n_tex * RT_TEX_ISTRIDE * 4);
  take(s.perlin_ranvec, n_perlin * 256 * 3 * 8);
  take(s.perlin_perm, n_perlin * 3 * 256 * 4);
  take(s.img_w, n_images * 4);
  take(s.img_h, n_images * 4);
  s.images = o;
  s.total = o;
  return s;
}

void desc_counts(const rt_scene_desc* d, int32_t c[9]) {
  c[0] = d->n_prims; c[1] = d->n_nodes; c[2] = d->n_children; c[3] = d->root; c[4] = d->n_media;
  c[5] = d->n_mats; c[6] = d->n_tex; c[7] = d->n_perlin; c[8] = d->n_images;
}

// Points `d` into a blob (8-byte aligned, n bytes); `imgs` receives the per-image pointers.
int desc_from_blob(const unsigned char* b, uint64_t n, rt_scene_desc* d, std::vector<const uint8_t*>& imgs) {
  if (n < kSceneHeader || memcmp(b, kSceneMagic, 8) != 0) return fail(RT_ERR_ARG, "not a scene file (bad magic or shorter than its header)");
  int32_t head[14];
  memcpy(head, b + 8, sizeof head);
  const int32_t* c = head + 1;
  for (int i = 0; i < 9; ++i)
    if (c[i] < 0 || c[i] > (1 << 28)) return fail(RT_ERR_ARG, "scene file: count %d out of range", i);
  SceneSpans s = scene_spans(c);
  if (s.images > n) return fail(RT_ERR_ARG, "scene file truncated: arrays need %llu bytes, file has %llu", (unsigned long long)s.images, (unsigned long long)n);
  memset(d, 0, sizeof *d);
  d->abi_version = head[0];
  d->n_prims = c[0]; d->n_nodes = c[1]; d->n_children = c[2]; d->root = c[3]; d->n_media = c[4];
  d->n_mats = c[5]; d->n_tex = c[6]; d->n_perlin = c[7]; d->n_images = c[8];
  auto i32 = [&](uint64_t o) { return reinterpret_cast<const int32_t*>(b + o); };
  auto f64 = [&](uint64_t o) { return reinterpret_cast<const double*>(b + o); };
  d->prim_type = i32(s.prim_type); d->prim_d = f64(s.prim_d); d->prim_mat = i32(s.prim_mat);
  d->node_kind = i32(s.node_kind); d->node_i = i32(s.node_i); d->node_d = f64(s.node_d);
  d->node_children = i32(s.node_children);
  d->mat_type = i32(s.mat_type); d->mat_d = f64(s.mat_d); d->mat_tex = i32(s.mat_tex);
  d->tex_type = i32(s.tex_type); d->tex_d = f64(s.tex_d); d->tex_i = i32(s.tex_i);
  d->perlin_ranvec = f64(s.perlin_ranvec); d->perlin_perm = i32(s.perlin_perm);
  d->img_w = i32(s.img_w); d->img_h = i32(s.img_h);
  imgs.assign((size_t)d->n_images, nullptr);
  uint64_t o = s.images;
  for (int i = 0; i < d->n_images; ++i) {
    const int64_t w = d->img_w[i], h = d->img_h[i];
    if (w <= 0 || h <= 0 || w > 65536 || h > 65536) return fail(RT_ERR_ARG, "scene file: image %d has size %lld x %lld", i, (long long)w, (long long)h);
    const uint64_t bytes = (uint64_t)w * (uint64_t)h * 3u;
    if (o + bytes > n) return fail(RT_ERR_ARG, "scene file truncated inside image %d", i);
    imgs[(size_t)i] = b + o;
    o += pad8(bytes);
  }
  d->img_rgb8 = imgs.data();
  return RT_OK;
}

}  // namespace

static uint64_t scene_file_size_impl(const rt_scene_desc* d) {
  if (validate_desc(d) != RT_OK) return 0;
  int32_t c[9];
  desc_counts(d, c);
  uint64_t total = scene_spans(c).images;
  for (int i = 0; i < d->n_images; ++i) total += pad8((uint64_t)d->img_w[i] * (uint64_t)d->img_h[i] * 3u);
  return total;
}

uint64_t rt_scene_file_size(const rt_scene_desc* d) {
  uint64_t total = 0;
  if (guarded("rt_scene_file_size", [&] { total = scene_file_size_impl(d); return RT_OK; }) != RT_OK) return 0;
  return total;
}

int rt_scene_write(const rt_scene_desc* d, void* bytes, uint64_t capacity, uint64_t* written) {
  const uint64_t total = rt_scene_file_size(d);
  if (!total) return RT_ERR_ARG;  // message set by the validation
  if (!bytes || capacity < total) return fail(RT_ERR_ARG, "scene file needs %llu bytes, buffer has %llu", (unsigned long long)total, (unsigned long long)capacity);
  unsigned char* b = static_cast<unsigned char*>(bytes);
  memset(b, 0, total);  // header tail and every pad byte are zero: equal scenes give equal files
  int32_t head[14] = {0};
  head[0] = d->abi_version;
  desc_counts(d, head + 1);
  memcpy(b, kSceneMagic, 8);
  memcpy(b + 8, head, sizeof head);
  SceneSpans s = scene_spans(head + 1);
  auto put = [&](uint64_t o, const void* src, uint64_t n) { if (n) memcpy(b + o, src, n); };
  const uint64_t np = d->n_prims, nn = d->n_nodes, nm = d->n_mats, nt = d->n_tex, nq = d->n_perlin, ni = d->n_images;
  put(s.prim_type, d->prim_type, np * 4); put(s.prim_d, d->prim_d, np * RT_PRIM_STRIDE * 8); put(s.prim_mat, d->prim_mat, np * 4);
  put(s.node_kind, d->node_kind, nn * 4); put(s.node_i, d->node_i, nn * RT_NODE_ISTRIDE * 4); put(s.node_d, d->node_d, nn * RT_NODE_DSTRIDE * 8);
  put(s.node_children, d->node_children, (uint64_t)d->n_children * 4);
  put(s.mat_type, d->mat_type, nm * 4); put(s.mat_d, d->mat_d, nm * RT_MAT_DSTRIDE * 8); put(s.mat_tex, d->mat_tex, nm * 4);
  put(s.tex_type, d->tex_type, nt * 4); put(s.tex_d, d->tex_d, nt * RT_TEX_DSTRIDE * 8); put(s.tex_i, d->tex_i, nt * RT_TEX_ISTRIDE * 4);
  put(s.perlin_ranvec, d->perlin_ranvec, nq * 256 * 3 * 8); put(s.perlin_perm, d->perlin_perm, nq * 3 * 256 * 4);
  put(s.img_w, d->img_w, ni * 4); put(s.img_h, d->img_h, ni * 4);
  uint64_t o = s.images;
  for (int i = 0; i < d->n_images; ++i) {
    const uint64_t n = (uint64_t)d->img_w[i] * (uint64_t)d->img_h[i] * 3u;
    put(o, d->img_rgb8[i], n);
    o += pad8(n);
  }
  if (written) *written = total;
  return RT_OK;
}

int rt_scene_create_from_memory(rt_ctx* ctx, const void* bytes, uint64_t n_bytes, rt_scene** out) {
  if (!bytes || !out) return fail(RT_ERR_ARG, "scene bytes or out is NULL");
  try {
    std::vector<uint64_t> aligned;  // the arrays are read in place, so the blob must sit on an 8-byte boundary
    const unsigned char* b = static_cast<const unsigned char*>(bytes);
    if (reinterpret_cast<uintptr_t>(b) & 7u) {
      aligned.resize((size_t)((n_bytes + 7) / 8));
      memcpy(aligned.data(), bytes, (size_t)n_bytes);
      b = reinterpret_cast<const unsigned char*>(aligned.data());
    }
    rt_scene_desc d;
    std::vector<const uint8_t*> imgs;
    int rc = desc_from_blob(b, n_bytes, &d, imgs);
    if (rc) return rc;
    return rt_scene_create(ctx, &d, out);
  } catch (const std::bad_alloc&) {
    return fail(RT_ERR_OOM, "out of host memory reading a scene");
  }
}

int rt_scene_create_from_file(rt_ctx* ctx, const char* path, rt_scene** out) {
  if (!path || !out) return fail(RT_ERR_ARG, "scene path or out is NULL");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(RT_ERR_ARG, "cannot open scene file %s", path);
  int rc = RT_OK;
  try {
    std::vector<uint64_t> buf;
    long len = -1;
    if (fseek(f, 0, SEEK_END) == 0) len = ftell(f);
    if (len < 0 || fseek(f, 0, SEEK_SET) != 0) {
      rc = fail(RT_ERR_ARG, "cannot size scene file %s", path);
    } else {
      buf.resize(((size_t)len + 7) / 8);
      if (len > 0 && fread(buf.data(), 1, (size_t)len, f) != (size_t)len) rc = fail(RT_ERR_ARG, "short read on scene file %s", path);
    }
    fclose(f);
    f = nullptr;
    if (rc) return rc;
    return rt_scene_create_from_memory(ctx, buf.data(), (uint64_t)len, out);
  } catch (const std::bad_alloc&) {
    if (f) fclose(f);
    return fail(RT_ERR_OOM, "out of host memory reading scene file %s", path);
  }
}

}  // extern "C"
