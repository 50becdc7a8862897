// rt_kernels.hip — the kernels of librtmi.so (gfx950).
//
//   k_render        the per-pixel sample loop of the reference (core.clj:96-120) as a persistent
//                   path loop: every lane carries one path, regenerates it in place when it
//                   ends, and pulls new (pixel, sample-chunk) work items through a wave-local
//                   queue refilled with one atomic per 64 items; lanes needing work are ranked
//                   with a ballot + mbcnt prefix count.
//   k_resolve       sums the per-chunk partial sums in a fixed order and optionally applies the
//                   pixel post-process (core.clj:106-108,118,121-122).
//   k_tonemap       that post-process alone.
//   k_hit_batch     closest-hit records for caller-supplied rays (parity hook).
//   k_sample_batch  radiance of single (pixel, sample) units (parity hook).
#include <hip/hip_runtime.h>

#include "rt_device.hpp"
#include "rt_launch.h"

#ifndef RT_REGEN_MIN
#define RT_REGEN_MIN 8
#endif
#ifndef RT_WAVES_SPH
#define RT_WAVES_SPH 5
#endif
#ifndef RT_WAVES_BALLS
#define RT_WAVES_BALLS 6
#endif
#ifndef RT_WAVES_ALL
#define RT_WAVES_ALL 4
#endif
#ifndef RT_WAVES_DEEP
#define RT_WAVES_DEEP 4
#endif
#ifndef RT_WAVES_NOISE
#define RT_WAVES_NOISE 6
#endif
#ifndef RT_WAVES_LIST
#define RT_WAVES_LIST 7
#endif
#ifndef RT_REGEN_SMOKE
#define RT_REGEN_SMOKE 1
#endif
#ifndef RT_REGEN_REST
#define RT_REGEN_REST 1
#endif
#ifndef RT_WAVES_SMOKE
#define RT_WAVES_SMOKE 5
#endif
#ifndef RT_WAVES_LEAN
#define RT_WAVES_LEAN 6
#endif

namespace rt {

// Dynamic LDS layout shared by all kernels that shade:
//   [0, n_perlin*4096)                 Perlin gradient vectors, float4 each
//   [.., + n_perlin*768 rounded to 16) Perlin permutations, bytes
//   [.., + n_prims*96 + n_chains*64)   DPrimInfo and DChainRec tables (F_SMALL instantiations only)
//   [.., + blockDim/64 * 1024)         one exchange area per wave (wave_random_in_unit_sphere); kernels that walk trees
//                                      whose stacks have four levels or more lay it over those instead (xch_on_stacks)
//   [.., + blockDim*bvh_stack*4)       per-lane BVH stacks, entry e of lane l at e*blockDim + l
//   [.., + lds_nodes*16)               the second half of every packed tree node (instantiations that walk trees,
//                                      when the scene's trees fit: DScene::lds_nodes)
template <uint32_t FEAT>
__device__ __forceinline__ Ctx<FEAT> make_ctx(const DScene& S, unsigned char* lds) {
  Ctx<FEAT> X;
  X.S = S;
  size_t vec_bytes = (size_t)S.n_perlin * 4096;
  size_t perm_bytes = ((size_t)S.n_perlin * 768 + 15) & ~(size_t)15;
  float4* lv = reinterpret_cast<float4*>(lds);
  uint8_t* lp = lds + vec_bytes;
  const float4* gv = reinterpret_cast<const float4*>(S.perlin_vec);
  size_t info_bytes = 0;
  if constexpr ((FEAT & F_NOISE) != 0) {
    for (int i = threadIdx.x; i < S.n_perlin * 256; i += blockDim.x) lv[i] = gv[i];
    for (int i = threadIdx.x; i < S.n_perlin * 768; i += blockDim.x) lp[i] = S.perlin_perm[i];
  }
  X.lds_info = nullptr;
  X.lds_chains = nullptr;
  if constexpr ((FEAT & F_SMALL) != 0) {
    const size_t pi_bytes = (size_t)S.n_prims * sizeof(DPrimInfo);
    const size_t ch_bytes = (size_t)S.n_chains * sizeof(DChainRec);
    info_bytes = pi_bytes + ch_bytes;
    uint4* li = reinterpret_cast<uint4*>(lds + vec_bytes + perm_bytes);
    const uint4* gi = reinterpret_cast<const uint4*>(S.prim_info);
    for (int i = threadIdx.x; i < (int)(pi_bytes / 16); i += blockDim.x) li[i] = gi[i];
    uint4* lc = reinterpret_cast<uint4*>(lds + vec_bytes + perm_bytes + pi_bytes);
    const uint4* gc = reinterpret_cast<const uint4*>(S.chain_recs);
    for (int i = threadIdx.x; i < (int)(ch_bytes / 16); i += blockDim.x) lc[i] = gc[i];
    X.lds_info = (const DPrimInfo RT_LDS*)(lds + vec_bytes + perm_bytes);
    X.lds_chains = (const DChainRec RT_LDS*)(lds + vec_bytes + perm_bytes + pi_bytes);
  }
  if constexpr ((FEAT & (F_NOISE | F_SMALL)) != 0) __syncthreads();
  X.lds_perlin_vec = reinterpret_cast<const float*>(lv);
  X.lds_perlin_perm = lp;
  // (scene-uniform) kernels that walk trees lay the exchange area over their traversal stacks
  const bool xch_on_stacks = (FEAT & F_BVH) != 0 && S.xch_on_stacks != 0;
  const size_t xch_bytes = xch_on_stacks ? 0 : (size_t)(blockDim.x >> 6) * 1024;
  X.lds_xch = xch_on_stacks ? (uint32_t RT_LDS*)(lds + vec_bytes + perm_bytes + info_bytes) + (threadIdx.x >> 6) * 64
                            : (uint32_t RT_LDS*)(lds + vec_bytes + perm_bytes + info_bytes) + (threadIdx.x >> 6) * 256;
  X.xch_stride = xch_on_stacks ? blockDim.x : 64u;
  X.lds_stack = reinterpret_cast<int*>(lds + vec_bytes + perm_bytes + info_bytes + xch_bytes) + threadIdx.x;
  X.stack_stride = blockDim.x;
  X.lds_node_half = nullptr;
  if constexpr ((FEAT & F_BVH) != 0) {
    if (S.lds_nodes > 0) {  // scene-uniform
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      u4* ln = reinterpret_cast<u4*>(lds + vec_bytes + perm_bytes + info_bytes + xch_bytes + (size_t)blockDim.x * (size_t)S.bvh_stack * 4);
      const u4* gn = reinterpret_cast<const u4*>(S.nodes);
      for (int i = threadIdx.x; i < S.lds_nodes; i += blockDim.x) ln[i] = gn[i * 2 + 1];
      __syncthreads();
      X.lds_node_half = (const u4 RT_LDS*)ln;
    }
  }
#ifdef RT_STAMPS
  X.st_leaf = X.st_walk = X.st_rounds = nullptr; X.st_grp = nullptr;
#endif
  return X;
}

// Sum of v over the 64 lanes of the wave (every lane must call it), returned in all lanes.
__device__ __forceinline__ unsigned long long wave_sum(uint32_t v) {
  unsigned long long x = v;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}

__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// One ray-color invocation (core.clj:19-30) for the lane's current path. Returns true when the
// path ended; `L` is then the radiance of the whole path (throughput already applied).
// Diagnostic build only (-DRT_STAMPS): wave-cycle shares of the loop phases, accumulated per wave
// and written over the event slots of the stats buffer. Never timed, never shipped.
#ifdef RT_STAMPS
#define RT_STAMP(var) { __builtin_amdgcn_sched_barrier(0); var = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#elif defined(RT_PHASE_MARKS)
// listing build (tools/isa_histogram.py): a comment in the assembly where each phase of the loop begins
#define RT_STAMP(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("; RTPHASE " #var); __builtin_amdgcn_sched_barrier(0); }
#else
#define RT_STAMP(var)
#endif

// The part of ray-color after the `hit` call (core.clj:24-30): `sv` is this bounce's
// random-in-unit-sphere vector when the material draws one. Returns true when the path ended;
// `L` is then the radiance of the whole path (throughput already applied).
template <bool PRE, class XC, class T>
__device__ __forceinline__ bool path_finish(const XC& X, const DRender& R, bool hit, const Rec& rec, V3 sv, Ray& ray,
                                            V3& beta, uint32_t& bounce, const Rng& g, V3& L, T& tally) {
  if (!hit) {
    L = beta * mk(R.bg[0], R.bg[1], R.bg[2]);  // core.clj:30
    tally.template add<RT_EV_THROUGHPUT>();
    return true;
  }
  V3 attenuation, emitted;
  if (scatter<PRE>(X, rec, ray, g, bounce, sv, attenuation, emitted, tally)) {
    beta = beta * attenuation;  // core.clj:27
    tally.template add<RT_EV_THROUGHPUT>();
    ++bounce;
    if ((int)bounce >= R.max_depth) {  // core.clj:20-21
      L = mk(0.f, 0.f, 0.f);
      return true;
    }
    return false;
  }
  L = beta * emitted;  // core.clj:29
  tally.template add<RT_EV_THROUGHPUT>();
  return true;
}

// One whole ray-color invocation for a lane on its own: the material draws its sphere vector with
// the plain rejection loop.
template <class XC, class T>
__device__ __forceinline__ bool path_step(const XC& X, const DRender& R, Ray& ray, V3& beta, uint32_t& bounce,
                                          const Rng& g, V3& L, T& tally) {
  Rec rec;
  const bool hit = closest_hit(X, ray, 0.001f, __builtin_inff(), g, bounce, nullptr, false, false, rec, tally);
  return path_finish<false>(X, R, hit, rec, mk(0.f, 0.f, 0.f), ray, beta, bounce, g, L, tally);
}

// Waves per SIMD each instantiation is compiled for (the register allocator otherwise spreads over
// the whole file): the lean instantiations are latency-bound on scalar loads and the RNG's
// integer multiplies, which more resident waves hide.
constexpr int waves_per_simd(uint32_t feat, bool count) {
  // the all-features kernel fits 128 registers with 26 spilled (the FP64 arccosine and arctangent of
  // the sphere u/v once made it want 211): four waves measured 9 % faster than three on the texture
  // worlds, five 20 % slower everywhere (F_DEEP is the same code under a knob of its own)
  // the noise-only kernel wants 105 registers where the all-features one wants 128 and spills: measured
  // at 3 / 4 / 5 / 6 / 7 / 8 waves: 94 / 96 / 91 / 90 / 96 / 112 ms on the Perlin spheres (all-features: 109)
  if ((feat & F_NOISE) && !(feat & (F_BVH | F_MEDIA | F_CHAINS))) return count ? 2 : RT_WAVES_NOISE;
  if (feat & F_NOISE) return count ? 2 : ((feat & F_DEEP) ? RT_WAVES_DEEP : RT_WAVES_ALL);
  // spheres in plain lists: at 5 / 6 / 7 / 8 waves the earth takes 22.5 / 21.2 / 21.0 / 22.6 ms (23.5 on the
  // kernel that also walks trees), the two checkered spheres 73.3 / 70.0 / 69.8 / 75.1 (75.9)
  if ((feat & F_SPHERES) && !(feat & (F_BVH | F_NOISE | F_MEDIA | F_CHAINS))) return count ? 2 : RT_WAVES_LIST;
  // spheres alone: 96 registers at five waves, 80 at six, nothing spilled either way; 4 / 5 / 6 waves render the
  // random-spheres world in 67.5 / 62.4 / 60.5 ms (a seventh workgroup per CU does not fit its LDS)
  if ((feat & F_BVH) && !(feat & F_RECTS)) return count ? 2 : RT_WAVES_BALLS;
  if (feat & (F_SPHERES | F_BVH | F_TEXTURES)) return count ? 2 : RT_WAVES_SPH;
  // the smoke worlds' kernel spills at 72 and at 80 registers: five waves measured 5 % faster than seven
  if (feat & F_MEDIA) return count ? 4 : RT_WAVES_SMOKE;
  // the rect-only kernel: six waves (85 registers, nothing spilled) measured 7 to 9 % faster than seven
  // (71 registers, 14 of them spilled and re-read in every record), five the same as six
  return count ? 4 : RT_WAVES_LEAN;
}

// Which instantiations draw the bounce's random-in-unit-sphere vectors with the whole wave
// (wave_random_in_unit_sphere) instead of a rejection loop per lane. Measured: the all-features
// kernels +1 to +3 % (it used to cost them registers they did not have), the sphere/BVH kernel
// -8 % (random-scene) and -20 % (two-spheres), the list kernel -12 % (earth) and -7 % (two-spheres) at six
// waves per SIMD and -29 % at its own seven.
#ifndef RT_WAVE_DRAW_ALL
#define RT_WAVE_DRAW_ALL 1
#endif
#ifndef RT_WAVE_DRAW_DEEP
#define RT_WAVE_DRAW_DEEP 1
#endif
#ifndef RT_WAVE_DRAW_SPH
#define RT_WAVE_DRAW_SPH 0
#endif
#ifndef RT_WAVE_DRAW_LIST
#define RT_WAVE_DRAW_LIST 0
#endif
constexpr bool wave_draw(uint32_t feat) {
  if (feat & F_SMALL) return true;
  if ((feat & F_SPHERES) && !(feat & (F_BVH | F_NOISE | F_MEDIA | F_CHAINS))) return RT_WAVE_DRAW_LIST != 0;
  if (feat & F_NOISE) return (feat & F_DEEP) ? RT_WAVE_DRAW_DEEP != 0 : RT_WAVE_DRAW_ALL != 0;
  return RT_WAVE_DRAW_SPH != 0;
}

// Where shading costs far more than finding the hit (a noise texture's seven octaves against two or three sphere
// tests), a path that misses leaves its lane idle through the whole shading phase. The kernels of such worlds hold
// their hits back: lanes whose path ended take their next sample and look again, and the wave shades when
// RT_SHADE_MIN lanes hold a hit (or nothing else can be done, or RT_SHADE_WAIT rounds went by). Samples are keyed,
// and a lane still takes its item's samples in order: the frame is the same, bit for bit.
#ifndef RT_SHADE_MIN
#define RT_SHADE_MIN 52
#endif
#ifndef RT_SHADE_WAIT
#define RT_SHADE_WAIT 3
#endif
#ifndef RT_DEFER_NOISE
#define RT_DEFER_NOISE 1
#endif
constexpr bool defer_shading(uint32_t feat) {
  return RT_DEFER_NOISE != 0 && (feat & F_NOISE) && !(feat & (F_BVH | F_MEDIA | F_CHAINS | F_SMALL)) && wave_draw(feat);
}

// All arguments of k_render in one struct, so that its layout IS the kernarg segment's. Fields the
// loop needs rarely (the camera's 21 doubles, the queue geometry) are not kept in SGPRs across the
// traversal loops, where they would be spilled into VGPR lanes and read back with VALU
// instructions; they are re-read from the kernarg segment (scalar loads, scalar cache) at the point
// of use, through a pointer the optimiser cannot see through (see `cold()`).
struct RenderArgs {
  DScene S;
  DCamera cam;
  DRender R;
  float* out;
  uint32_t* queue;
  unsigned long long* stats;
};
__device__ __forceinline__ const RenderArgs RT_CONST* cold() {
  const RenderArgs RT_CONST* p = (const RenderArgs RT_CONST*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));  // opaque: loads through p are neither hoisted out of the loop nor merged
  return p;
}

template <uint32_t FEAT, bool COUNT>
__global__ __launch_bounds__(RT_BLOCK, waves_per_simd(FEAT, COUNT)) void k_render(RenderArgs A) {
  extern __shared__ __align__(16) unsigned char lds[];
  Ctx<FEAT> X = make_ctx<FEAT>(A.S, lds);
#ifdef RT_STAMPS
  unsigned long long c_leaf = 0, c_walk = 0, c_rounds = 0;
  unsigned long long c_grp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  X.st_leaf = &c_leaf; X.st_walk = &c_walk; X.st_rounds = &c_rounds; X.st_grp = c_grp;
#endif
  const DRender& R = A.R;
  float* __restrict__ out = A.out;
  unsigned long long* __restrict__ stats = A.stats;
  Tally<COUNT> tally;
  if constexpr (COUNT) tally.clear();
  // the shader clock during this kernel: one wave notes the cycle counter and the constant-rate timer
  const bool clock_wave = blockIdx.x == 0 && threadIdx.x == 0;
  if (clock_wave) {
    stats[RT_STATS_CLOCK + 0] = __builtin_readcyclecounter();
    stats[RT_STATS_CLOCK + 2] = __builtin_amdgcn_s_memrealtime();
  }

  // wave-uniform queue window [q_next, q_end) of work items taken from the global queue
  uint32_t q_next = 0, q_end = 0;
  uint32_t q_head = blockIdx.x % RT_QUEUE_HEADS, q_heads_dry = 0;  // the head this wave takes from; heads found empty
  bool q_dry = false;
  int q_tx0 = 0, q_ty0 = 0, q_s0 = 0, q_s1 = 0;  // what the window's items share

  // lane state
  bool have_item = false, active = false, done = false;
  // (the pixel index lives in g.k0 and the current sample index in g.k1: the keys of the path's
  // random stream are the item's bookkeeping)
  uint32_t pxy = 0;  // pixel x | y << 16 (both below 65536)
  uint32_t item = 0;  // the lane's work item = its slot in the partial-sum slab
  uint32_t k1_end = 0;  // g.k1 of the sample after the item's last
  V3 acc = mk(0.f, 0.f, 0.f);
  Ray ray{mk(0, 0, 0), mk(0, 0, 1), 0.f};
  V3 beta = mk(1.f, 1.f, 1.f);
  uint32_t bounce = 0;
  Rng g{0u, 0u, R.seed_lo, R.seed_hi};
  unsigned long long n_rays = 0, n_samples = 0;  // of the whole wave (wave-uniform, scalar registers)
  // defer_shading: the hit a lane holds back (its record), and how many rounds the wave has held hits back
  Rec held;
  held.mtype = -1;
  bool pending = false;
  int waited = 0;
  // HW_ID.wave_id (s_getreg id 4, offset 0, 4 bits): the wave's slot in its SIMD
  uint32_t prio_tick = __builtin_amdgcn_s_getreg(6148);

#ifdef RT_STAMPS
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, tsi = 0, c_queue = 0, c_cam = 0, c_hit = 0, c_shade = 0, c_iters = 0, c_groups = 0;
#endif
  for (;;) {
    // Rotate the issue priority among the waves of a SIMD, one step per iteration. With equal
    // priorities the arbiter favours the oldest wave, and in a kernel that saturates the VALU the
    // youngest waves then run several times slower (measured: 293 against 2 509 iterations per
    // wave in one launch). The dynamic queue hides that until it runs dry; then the slow waves
    // still sit in items they took long ago, which is what cost small jobs (one shard of eight)
    // their efficiency.
    switch (prio_tick++ & 3u) {
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
    RT_STAMP(ts0);
    // Lanes whose path ended wait until `regen_min` of them are idle: the refill and camera code
    // below runs for the whole wave however few lanes want it. Worth it where paths are long and the
    // loop is lean (the rect-only kernel: +2 % at 8, flat to worse from 16 on and elsewhere).
    constexpr int regen_min = ((FEAT & F_SMALL) != 0 && (FEAT & F_MEDIA) == 0) ? RT_REGEN_MIN : ((FEAT & F_SMALL) != 0 ? RT_REGEN_SMOKE : RT_REGEN_REST);
    // (With no path left in the wave the idle lanes go ahead whatever their number: every lane must
    // reach `done`.)
    // Once the queue has run dry the wave only has the rest of its own window left: its idle lanes take that at once
    // (an eighth-of-the-frame shard of the Cornell box ends 0.05 ms sooner, the whole frame is unchanged).
    const bool regen = __popcll(__ballot(!active && !done)) >= (q_dry ? 1 : regen_min) || __ballot(active) == 0ull;
    bool need = regen && !active && !done && !(have_item && g.k1 + 1u != k1_end);
    if (need && have_item) {
      // core.clj:115-120 finished for this chunk: store the partial sum of its samples
      // (items are numbered chunk by chunk, tile by tile: the 64 lanes of a window write 768
      // contiguous bytes)
      size_t o = (size_t)item * 3;
      out[o] = acc.x; out[o + 1] = acc.y; out[o + 2] = acc.z;
      have_item = false;
    }
    unsigned long long mask = __ballot(need);
    if (mask != 0ull) {  // wave-uniform
      const RenderArgs RT_CONST* C = cold();
      if (q_next == q_end && !q_dry) {
        // A window is 64 consecutive items = the 64 pixels of ONE tile for ONE chunk (n_items is a
        // multiple of 64), so the item is decoded once per window on wave-uniform values and a
        // lane only adds its place in the tile.
        // The queue has RT_QUEUE_HEADS heads, each owning one window of every block of RT_QUEUE_HEADS: atomics
        // on ONE address go through at about 90 M/s, which held every light world at few samples
        // per pixel to the same 5.5 G samples/s. A wave starts on the head of its workgroup
        // (workgroups go round the XCDs, so a head is mostly hit from one L2) and moves on to the
        // next one when it finds its own empty; after a full round the queue is dry.
        const uint32_t n_windows = (uint32_t)C->R.n_items >> 6;
        uint32_t base = ~0u;
        while (q_heads_dry < RT_QUEUE_HEADS) {  // wave-uniform
          uint32_t k = 0;
          if (wave_lane() == 0) k = atomicAdd(C->queue + q_head * RT_QUEUE_STRIDE, 1u);
          k = __builtin_amdgcn_readfirstlane(k);
          // (k < 2^26: at most n_windows / RT_QUEUE_HEADS + one miss per wave)
          // within each block of RT_QUEUE_HEADS windows the heads take turns (rotated by k and by
          // k / 8): a head that always got the same place in a block would own fixed tile columns
          // of the image, and columns differ in cost
          const uint32_t w = k * RT_QUEUE_HEADS + ((q_head + k + (k >> 3)) % RT_QUEUE_HEADS);
          if (w < n_windows) { base = w << 6; break; }
          q_head = (q_head + 1u) % RT_QUEUE_HEADS;
          ++q_heads_dry;
        }
        if (base == ~0u) {
          q_dry = true;
        } else {
          q_next = base;
          q_end = base + 64u;
          const uint32_t items_per_chunk = (uint32_t)C->R.n_owned_tiles * 64u;
          const uint32_t c = base / items_per_chunk;
          const uint32_t rem = base - c * items_per_chunk;
          const int tiles_x = C->R.tiles_x;
          const int tile = shard_tile((int)(rem >> 6), C->R.shard_index, C->R.shard_count, tiles_x);
          q_tx0 = (tile % tiles_x) * RT_TILE;
          q_ty0 = (tile / tiles_x) * RT_TILE;
          q_s0 = C->R.chunk_start[c];
          q_s1 = C->R.chunk_start[c + 1];
        }
      }
      uint32_t avail = q_end - q_next;
      uint32_t cnt = (uint32_t)__popcll(mask);
      if (need) {
        uint32_t rank = lane_rank(mask);
        if (rank < avail) {
          const int in_tile = (int)((q_next & 63u) + rank);
          const int px = q_tx0 + (in_tile & 7);
          const int py = q_ty0 + (in_tile >> 3);
          if (px < R.width && py < R.height) {
            pxy = (uint32_t)px | ((uint32_t)py << 16);
            g.k0 = (uint32_t)(py * R.width + px);
            item = q_next + rank;
            g.k1 = (uint32_t)(R.sample_offset + q_s0) - 1u;
            k1_end = (uint32_t)(R.sample_offset + q_s1);
            acc = mk(0.f, 0.f, 0.f);
            have_item = true;
          }
        } else if (q_dry || avail == 0) {
          if (q_dry) done = true;
        }
      }
      q_next += min(cnt, avail);
    }
    RT_STAMP(ts1);
    const bool fresh = regen && !active && have_item && g.k1 + 1u != k1_end;
    n_samples += (unsigned long long)__popcll(__ballot(fresh));
    if (fresh) {
      // core.clj:96-105: a fresh camera path for the item's next sample of this pixel
      g.k1 += 1u;
      if (R.max_depth <= 0) {
        tally.template add<RT_EV_ACCUMULATE>();
      } else {
        {
          const RenderArgs RT_CONST* C = cold();
          ray = camera_ray(&C->cam, (int)(pxy & 0xFFFFu), (int)(pxy >> 16), g, tally);
        }
        beta = mk(1.f, 1.f, 1.f);
        bounce = 0;
        active = true;
      }
    }
    if (__ballot(!done) == 0ull) break;
    RT_STAMP(ts2);
#ifdef RT_STAMPS
    ts3 = ts2; tsi = ts2;
#endif
    if constexpr (defer_shading(FEAT)) {
      // core.clj:23 for the lanes whose path has no hit in hand; a miss ends the path at once (core.clj:30)
      const bool look = active && !pending;
      n_rays += (unsigned long long)__popcll(__ballot(look));
      if (look) {
        pending = closest_hit(X, ray, 0.001f, __builtin_inff(), g, bounce, nullptr, false, false, held, tally);
        if (!pending) {
          V3 L;
          path_finish<true>(X, R, false, held, mk(0.f, 0.f, 0.f), ray, beta, bounce, g, L, tally);
          acc = acc + L;
          tally.template add<RT_EV_ACCUMULATE>();
          active = false;
        }
      }
      RT_STAMP(ts3);
      // shade now? Enough hits in hand, or no lane left that could still find one, or waited long enough
      const int n_pending = __popcll(__ballot(pending));
      const bool can_look = __ballot(!pending && !done) != 0ull;
      ++waited;
      if (n_pending >= RT_SHADE_MIN || !can_look || waited >= RT_SHADE_WAIT) {  // wave-uniform
        waited = 0;
        const V3 sv = wave_random_in_unit_sphere<(FEAT & F_BVH) != 0>(pending && needs_sphere_draw(held.mtype), g, bounce, X.lds_xch, X.xch_stride, tally);
        if (pending) {
          V3 L;
          if (path_finish<true>(X, R, true, held, sv, ray, beta, bounce, g, L, tally)) {
            acc = acc + L;  // core.clj:120
            tally.template add<RT_EV_ACCUMULATE>();
            active = false;
          }
          pending = false;
        }
      }
    } else if constexpr (wave_draw(FEAT)) {
      // core.clj:23: the closest hit, for the lanes that carry a path
      Rec rec;
      rec.mtype = -1;
      bool hit = false;
      n_rays += (unsigned long long)__popcll(__ballot(active));
      if (active) {
#ifdef RT_STAMPS
        hit = closest_hit(X, ray, 0.001f, __builtin_inff(), g, bounce, nullptr, false, false, rec, tally, &tsi);
#else
        hit = closest_hit(X, ray, 0.001f, __builtin_inff(), g, bounce, nullptr, false, false, rec, tally);
#endif
      }
      RT_STAMP(ts3);
      // The bounce's random-in-unit-sphere vectors, drawn by the whole wave together. (Drawing
      // before the hit record is built, while fewer registers are live, measured slower.)
      const V3 sv = wave_random_in_unit_sphere<(FEAT & F_BVH) != 0>(active && hit && needs_sphere_draw(rec.mtype), g, bounce, X.lds_xch, X.xch_stride, tally);
      if (active) {
        V3 L;
        if (path_finish<true>(X, R, hit, rec, sv, ray, beta, bounce, g, L, tally)) {
          acc = acc + L;  // core.clj:120
          tally.template add<RT_EV_ACCUMULATE>();
          active = false;
        }
      }
    } else {
      // The instantiations that walk a BVH lose more to the extra live state than the shared draw
      // saves: there each lane runs ray-color on its own.
      n_rays += (unsigned long long)__popcll(__ballot(active));
#ifdef RT_STAMPS
      {  // the same as path_step below, with the phases stamped
        Rec rec;
        rec.mtype = -1;
        bool hit = false;
        if (active) hit = closest_hit(X, ray, 0.001f, __builtin_inff(), g, bounce, nullptr, false, false, rec, tally, &tsi);
        RT_STAMP(ts3);
        if (active) {
          V3 L;
          if (path_finish<false>(X, R, hit, rec, mk(0.f, 0.f, 0.f), ray, beta, bounce, g, L, tally)) {
            acc = acc + L;
            active = false;
          }
        }
      }
      if (false) {
#else
      if (active) {
#endif
        V3 L;
        if (path_step(X, R, ray, beta, bounce, g, L, tally)) {
          acc = acc + L;  // core.clj:120
          tally.template add<RT_EV_ACCUMULATE>();
          active = false;
        }
      }
    }
#ifdef RT_PHASE_MARKS
    RT_STAMP(ts4);
#endif
#ifdef RT_STAMPS
    RT_STAMP(ts4);
    c_queue += ts1 - ts0; c_cam += ts2 - ts1; c_hit += ts3 - ts2; c_shade += ts4 - ts3; c_iters += 1; c_groups += tsi - ts2;
#endif
  }
  const uint32_t lane = wave_lane();
  if (clock_wave) {
    stats[RT_STATS_CLOCK + 1] = __builtin_readcyclecounter();
    stats[RT_STATS_CLOCK + 3] = __builtin_amdgcn_s_memrealtime();
  }
#ifdef RT_STAMPS
  if (lane == 0) {
    atomicAdd(&stats[2 + 0], c_queue); atomicAdd(&stats[2 + 1], c_cam); atomicAdd(&stats[2 + 2], c_hit);
    atomicAdd(&stats[2 + 3], c_shade); atomicAdd(&stats[2 + 4], c_iters); atomicAdd(&stats[2 + 5], c_groups);
    atomicMax(&stats[2 + 6], c_iters);                       // iterations of the busiest wave
    atomicMax(&stats[2 + 7], ~0ull - c_iters);               // ... and (inverted) of the idlest
    atomicAdd(&stats[2 + 8], c_iters * c_iters);             // for the spread
  }
  {  // BVH walk: the wave's time is that of the lane that stayed in each walk longest
    unsigned long long w = c_walk, l = c_leaf, r = c_rounds;
    for (int off = 32; off >= 1; off >>= 1) {
      unsigned long long w2 = __shfl_xor(w, off, 64), l2 = __shfl_xor(l, off, 64), r2 = __shfl_xor(r, off, 64);
      w = w2 > w ? w2 : w; l = l2 > l ? l2 : l; r = r2 > r ? r2 : r;
    }
    if (lane == 0) { atomicAdd(&stats[2 + 9], w); atomicAdd(&stats[2 + 10], l); atomicAdd(&stats[2 + 11], r); }
    for (int q = 0; q < 8; ++q) {  // per group: the lane that was in the group longest = the wave's time
      unsigned long long v = c_grp[q];
      for (int off = 32; off >= 1; off >>= 1) { unsigned long long v2 = __shfl_xor(v, off, 64); v = v2 > v ? v2 : v; }
      if (lane == 0) atomicAdd(&stats[2 + 12 + q], v);
    }
  }
#endif
  // One atomic per wave and counter, not per lane: half a million same-address atomics at the end
  // of the kernel serialise in L2 and cost milliseconds.
  if (lane == 0) { atomicAdd(&stats[0], n_samples); atomicAdd(&stats[1], n_rays); }
  if constexpr (COUNT) {
#pragma unroll
    for (int e = 0; e < RT_EV_COUNT; ++e) {
      unsigned long long w = wave_sum(tally.ev[e]);
      if (lane == 0 && w) atomicAdd(&stats[2 + e], w);
    }
  }
}

// Sums the chunk partials of every pixel in chunk order (a fixed order, so the sum does not
// depend on which wave rendered which chunk), writes 0 for pixels of other shards.
__global__ void k_resolve(DRender R, const float* __restrict__ partial, float* __restrict__ rgb_sum,
                          uint32_t* __restrict__ argb) {
  int pix = blockIdx.x * blockDim.x + threadIdx.x;
  int npix = R.width * R.height;
  if (pix >= npix) return;
  int x = pix % R.width, y = pix / R.width;
  int tile = (y / RT_TILE) * R.tiles_x + (x / RT_TILE);
  float r = 0.f, g = 0.f, b = 0.f;
  if (tile_shard(tile, R.shard_count, R.tiles_x) == R.shard_index) {
    // the slab is indexed by work item: chunk, then owned tile, then place in the tile
    const size_t items_per_chunk = (size_t)R.n_owned_tiles * 64;
    const size_t in_chunk = (size_t)(tile / R.shard_count) * 64 + (size_t)((y % RT_TILE) * RT_TILE + x % RT_TILE);
    for (int c = 0; c < R.n_chunks; ++c) {
      size_t o = ((size_t)c * items_per_chunk + in_chunk) * 3;
      r += partial[o]; g += partial[o + 1]; b += partial[o + 2];
    }
  }
  if (rgb_sum) { rgb_sum[3 * (size_t)pix] = r; rgb_sum[3 * (size_t)pix + 1] = g; rgb_sum[3 * (size_t)pix + 2] = b; }
  if (argb) argb[pix] = pack_pixel(r, g, b, R.spp);
}

__global__ void k_tonemap(const float* __restrict__ rgb_sum, int n_pixels, int spp, uint32_t* __restrict__ argb) {
  int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= n_pixels) return;
  argb[pix] = pack_pixel(rgb_sum[3 * (size_t)pix], rgb_sum[3 * (size_t)pix + 1], rgb_sum[3 * (size_t)pix + 2], spp);
}

// FEAT: any instantiation whose mask covers the scene answers with the same records, bit for bit
// (tests/test_gpu_variants.py); the render kernels share every device function with this one.
template <uint32_t FEAT>
__global__ __launch_bounds__(RT_BLOCK) void k_hit_batch(DScene S, int n, const float* __restrict__ orig3,
                                                        const float* __restrict__ dir3,
                                                        const float* __restrict__ time, float tmin, float tmax,
                                                        const float* __restrict__ medium_xi, rt_hit* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char lds[];
  Ctx<FEAT> X = make_ctx<FEAT>(S, lds);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ray r{mk(orig3[3 * i], orig3[3 * i + 1], orig3[3 * i + 2]), mk(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]),
        time ? time[i] : 0.f};
  Rng g{0u, 0u, 0u, 0u};
  Tally<false> tally;
  Rec rec;
  rt_hit h;
  h.t = 0.f; h.p[0] = h.p[1] = h.p[2] = 0.f; h.normal[0] = h.normal[1] = h.normal[2] = 0.f;
  h.u = h.v = 0.f; h.front_face = 0; h.mat_id = 0; h.hit = 0;
  if (closest_hit(X, r, tmin, tmax, g, 0, medium_xi ? medium_xi + (size_t)i * S.n_media : nullptr,
                  medium_xi == nullptr, true, rec, tally)) {
    h.t = rec.t;
    h.p[0] = rec.p.x; h.p[1] = rec.p.y; h.p[2] = rec.p.z;
    h.normal[0] = rec.n.x; h.normal[1] = rec.n.y; h.normal[2] = rec.n.z;
    h.u = rec.u; h.v = rec.v;
    h.front_face = rec.front ? 1 : 0;
    h.mat_id = rec.mat;
    h.hit = 1;
  }
  out[i] = h;
}

// One ray-color invocation (core.clj:23-25) for caller-supplied rays: the closest hit, then
// (scatter mat r rec) and (emitted mat u v p) of the material hit, with the keyed draws of
// (pixel px, sample smp, bounce) — what the render loop does for that path at that bounce.
template <uint32_t FEAT>
__global__ __launch_bounds__(RT_BLOCK) void k_scatter_batch(DScene S, uint32_t seed_lo, uint32_t seed_hi, int n,
                                                            const float* __restrict__ orig3, const float* __restrict__ dir3,
                                                            const float* __restrict__ time, const int* __restrict__ px,
                                                            const int* __restrict__ smp, const int* __restrict__ bounce,
                                                            rt_scatter* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char lds[];
  Ctx<FEAT> X = make_ctx<FEAT>(S, lds);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ray r{mk(orig3[3 * i], orig3[3 * i + 1], orig3[3 * i + 2]), mk(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]),
        time ? time[i] : 0.f};
  Rng g{(uint32_t)px[i], (uint32_t)smp[i], seed_lo, seed_hi};
  const uint32_t b = (uint32_t)bounce[i];
  Tally<false> tally;
  Rec rec;
  rt_scatter s;
  s.hit = 0; s.ok = 0; s.mat_id = -1; s.pad = 0;
  s.orig[0] = s.orig[1] = s.orig[2] = 0.f; s.dir[0] = s.dir[1] = s.dir[2] = 0.f; s.time = 0.f;
  s.attenuation[0] = s.attenuation[1] = s.attenuation[2] = 0.f;
  s.emitted[0] = s.emitted[1] = s.emitted[2] = 0.f;
  if (closest_hit(X, r, 0.001f, __builtin_inff(), g, b, nullptr, false, false, rec, tally)) {
    V3 att, em;
    const bool ok = scatter<false>(X, rec, r, g, b, mk(0.f, 0.f, 0.f), att, em, tally);
    s.hit = 1; s.ok = ok ? 1 : 0; s.mat_id = rec.mat;
    if (ok) {
      s.orig[0] = r.o.x; s.orig[1] = r.o.y; s.orig[2] = r.o.z;
      s.dir[0] = r.d.x; s.dir[1] = r.d.y; s.dir[2] = r.d.z;
      s.time = r.tm;
      s.attenuation[0] = att.x; s.attenuation[1] = att.y; s.attenuation[2] = att.z;
    } else {
      s.emitted[0] = em.x; s.emitted[1] = em.y; s.emitted[2] = em.z;
    }
  }
  out[i] = s;
}

template <uint32_t FEAT>
__global__ __launch_bounds__(RT_BLOCK) void k_sample_batch(DScene S, DCamera cam, DRender R, int n,
                                                           const int* __restrict__ px, const int* __restrict__ smp,
                                                           float* __restrict__ out_rgb) {
  extern __shared__ __align__(16) unsigned char lds[];
  Ctx<FEAT> X = make_ctx<FEAT>(S, lds);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Tally<false> tally;
  Rng g{(uint32_t)px[i], (uint32_t)smp[i], R.seed_lo, R.seed_hi};
  V3 L = mk(0.f, 0.f, 0.f);
  if (R.max_depth > 0) {
    Ray ray = camera_ray(&cam, px[i] % R.width, px[i] / R.width, g, tally);
    V3 beta = mk(1.f, 1.f, 1.f);
    uint32_t bounce = 0;
    while (!path_step(X, R, ray, beta, bounce, g, L, tally)) {}
  }
  out_rgb[3 * i] = L.x; out_rgb[3 * i + 1] = L.y; out_rgb[3 * i + 2] = L.z;
}

// ------------------------------------------------------------------------------ launchers
// Kernel instantiations, smallest first. A scene runs on the first one whose mask covers its
// features: fewer features = fewer registers = more waves per SIMD.
//   V_RECT   axis-aligned rects under translate/rotate-y chains, solid-colour Lambertians and lights
//            (the Cornell box)
//   V_SMOKE  + constant media with isotropic phase functions (the Cornell smoke box)
//   V_LIST   spheres and rects in plain lists (no tree), metal / dielectric, checker and image textures
//            (the two checkered spheres, the earth)
//   V_SPH    + BVHs: everything of the first book (the random-spheres world)
//   V_NOISE  spheres and rects at the top level with Perlin-noise and checker textures (the Perlin
//            spheres, the lit Perlin sphere)
//   V_ALL    everything the reference's worlds use (V_ALL_DEEP: the instantiation that renders it)
//   V_FULL   + media whose boundary is any hittable (V_ALL answers one-sphere and one-box boundaries)
//   V_BALLS  spheres alone, in trees, solid colours and checkers: no rect tests or records, no image textures (the random-spheres world)
constexpr uint32_t V_RECT = F_CHAINS | F_SMALL | F_RECTS;
constexpr uint32_t V_SMOKE = F_CHAINS | F_MEDIA | F_SMALL | F_RECTS;
// (moving spheres: only in the kernels that walk trees and the all-features ones. Without the time interpolation in the
// sphere test and record the list kernel is 6 % faster on the earth and the two checkered spheres, the noise kernel
// 3-4 % on the Perlin worlds; a list world that does have a moving sphere renders on V_SPH, a noise world on V_ALL_DEEP.)
constexpr uint32_t V_LIST = F_SPHERES | F_SPECULAR | F_TEXTURES | F_IMAGE | F_RECTS;
constexpr uint32_t V_BALLS = F_SPHERES | F_SPECULAR | F_BVH | F_TEXTURES;
constexpr uint32_t V_SPH = V_LIST | F_BVH | F_MOVING;
constexpr uint32_t V_NOISE = F_SPHERES | F_TEXTURES | F_NOISE | F_RECTS;
constexpr uint32_t V_ALL = F_ALL;
constexpr uint32_t V_ALL_DEEP = F_ALL | F_DEEP;  // everything, for worlds with BVHs
constexpr uint32_t V_FULL = F_ALL | F_MEDIA_ANY;

int pick_variant(const DScene& S) {
  const bool small = S.n_prims <= RT_LDS_PRIMS && S.n_chains <= RT_LDS_CHAINS;
  if (S.features & F_MEDIA_ANY) return 6;
  if ((S.features & F_BVH) && (S.features & ~V_BALLS) == 0) return 7;
  if (small && (S.features & ~V_RECT) == 0) return 0;
  if (small && (S.features & ~V_SMOKE) == 0) return 1;
  if ((S.features & ~V_LIST) == 0) return 2;
  if ((S.features & ~V_SPH) == 0) return 3;
  return (S.features & ~V_NOISE) == 0 ? 4 : 5;
}

// LDS per workgroup of instantiation `variant` (the F_SMALL ones, 0 and 1, keep the record tables there).
size_t variant_lds_bytes(const DScene& S, int block, int variant) {
  size_t vec_bytes = (size_t)S.n_perlin * 4096;
  size_t perm_bytes = ((size_t)S.n_perlin * 768 + 15) & ~(size_t)15;
  size_t info_bytes = variant < 2 ? (size_t)S.n_prims * sizeof(DPrimInfo) + (size_t)S.n_chains * sizeof(DChainRec) : 0;
  const size_t node_bytes = (variant == 3 || variant >= 5) ? (size_t)S.lds_nodes * 16 : 0;
  const bool walks = variant == 3 || variant >= 5;
  const size_t xch_bytes = (walks && S.xch_on_stacks) ? 0 : (size_t)(block / 64) * 1024;
  return vec_bytes + perm_bytes + info_bytes + xch_bytes + (size_t)block * (size_t)S.bvh_stack * 4 + node_bytes;
}
size_t shade_lds_bytes(const DScene& S, int block) { return variant_lds_bytes(S, block, pick_variant(S)); }

}  // namespace rt
#include "rt_pool.hpp"
namespace rt {

// k_render_pool exists for the two instantiations worlds with trees select; its 16-bit stack entries
// and 8-bit stack depths bound the trees it takes.
bool pool_kernel_covers(const DScene& S) {
  const int v = pick_variant(S);
  return (v == 3 || v == 5) && S.first_bvh_group >= 0 && S.n_bvh_nodes < 32768 && S.n_prims < 32768 &&
         S.bvh_stack < 256 && S.n_groups < (1 << 23);
}
size_t pool_lds_bytes(const DScene& S, int block) {
  const size_t vec_bytes = (size_t)S.n_perlin * 4096;
  const size_t perm_bytes = ((size_t)S.n_perlin * 768 + 15) & ~(size_t)15;
  return vec_bytes + perm_bytes + (size_t)(block / 64) * 1024 + (size_t)(block / 64) * pool_wave_bytes(S.bvh_stack);
}
template <class F>
auto with_pool_kernel(const DScene& S, bool count, F&& f) {
  if (pick_variant(S) == 3) return count ? f(k_render_pool<V_SPH, true>) : f(k_render_pool<V_SPH, false>);
  return count ? f(k_render_pool<V_ALL, true>) : f(k_render_pool<V_ALL_DEEP, false>);
}

template <class F>
auto with_render_kernel(const DScene& S, bool count, F&& f) {
  switch (pick_variant(S) * 2 + (count ? 1 : 0)) {
    case 0: return f(k_render<V_RECT, false>);
    case 1: return f(k_render<V_RECT, true>);
    case 2: return f(k_render<V_SMOKE, false>);
    case 3: return f(k_render<V_SMOKE, true>);
    case 4: return f(k_render<V_LIST, false>);
    case 5: return f(k_render<V_LIST, true>);
    case 6: return f(k_render<V_SPH, false>);
    case 7: return f(k_render<V_SPH, true>);
    case 8: return f(k_render<V_NOISE, false>);
    case 9: return f(k_render<V_NOISE, true>);
    case 10: return f(k_render<V_ALL_DEEP, false>);
    case 11: return f(k_render<V_ALL, true>);
    case 12: return f(k_render<V_FULL, false>);
    case 13: return f(k_render<V_FULL, true>);
    case 14: return f(k_render<V_BALLS, false>);
    default: return f(k_render<V_BALLS, true>);
  }
}

hipError_t allow_lds(size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  hipError_t e = hipSuccess;
  auto allow = [&](auto kernel) {
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  };
#define RT_ALLOW(MASK) allow(k_render<MASK, false>); allow(k_render<MASK, true>); allow(k_hit_batch<MASK>); allow(k_sample_batch<MASK>); allow(k_scatter_batch<MASK>);
  RT_ALLOW(V_RECT) RT_ALLOW(V_SMOKE) RT_ALLOW(V_LIST) RT_ALLOW(V_SPH) RT_ALLOW(V_NOISE) RT_ALLOW(V_FULL) RT_ALLOW(V_BALLS)
#undef RT_ALLOW
  allow(k_render<V_ALL_DEEP, false>); allow(k_render<V_ALL, true>); allow(k_hit_batch<V_ALL_DEEP>);
  allow(k_sample_batch<V_ALL>); allow(k_scatter_batch<V_ALL>);
  allow(k_render_pool<V_SPH, false>); allow(k_render_pool<V_SPH, true>);
  allow(k_render_pool<V_ALL_DEEP, false>); allow(k_render_pool<V_ALL, true>);
  return e;
}

hipError_t launch_render(const DScene& S, const DCamera& cam, const DRender& R, float* out, uint32_t* queue,
                         unsigned long long* stats, bool count, bool pool, int grid, hipStream_t stream) {
  (void)hipGetLastError();  // an error left behind by an unrelated earlier call is not this launch's
  if (pool) {
    const size_t lds = pool_lds_bytes(S, RT_BLOCK);
    return with_pool_kernel(S, count, [&](auto kernel) {
      RenderArgs A;
      A.S = S; A.cam = cam; A.R = R; A.out = out; A.queue = queue; A.stats = stats;
      A.S.lds_nodes = 0;  // the pool has its own LDS layout and its own walk
      A.S.xch_on_stacks = 0;
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(RT_BLOCK), lds, stream, A);
      return hipGetLastError();
    });
  }
  size_t lds = shade_lds_bytes(S, RT_BLOCK);
  return with_render_kernel(S, count, [&](auto kernel) {
    RenderArgs A;
    A.S = S; A.cam = cam; A.R = R; A.out = out; A.queue = queue; A.stats = stats;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(RT_BLOCK), lds, stream, A);
    return hipGetLastError();
  });
}

hipError_t render_occupancy(const DScene& S, bool count, bool pool, int* blocks_per_cu) {
  if (pool) {
    const size_t lds = pool_lds_bytes(S, RT_BLOCK);
    return with_pool_kernel(S, count, [&](auto kernel) {
      return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, kernel, RT_BLOCK, lds);
    });
  }
  size_t lds = shade_lds_bytes(S, RT_BLOCK);
  return with_render_kernel(S, count, [&](auto kernel) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, kernel, RT_BLOCK, lds);
  });
}

hipError_t launch_resolve(const DRender& R, const float* partial, float* rgb_sum, uint32_t* argb,
                          hipStream_t stream) {
  (void)hipGetLastError();  // an error left behind by an unrelated earlier call is not this launch's
  int npix = R.width * R.height;
  hipLaunchKernelGGL(k_resolve, dim3((npix + 255) / 256), dim3(256), 0, stream, R, partial, rgb_sum, argb);
  return hipGetLastError();
}

hipError_t launch_tonemap(const float* rgb_sum, int n_pixels, int spp, uint32_t* argb, hipStream_t stream) {
  (void)hipGetLastError();  // an error left behind by an unrelated earlier call is not this launch's
  hipLaunchKernelGGL(k_tonemap, dim3((n_pixels + 255) / 256), dim3(256), 0, stream, rgb_sum, n_pixels, spp, argb);
  return hipGetLastError();
}

// Does instantiation `variant` (0 V_RECT .. 7 V_BALLS, the order of `masks` below) support everything the scene needs?
bool variant_covers(const DScene& S, int variant) {
  const bool small = S.n_prims <= RT_LDS_PRIMS && S.n_chains <= RT_LDS_CHAINS;
  const uint32_t masks[RT_N_VARIANTS] = {V_RECT, V_SMOKE, V_LIST, V_SPH, V_NOISE, V_ALL_DEEP, V_FULL, V_BALLS};
  if (variant < 0 || variant >= RT_N_VARIANTS) return false;
  if ((masks[variant] & F_SMALL) && !small) return false;
  return (S.features & ~masks[variant]) == 0;
}
int render_variant(const DScene& S) { return pick_variant(S); }

hipError_t launch_hit_batch(const DScene& S, int variant, int n, const float* orig3, const float* dir3,
                            const float* time, float tmin, float tmax, const float* medium_xi, rt_hit* out,
                            hipStream_t stream) {
  (void)hipGetLastError();  // an error left behind by an unrelated earlier call is not this launch's
  const size_t lds = variant_lds_bytes(S, RT_BLOCK, variant);
  dim3 grid((n + RT_BLOCK - 1) / RT_BLOCK), block(RT_BLOCK);
#define RT_HIT_CASE(V, MASK) case V: hipLaunchKernelGGL(k_hit_batch<MASK>, grid, block, lds, stream, S, n, orig3, dir3, time, tmin, tmax, medium_xi, out); break;
  switch (variant) {
    RT_HIT_CASE(0, V_RECT)
    RT_HIT_CASE(1, V_SMOKE)
    RT_HIT_CASE(2, V_LIST)
    RT_HIT_CASE(3, V_SPH)
    RT_HIT_CASE(4, V_NOISE)
    RT_HIT_CASE(5, V_ALL_DEEP)
    RT_HIT_CASE(7, V_BALLS)
    default: hipLaunchKernelGGL(k_hit_batch<V_FULL>, grid, block, lds, stream, S, n, orig3, dir3, time, tmin, tmax, medium_xi, out); break;
  }
#undef RT_HIT_CASE
  return hipGetLastError();
}

hipError_t launch_scatter_batch(const DScene& S, uint32_t seed_lo, uint32_t seed_hi, int n, const float* orig3,
                                const float* dir3, const float* time, const int* px, const int* smp,
                                const int* bounce, rt_scatter* out, hipStream_t stream) {
  (void)hipGetLastError();  // an error left behind by an unrelated earlier call is not this launch's
  size_t lds = shade_lds_bytes(S, RT_BLOCK);
  dim3 grid((n + RT_BLOCK - 1) / RT_BLOCK), block(RT_BLOCK);
#define RT_SC_CASE(V, MASK) case V: hipLaunchKernelGGL(k_scatter_batch<MASK>, grid, block, lds, stream, S, seed_lo, seed_hi, n, orig3, dir3, time, px, smp, bounce, out); break;
  switch (pick_variant(S)) {  // the same instantiation the render would use
    RT_SC_CASE(0, V_RECT)
    RT_SC_CASE(1, V_SMOKE)
    RT_SC_CASE(2, V_LIST)
    RT_SC_CASE(3, V_SPH)
    RT_SC_CASE(4, V_NOISE)
    RT_SC_CASE(6, V_FULL)
    RT_SC_CASE(7, V_BALLS)
    default: hipLaunchKernelGGL(k_scatter_batch<V_ALL>, grid, block, lds, stream, S, seed_lo, seed_hi, n, orig3, dir3, time, px, smp, bounce, out); break;
  }
#undef RT_SC_CASE
  return hipGetLastError();
}

hipError_t launch_sample_batch(const DScene& S, const DCamera& cam, const DRender& R, int n, const int* px,
                               const int* smp, float* out_rgb, hipStream_t stream) {
  (void)hipGetLastError();  // an error left behind by an unrelated earlier call is not this launch's
  size_t lds = shade_lds_bytes(S, RT_BLOCK);
  dim3 grid((n + RT_BLOCK - 1) / RT_BLOCK), block(RT_BLOCK);
  switch (pick_variant(S)) {  // the same instantiation the render would use
    case 0: hipLaunchKernelGGL(k_sample_batch<V_RECT>, grid, block, lds, stream, S, cam, R, n, px, smp, out_rgb); break;
    case 1: hipLaunchKernelGGL(k_sample_batch<V_SMOKE>, grid, block, lds, stream, S, cam, R, n, px, smp, out_rgb); break;
    case 2: hipLaunchKernelGGL(k_sample_batch<V_LIST>, grid, block, lds, stream, S, cam, R, n, px, smp, out_rgb); break;
    case 3: hipLaunchKernelGGL(k_sample_batch<V_SPH>, grid, block, lds, stream, S, cam, R, n, px, smp, out_rgb); break;
    case 4: hipLaunchKernelGGL(k_sample_batch<V_NOISE>, grid, block, lds, stream, S, cam, R, n, px, smp, out_rgb); break;
    case 6: hipLaunchKernelGGL(k_sample_batch<V_FULL>, grid, block, lds, stream, S, cam, R, n, px, smp, out_rgb); break;
    case 7: hipLaunchKernelGGL(k_sample_batch<V_BALLS>, grid, block, lds, stream, S, cam, R, n, px, smp, out_rgb); break;
    default: hipLaunchKernelGGL(k_sample_batch<V_ALL>, grid, block, lds, stream, S, cam, R, n, px, smp, out_rgb); break;
  }
  return hipGetLastError();
}

}  // namespace rt
