// rt_pool.hpp — the path loop for worlds with BVHs: live paths are re-packed between the stages
// of a bounce.
//
// In k_render a lane owns one path from camera ray to termination, and every phase of an iteration
// (regeneration, traversal, leaf tests, record, scatter) runs for the whole wave however few lanes
// need it: a BVH walk is as long as its slowest lane, a leaf round waits for every lane to reach a
// leaf, the camera code runs for the third of the lanes whose path just ended. Here a WAVE owns a
// pool of RT_POOL paths kept in LDS (twice its lanes), every path is in one of five states, and the
// wave repeatedly picks the state most paths are in, packs up to 64 of those paths onto its lanes
// (ballot + mbcnt prefix ranks) and runs that stage's code for them:
//
//   GEN   the path ended (or never started): next sample of its work item, or a new item from the
//         queue; camera ray (core.clj:96-102)                                        -> NEW
//   NEW   a ray that has not been tested: the brute-force surface groups             -> WALK
//   WALK  between BVH leaves: internal-node steps only. A lane whose ray reaches a leaf or the end of
//         its tree parks it and takes the next waiting ray at once, so the walk never waits for a
//         slow lane or for a leaf test                                               -> LEAF / POST
//   LEAF  holds a leaf: the primitive test (FP64 for spheres), then pops its stack   -> WALK
//   POST  every group answered: media, hit record, chain unwind, scatter
//         (core.clj:24-30)                                                           -> NEW / GEN
//
// A path's samples are still taken in order by the one pool entry that owns its work item, every
// draw is addressed by (pixel, sample, bounce, purpose), and the stages call the device functions
// k_render and k_sample_batch call: frames are bit-identical to k_render's.
//
// The waves of a workgroup share nothing but the Perlin tables; there is no barrier after start-up.
#ifndef RT_POOL_HPP
#define RT_POOL_HPP

namespace rt {

#ifndef RT_POOL
#define RT_POOL 128           // paths per wave
#endif
#ifndef RT_POOL_REFILL_MIN
#define RT_POOL_REFILL_MIN 8  // WALK: idle lanes wait for this many before the refill code runs
#endif
#ifndef RT_POOL_SUSPEND_BELOW
#define RT_POOL_SUSPEND_BELOW 16  // WALK: with no ray waiting and fewer walkers than this, park them if other stages have work
#endif
#ifndef RT_POOL_MAX_ROUNDS
#define RT_POOL_MAX_ROUNDS (1u << 30)  // scheduling rounds after which a wave gives up (never reached; no hang on a bug)
#endif

enum { PS_GEN = 0, PS_NEW = 1, PS_WALK = 2, PS_LEAF = 3, PS_POST = 4, PS_DONE = 5, PS_COUNT = 6 };
enum {
  PF_OX = 0, PF_OY, PF_OZ, PF_DX, PF_DY, PF_DZ, PF_TM,   // the ray (world space)
  PF_BR, PF_BG, PF_BB,                                     // throughput
  PF_AR, PF_AG, PF_AB,                                     // partial sum of the work item
  PF_BOUNCE, PF_PXY, PF_ITEM, PF_K1, PF_K1END,             // bookkeeping as in k_render
  PF_BT, PF_BPRIM, PF_ROOTLO, PF_ROOTHI,                   // closest hit so far
  PF_NODE, PF_SPG,                                         // walk: node reference, stack depth | group << 8
  PF_STATE,
  PF_COUNT
};
#define RT_POOL_NO_ITEM 0xFFFFFFFFu

// LDS of one wave's pool: fields (SoA, PF_COUNT x RT_POOL dwords), the compaction list (RT_POOL
// dwords), the per-path traversal stacks (depth x RT_POOL halfwords).
__host__ __device__ inline size_t pool_wave_bytes(int stack_depth) {
  size_t b = (size_t)(PF_COUNT + 1) * RT_POOL * 4 + (size_t)stack_depth * RT_POOL * 2;
  return (b + 15) & ~(size_t)15;
}

struct Pool {
  uint32_t RT_LDS* f;
  uint32_t RT_LDS* list;
  uint16_t RT_LDS* stack;
  RT_DEV uint32_t ld(int field, int e) const { return f[field * RT_POOL + e]; }
  RT_DEV float ldf(int field, int e) const { return __uint_as_float(f[field * RT_POOL + e]); }
  RT_DEV void st(int field, int e, uint32_t v) const { f[field * RT_POOL + e] = v; }
  RT_DEV void stf(int field, int e, float v) const { f[field * RT_POOL + e] = __float_as_uint(v); }
  RT_DEV Ray ray(int e) const {
    return Ray{mk(ldf(PF_OX, e), ldf(PF_OY, e), ldf(PF_OZ, e)), mk(ldf(PF_DX, e), ldf(PF_DY, e), ldf(PF_DZ, e)), ldf(PF_TM, e)};
  }
  RT_DEV Best best(int e) const {
    Best b;
    b.t = ldf(PF_BT, e);
    b.prim = (int)ld(PF_BPRIM, e);
    uint32_t lo = ld(PF_ROOTLO, e), hi = ld(PF_ROOTHI, e);
    b.root = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    return b;
  }
  RT_DEV void put_best(int e, const Best& b) const {
    stf(PF_BT, e, b.t);
    st(PF_BPRIM, e, (uint32_t)b.prim);
    const unsigned long long bits = (unsigned long long)__double_as_longlong(b.root);
    st(PF_ROOTLO, e, (uint32_t)bits);
    st(PF_ROOTHI, e, (uint32_t)(bits >> 32));
  }
  // node references on the 16-bit stacks: internal nodes as they are (< 32768), leaves as 0x8000 | primitive
  // (the type a leaf reference carries beside the index is dropped here and read back from the primitive)
  RT_DEV void push(int e, int level, int ref) const {
    stack[level * RT_POOL + e] = (uint16_t)(ref >= 0 ? ref : (0x8000 | (~ref & RT_LEAF_INDEX_MASK)));
  }
  RT_DEV int pop(int e, int level) const {
    const int r = stack[level * RT_POOL + e];
    return (r & 0x8000) ? ~(r & 0x7fff) : r;
  }
};

// Everything made visible to the other lanes of the wave before they read it (LDS operations of one
// wave complete in order; this keeps the compiler from moving them across).
RT_DEV void pool_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <uint32_t FEAT, bool COUNT>
__global__ __launch_bounds__(RT_BLOCK, 2) void k_render_pool(RenderArgs A) {
  extern __shared__ __align__(16) unsigned char lds[];
  Ctx<FEAT> X = make_ctx<FEAT>(A.S, lds);
  const DScene& S = X.S;
  const DRender& R = A.R;
  float* __restrict__ out = A.out;
  unsigned long long* __restrict__ stats = A.stats;
  Tally<COUNT> tally;
  if constexpr (COUNT) tally.clear();
  const bool clock_wave = blockIdx.x == 0 && threadIdx.x == 0;
  if (clock_wave) {
    stats[RT_STATS_CLOCK + 0] = __builtin_readcyclecounter();
    stats[RT_STATS_CLOCK + 2] = __builtin_amdgcn_s_memrealtime();
  }
  const uint32_t lane = wave_lane();
  const float INF = __builtin_inff();
  const float TMIN = 0.001f;

  // this wave's pool, behind the areas make_ctx laid out (Perlin tables, one exchange area per wave)
  Pool Q;
  {
    const size_t vec_bytes = (size_t)S.n_perlin * 4096;
    const size_t perm_bytes = ((size_t)S.n_perlin * 768 + 15) & ~(size_t)15;
    const size_t xch_bytes = (size_t)(blockDim.x >> 6) * 1024;
    unsigned char* base = lds + vec_bytes + perm_bytes + xch_bytes + (size_t)(threadIdx.x >> 6) * pool_wave_bytes(S.bvh_stack);
    Q.f = (uint32_t RT_LDS*)base;
    Q.list = Q.f + PF_COUNT * RT_POOL;
    Q.stack = (uint16_t RT_LDS*)(Q.list + RT_POOL);
  }
  for (int e = (int)lane; e < RT_POOL; e += 64) {
    Q.st(PF_STATE, e, PS_GEN);
    Q.st(PF_ITEM, e, RT_POOL_NO_ITEM);
    Q.st(PF_K1, e, 0u);
    Q.st(PF_K1END, e, 1u);
  }
  pool_sync();

  // the wave's window of the work-item queue (as in k_render)
  uint32_t q_next = 0, q_end = 0;
  uint32_t q_head = blockIdx.x % RT_QUEUE_HEADS, q_heads_dry = 0;
  bool q_dry = false;
  int q_tx0 = 0, q_ty0 = 0, q_s0 = 0, q_s1 = 0;
  unsigned long long n_rays = 0, n_samples = 0;
  uint32_t prio_tick = __builtin_amdgcn_s_getreg(6148);

#ifdef RT_POOL_STAMPS
  unsigned long long sc_cyc[PS_COUNT] = {0, 0, 0, 0, 0, 0}, sc_rounds[PS_COUNT] = {0, 0, 0, 0, 0, 0}, sc_n[PS_COUNT] = {0, 0, 0, 0, 0, 0};
  unsigned long long sc_census = 0, sc_walk_iters = 0, sc_walk_lanes = 0, ts_a = 0, ts_b = 0, ts_c = 0;
#endif
  for (uint32_t round = 0; round < RT_POOL_MAX_ROUNDS; ++round) {
#ifdef RT_POOL_STAMPS
    __builtin_amdgcn_sched_barrier(0); ts_a = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
    switch (prio_tick++ & 3u) {
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
    // ---- census: how many paths are in each state (wave-uniform counts)
    uint32_t st_of[(RT_POOL + 63) / 64];
    int cnt[PS_COUNT];
#pragma unroll
    for (int s = 0; s < PS_COUNT; ++s) cnt[s] = 0;
#pragma unroll
    for (int k = 0; k < (RT_POOL + 63) / 64; ++k) {
      const int e = k * 64 + (int)lane;
      st_of[k] = e < RT_POOL ? Q.ld(PF_STATE, e) : (uint32_t)PS_COUNT;
#pragma unroll
      for (int s = 0; s < PS_COUNT; ++s) cnt[s] += __popcll(__ballot(st_of[k] == (uint32_t)s));
    }
    if (cnt[PS_DONE] == RT_POOL) break;
    // ---- the stage most paths wait for (ties: the later stage, so that paths finish before new ones start)
    int stage = PS_GEN, most = cnt[PS_GEN];
#pragma unroll
    for (int s = PS_NEW; s <= PS_POST; ++s)
      if (cnt[s] >= most && cnt[s] > 0) { most = cnt[s]; stage = s; }
    // ---- pack the paths of that state: list[0, n)
    int n = 0;
#pragma unroll
    for (int k = 0; k < (RT_POOL + 63) / 64; ++k) {
      const bool m = st_of[k] == (uint32_t)stage;
      const unsigned long long b = __ballot(m);
      if (m) Q.list[n + (int)lane_rank(b)] = (uint32_t)(k * 64 + (int)lane);
      n += __popcll(b);
    }
    pool_sync();
#ifdef RT_POOL_STAMPS
    __builtin_amdgcn_sched_barrier(0); ts_b = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
    const bool mine = (int)lane < n;               // (WALK takes its paths from the list itself)
    const int e = mine ? (int)Q.list[lane] : 0;

    if (stage == PS_GEN) {
      // ------------------------------------------------------------------ next sample / next item
      uint32_t item = RT_POOL_NO_ITEM, k1 = 0, k1_end = 1, pxy = 0;
      if (mine) { item = Q.ld(PF_ITEM, e); k1 = Q.ld(PF_K1, e); k1_end = Q.ld(PF_K1END, e); pxy = Q.ld(PF_PXY, e); }
      bool have_item = mine && item != RT_POOL_NO_ITEM;
      const bool need = mine && !(have_item && k1 + 1u != k1_end);
      if (need && have_item) {
        // core.clj:115-120 finished for this chunk: store the partial sum of its samples
        const size_t o = (size_t)item * 3;
        out[o] = Q.ldf(PF_AR, e); out[o + 1] = Q.ldf(PF_AG, e); out[o + 2] = Q.ldf(PF_AB, e);
        have_item = false;
        Q.st(PF_ITEM, e, RT_POOL_NO_ITEM);
      }
      const unsigned long long mask = __ballot(need);
      bool done = false;
      if (mask != 0ull) {  // wave-uniform
        const RenderArgs RT_CONST* C = cold();
        if (q_next == q_end && !q_dry) {
          const uint32_t n_windows = (uint32_t)C->R.n_items >> 6;
          uint32_t base = ~0u;
          while (q_heads_dry < RT_QUEUE_HEADS) {  // wave-uniform
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd(C->queue + q_head * RT_QUEUE_STRIDE, 1u);
            k = __builtin_amdgcn_readfirstlane(k);
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
        const uint32_t avail = q_end - q_next;
        const uint32_t want = (uint32_t)__popcll(mask);
        if (need) {
          const uint32_t rank = lane_rank(mask);
          if (rank < avail) {
            const int in_tile = (int)((q_next & 63u) + rank);
            const int px = q_tx0 + (in_tile & 7);
            const int py = q_ty0 + (in_tile >> 3);
            if (px < R.width && py < R.height) {
              pxy = (uint32_t)px | ((uint32_t)py << 16);
              item = q_next + rank;
              k1 = (uint32_t)(R.sample_offset + q_s0) - 1u;
              k1_end = (uint32_t)(R.sample_offset + q_s1);
              Q.st(PF_PXY, e, pxy); Q.st(PF_ITEM, e, item); Q.st(PF_K1, e, k1); Q.st(PF_K1END, e, k1_end);
              Q.stf(PF_AR, e, 0.f); Q.stf(PF_AG, e, 0.f); Q.stf(PF_AB, e, 0.f);
              have_item = true;
            }
          } else if (q_dry) {
            done = true;
          }
        }
        q_next += min(want, avail);
      }
      if (done) Q.st(PF_STATE, e, PS_DONE);
      const bool fresh = have_item && k1 + 1u != k1_end;
      n_samples += (unsigned long long)__popcll(__ballot(fresh));
      if (fresh) {
        // core.clj:96-105: a fresh camera path for the item's next sample of this pixel
        k1 += 1u;
        Q.st(PF_K1, e, k1);
        if (R.max_depth <= 0) {
          tally.template add<RT_EV_ACCUMULATE>();  // ray-color returns black at depth 0: the path is over
        } else {
          const int px = (int)(pxy & 0xFFFFu), py = (int)(pxy >> 16);
          const Rng g{(uint32_t)(py * R.width + px), k1, R.seed_lo, R.seed_hi};
          const RenderArgs RT_CONST* C = cold();
          const Ray ray = camera_ray(&C->cam, px, py, g, tally);
          Q.stf(PF_OX, e, ray.o.x); Q.stf(PF_OY, e, ray.o.y); Q.stf(PF_OZ, e, ray.o.z);
          Q.stf(PF_DX, e, ray.d.x); Q.stf(PF_DY, e, ray.d.y); Q.stf(PF_DZ, e, ray.d.z);
          Q.stf(PF_TM, e, ray.tm);
          Q.stf(PF_BR, e, 1.f); Q.stf(PF_BG, e, 1.f); Q.stf(PF_BB, e, 1.f);
          Q.st(PF_BOUNCE, e, 0u);
          Q.st(PF_STATE, e, PS_NEW);
        }
      }
    } else if (stage == PS_NEW) {
      // ------------------------------------------------------------------ core.clj:23, the groups without a tree
      n_rays += (unsigned long long)__popcll(__ballot(mine));
      if (mine) {
        const Ray wr = Q.ray(e);
        Best best{INF, -1, 0.0};
        const int n_surface = S.n_surface_groups;
        for (int gi = 0; gi < n_surface; ++gi) {  // wave-uniform
          const DGroup RT_CONST& G = cptr(S.groups)[gi];
          if (G.bvh_root >= 0) continue;
          const RayX r = group_ray<Ctx<FEAT>>(S, gi, G.flags, G.rc, G.rs, G.bx, G.by, G.bz, wr, tally);
          brute_group_hit<Ctx<FEAT>>(S, G, r, TMIN, best, tally, (double)TMIN);
        }
        Q.put_best(e, best);
        const int g0 = S.first_bvh_group;
        Q.st(PF_NODE, e, (uint32_t)cptr(S.groups)[g0].bvh_root);
        Q.st(PF_SPG, e, (uint32_t)g0 << 8);
        Q.st(PF_STATE, e, PS_WALK);
      }
    } else if (stage == PS_WALK) {
      // ------------------------------------------------------------------ internal nodes only, lanes refilled as they fall idle
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      const u4 RT_CONST* nodes4 = reinterpret_cast<const u4 RT_CONST*>(cptr(S.nodes));
      asm volatile("" : "+s"(nodes4));
      // Walkers are parked only when another stage has clearly more paths waiting than there are
      // walkers left: the next census then picks that stage, so every round makes progress.
      const int other_most = max(cnt[PS_GEN], cnt[PS_NEW]);
      int to_leaf = cnt[PS_LEAF], to_post = cnt[PS_POST];  // with the paths this run moves there
      int wnext = 0;
      bool have = false, init = false;
      int we = 0, node = RT_BVH_DONE, sp = 0, g = 0;
      float bt = INF;
      V3 inv = mk(0.f, 0.f, 0.f), oi = mk(0.f, 0.f, 0.f);
      for (;;) {
        const unsigned long long freem = __ballot(!have);
        if (wnext < n && (__popcll(freem) >= RT_POOL_REFILL_MIN)) {  // wave-uniform
          const int rank = (int)lane_rank(freem);
          if (!have && wnext + rank < n) {
            we = (int)Q.list[wnext + rank];
            node = (int)Q.ld(PF_NODE, we);
            const uint32_t spg = Q.ld(PF_SPG, we);
            sp = (int)(spg & 255u); g = (int)(spg >> 8);
            bt = Q.ldf(PF_BT, we);
            have = true; init = true;
          }
          wnext += min(__popcll(freem), n - wnext);
        }
        const unsigned long long havem = __ballot(have);
        if (havem == 0ull) {
          if (wnext >= n) break;
          continue;  // (never: with no walker every lane is free and the refill above ran)
        }
        if (wnext >= n && __popcll(havem) < RT_POOL_SUSPEND_BELOW && max(other_most, max(to_leaf, to_post)) >= 2 * RT_POOL_SUSPEND_BELOW) {
          // too few walkers left to be worth the wave: park them where they are
          if (have) { Q.st(PF_NODE, we, (uint32_t)node); Q.st(PF_SPG, we, (uint32_t)sp | ((uint32_t)g << 8)); }
          break;
        }
#ifdef RT_POOL_STAMPS
        sc_walk_iters += 1; sc_walk_lanes += (unsigned long long)__popcll(havem);
#endif
        // a tree is finished: on to the world's next one, or the search is over
        const bool tree_done = have && node == RT_BVH_DONE;
        if (__ballot(tree_done) != 0ull) {
          if (tree_done) {
            const int g2 = cptr(S.bvh_next)[g];
            if (g2 < 0) {
              Q.st(PF_STATE, we, PS_POST);
              have = false;
            } else {
              g = g2; node = cptr(S.groups)[g2].bvh_root; sp = 0; init = true;
            }
          }
          to_post += __popcll(__ballot(tree_done && !have));
        }
        // the ray in the space of group g (identity for the world's own lists)
        if (__ballot(have && init) != 0ull) {
          if (have && init) {
            const Ray wr = Q.ray(we);
            const DGroup RT_CONST& G = cptr(S.groups)[g];
            const RayX r = group_ray<Ctx<FEAT>>(S, g, G.flags, G.rc, G.rs, G.bx, G.by, G.bz, wr, tally);
            inv = mk(__builtin_amdgcn_fmed3f(r.inv.x, -1e30f, 1e30f), __builtin_amdgcn_fmed3f(r.inv.y, -1e30f, 1e30f),
                     __builtin_amdgcn_fmed3f(r.inv.z, -1e30f, 1e30f));
            oi = mk(r.o.x * inv.x, r.o.y * inv.y, r.o.z * inv.z);
            init = false;
          }
        }
        // one step
        if (have && (unsigned)node < (unsigned)RT_BVH_DONE) {
          const u4 q0 = nodes4[node * 2 + 0];
          const u4 q1 = nodes4[node * 2 + 1];
          const uint32_t w0 = q0.x, w1 = q0.y, w2 = q0.z, w3 = q0.w, w4 = q1.x, w5 = q1.y;
          const h2 a0 = __builtin_bit_cast(h2, w0), a1 = __builtin_bit_cast(h2, w1), a2 = __builtin_bit_cast(h2, w2);
          const h2 a3 = __builtin_bit_cast(h2, w3), a4 = __builtin_bit_cast(h2, w4), a5 = __builtin_bit_cast(h2, w5);
          const int left = (int)q1.z, right = (int)q1.w;
          float tl, tr;
          tally.template add<RT_EV_AABB_TEST>(2);
          const bool hl = slab16(a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, inv, oi, TMIN, bt, tl);
          const bool hr = slab16(a3.x, a3.y, a4.x, a4.y, a5.x, a5.y, inv, oi, TMIN, bt, tr);
          if (hl && hr) {
            const bool swap = tr < tl;
            Q.push(we, sp, swap ? left : right);
            sp += 1;
            node = swap ? right : left;
          } else if (hl || hr) {
            node = hl ? left : right;
          } else if (sp > 0) {
            sp -= 1;
            node = Q.pop(we, sp);
          } else {
            node = RT_BVH_DONE;
          }
        }
        // a leaf: park the path for the LEAF stage, the lane is free
        const bool at_leaf = have && node < 0;
        if (__ballot(at_leaf) != 0ull) {
          if (at_leaf) {
            Q.st(PF_NODE, we, (uint32_t)node);
            Q.st(PF_SPG, we, (uint32_t)sp | ((uint32_t)g << 8));
            Q.st(PF_STATE, we, PS_LEAF);
            have = false;
          }
          to_leaf += __popcll(__ballot(at_leaf));
        }
      }
    } else if (stage == PS_LEAF) {
      // ------------------------------------------------------------------ the primitive of a leaf, then the next node off the stack
      if (mine) {
        const Ray wr = Q.ray(e);
        const int leaf = (int)Q.ld(PF_NODE, e);
        const uint32_t spg = Q.ld(PF_SPG, e);
        int sp = (int)(spg & 255u);
        const int g = (int)(spg >> 8);
        Best best = Q.best(e);
        const DGroup RT_CONST& G = cptr(S.groups)[g];
        const RayX r = group_ray<Ctx<FEAT>>(S, g, G.flags, G.rc, G.rs, G.bx, G.by, G.bz, wr, tally);
        const int lidx = ~leaf & RT_LEAF_INDEX_MASK;
        test_prim_any_order<Ctx<FEAT>>(S, lidx | (prim_type(cptr(S.prims)[lidx].type) << RT_LEAF_TYPE_SHIFT), r, TMIN, best, tally, (double)TMIN);
        Q.put_best(e, best);
        int node = RT_BVH_DONE;
        if (sp > 0) { sp -= 1; node = Q.pop(e, sp); }
        Q.st(PF_NODE, e, (uint32_t)node);
        Q.st(PF_SPG, e, (uint32_t)sp | ((uint32_t)g << 8));
        Q.st(PF_STATE, e, PS_WALK);
      }
    } else {
      // ------------------------------------------------------------------ media, record, scatter (core.clj:24-30)
      Ray ray{mk(0, 0, 0), mk(0, 0, 1), 0.f};
      Best best{INF, -1, 0.0};
      V3 beta = mk(1.f, 1.f, 1.f);
      uint32_t bounce = 0;
      Rng g{0u, 0u, R.seed_lo, R.seed_hi};
      if (mine) {
        ray = Q.ray(e);
        best = Q.best(e);
        beta = mk(Q.ldf(PF_BR, e), Q.ldf(PF_BG, e), Q.ldf(PF_BB, e));
        bounce = Q.ld(PF_BOUNCE, e);
        const uint32_t pxy = Q.ld(PF_PXY, e);
        g.k0 = (pxy >> 16) * (uint32_t)R.width + (pxy & 0xFFFFu);
        g.k1 = Q.ld(PF_K1, e);
        media_hit(X, ray, TMIN, g, bounce, nullptr, best, tally);
      }
      const bool hit = mine && best.prim != -1;
      Rec rec;
      rec.mtype = -1;
      if (hit) hit_record(X, ray, best, false, rec, tally);
      V3 sv = mk(0.f, 0.f, 0.f);
      if constexpr (wave_draw(FEAT)) sv = wave_random_in_unit_sphere(hit && needs_sphere_draw(rec.mtype), g, bounce, X.lds_xch, X.xch_stride, tally);  // (its area is in one piece)
      if (mine) {
        V3 L;
        if (path_finish<wave_draw(FEAT)>(X, R, hit, rec, sv, ray, beta, bounce, g, L, tally)) {
          // core.clj:120
          Q.stf(PF_AR, e, Q.ldf(PF_AR, e) + L.x); Q.stf(PF_AG, e, Q.ldf(PF_AG, e) + L.y); Q.stf(PF_AB, e, Q.ldf(PF_AB, e) + L.z);
          tally.template add<RT_EV_ACCUMULATE>();
          Q.st(PF_STATE, e, PS_GEN);
        } else {
          Q.stf(PF_OX, e, ray.o.x); Q.stf(PF_OY, e, ray.o.y); Q.stf(PF_OZ, e, ray.o.z);
          Q.stf(PF_DX, e, ray.d.x); Q.stf(PF_DY, e, ray.d.y); Q.stf(PF_DZ, e, ray.d.z);
          Q.stf(PF_BR, e, beta.x); Q.stf(PF_BG, e, beta.y); Q.stf(PF_BB, e, beta.z);
          Q.st(PF_BOUNCE, e, bounce);
          Q.st(PF_STATE, e, PS_NEW);
        }
      }
    }
    pool_sync();
#ifdef RT_POOL_STAMPS
    __builtin_amdgcn_sched_barrier(0); ts_c = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
    sc_census += ts_b - ts_a;
#pragma unroll
    for (int s = 0; s < PS_COUNT; ++s)
      if (stage == s) { sc_cyc[s] += ts_c - ts_b; sc_rounds[s] += 1; sc_n[s] += (unsigned long long)min(n, 64); }
#endif
  }
#ifdef RT_POOL_STAMPS
  if (lane == 0) {
    for (int s = 0; s < 5; ++s) { atomicAdd(&stats[2 + s], sc_cyc[s]); atomicAdd(&stats[2 + 5 + s], sc_rounds[s]); atomicAdd(&stats[2 + 10 + s], sc_n[s]); }
    atomicAdd(&stats[2 + 15], sc_census); atomicAdd(&stats[2 + 16], sc_walk_iters); atomicAdd(&stats[2 + 17], sc_walk_lanes);
  }
#endif

  if (clock_wave) {
    stats[RT_STATS_CLOCK + 1] = __builtin_readcyclecounter();
    stats[RT_STATS_CLOCK + 3] = __builtin_amdgcn_s_memrealtime();
  }
  if (lane == 0) { atomicAdd(&stats[0], n_samples); atomicAdd(&stats[1], n_rays); }
  if constexpr (COUNT) {
#pragma unroll
    for (int ev = 0; ev < RT_EV_COUNT; ++ev) {
      unsigned long long w = wave_sum(tally.ev[ev]);
      if (lane == 0 && w) atomicAdd(&stats[2 + ev], w);
    }
  }
}

}  // namespace rt
#endif  // RT_POOL_HPP
