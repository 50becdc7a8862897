// rt_device.hpp — device functions of the path tracer (gfx950, wave64).
//
// Everything the reference's `ray-color` calls, restated for one lane of a wavefront:
// intersection of the flattened world (rt_scene.h), the five materials, the four textures,
// Perlin noise, the samplers and the camera. Arithmetic is FP32 except where the reference's
// FP64 results cannot be reached otherwise (sphere quadratic, sphere u/v, camera ray, Perlin
// cell fractions, checker sign). Each function cites the reference lines it implements
// (paths relative to src/raytracing/).
//
// Every function is templated on a context type X = Ctx<FEAT>. FEAT is a compile-time mask of
// the scene features a kernel instantiation supports; code for absent features is not
// generated, which is what keeps the register count (hence waves per SIMD) of the common
// instantiations low.
#ifndef RT_DEVICE_HPP
#define RT_DEVICE_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtmi.h"
#include "rt_scene.h"

namespace rt {

#define RT_DEV __device__ __forceinline__
// listing build (tools/isa_histogram.py): a comment in the assembly where a phase begins
#ifdef RT_PHASE_MARKS
#define RT_MARK(name) { __builtin_amdgcn_sched_barrier(0); asm volatile("; RTPHASE " #name); __builtin_amdgcn_sched_barrier(0); }
#else
#define RT_MARK(name)
#endif

// Scene tables are read-only for the lifetime of a launch. Reading them through the constant
// address space lets wave-uniform accesses become scalar loads (s_load into SGPRs), so the
// brute-force loops carry their primitive records in scalar registers, not in VGPRs.
#define RT_CONST __attribute__((address_space(4)))
template <class T>
RT_DEV const T RT_CONST* cptr(const T* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (const T RT_CONST*)p;
#pragma clang diagnostic pop
}

RT_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }   // v_rcp_f32, 1 ulp
RT_DEV float rsq(float x) { return __builtin_amdgcn_rsqf(x); }   // v_rsq_f32, 1 ulp

// ----------------------------------------------------------------------------- vectors
struct V3 {
  float x, y, z;
};
RT_DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
RT_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator*(V3 a, float k) { return mk(a.x * k, a.y * k, a.z * k); }
RT_DEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
// The library is compiled with -ffp-contract=off: where a fused multiply-add is wanted it is
// written out, so every instantiation of every kernel rounds the same way.
RT_DEV float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
RT_DEV V3 madd(V3 a, float k, V3 b) { return mk(fmaf(a.x, k, b.x), fmaf(a.y, k, b.y), fmaf(a.z, k, b.z)); }  // a*k + b
RT_DEV float len2(V3 a) { return dot(a, a); }
RT_DEV V3 unit(V3 a) { return a * rsq(len2(a)); }  // base.clj:39-40 (v * 1/|v|)

struct D3 {
  double x, y, z;
};
RT_DEV D3 mkd(double x, double y, double z) { return D3{x, y, z}; }
RT_DEV D3 tod(V3 a) { return mkd(a.x, a.y, a.z); }
RT_DEV V3 tof(D3 a) { return mk((float)a.x, (float)a.y, (float)a.z); }
RT_DEV double ddot(D3 a, D3 b) { return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); }

struct Ray {
  V3 o, d;
  float tm;
};

// ------------------------------------------------------------------------------ counters
// Event tallies for the FLOP accounting build; the production instantiation compiles to nothing.
template <bool ON>
struct Tally;
template <>
struct Tally<false> {
  static constexpr bool on = false;
  template <int E>
  RT_DEV void add(uint32_t = 1) {}
};
template <>
struct Tally<true> {
  static constexpr bool on = true;
  uint32_t ev[RT_EV_COUNT];
  RT_DEV void clear() {
#pragma unroll
    for (int i = 0; i < RT_EV_COUNT; ++i) ev[i] = 0;
  }
  template <int E>
  RT_DEV void add(uint32_t n = 1) { ev[E] += n; }
};

// ---------------------------------------------------------------------------------- RNG
// Philox-4x32 (Salmon et al., SC'11) with 7 rounds, the fewest the paper reports Crush-resistant:
// its 32x32->64 multiplies are the most expensive VALU instructions of the whole path loop. A draw
// is addressed, not sequenced:
//   key     = (seed_lo, seed_hi)
//   counter = (pixel index y*W+x, global sample index, block, bounce << 4 | purpose)
// so its value does not depend on evaluation order or on which GPU runs the sample. The key is the same
// for every lane: its schedule (two additions per round) runs on the scalar unit, and a lane's own
// block 0 has one multiplicand of its first round known at compile time.
// Uniforms are the top 24 bits / 2^24 (16-bit fields for the unit-sphere candidates): exactly
// representable in FP32 and FP64.
enum { P_CAM = 0, P_DISK = 1, P_SPHERE = 2, P_DIEL = 3, P_MEDIUM = 4 };
#define RT_MAX_TRIALS 256  // rejection loops give up here so every lane reaches the exit
#ifndef RT_PHILOX_ROUNDS
#define RT_PHILOX_ROUNDS 7
#endif

struct Rng {
  uint32_t k0, k1, s0, s1;  // k0, k1: the pixel and sample words of the counter (per lane); s0, s1: the key (the seed)
};

RT_DEV void philox(const Rng& g, uint32_t blk, uint32_t bounce, uint32_t purpose, uint32_t out[4]) {
  uint32_t c0 = g.k0, c1 = g.k1, c2 = blk, c3 = (bounce << 4) | purpose;
  uint32_t k0 = g.s0, k1 = g.s1;
#pragma unroll
  for (int r = 0; r < RT_PHILOX_ROUNDS; ++r) {
    // one 32x32->64 multiply per word pair (v_mad_u64_u32)
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    // a ^ b ^ c in one instruction: gfx950's three-input bitwise op, truth table 0x96
    c0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
    c1 = (uint32_t)p1;
    c2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
    c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
RT_DEV float u24(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }

// base.clj:182-186: the reduce seeds its accumulator with draw 0 and never tests it, then
// keeps the first later draw with |v|^2 >= 1 (the inverse of the textbook test). Draw 0 is never
// observable, so the keyed stream starts at draw 1 = candidate 0. One Philox block carries two
// candidates of three 16-bit uniforms each (candidate j: block j >> 1, words 2 (j & 1) and 2 (j & 1) + 1):
// both are evaluated without a branch, which halves the trips through this divergent loop.
RT_DEV V3 cube_point(uint32_t wa, uint32_t wb) {
  // x, y: the two halves of the first word, z: the upper half of the second: 16-bit uniforms at cell centres,
  // (k + 1/2) * 2^-15 - 1 in (-1, 1), symmetric about 0. A half-word converts with one instruction (the conversion
  // reads the half in place), so a coordinate is a conversion and an FMA; 21-bit fields cost a shift or a mask more
  // each, and two of the three straddled words.
  const float S = 3.0517578125e-5f;        // 2^-15
  const float C = -0.9999847412109375f;    // -(1 - 2^-16)
  return mk(fmaf((float)(wa >> 16), S, C), fmaf((float)(wa & 0xFFFFu), S, C), fmaf((float)(wb >> 16), S, C));
}
template <class T>
RT_DEV V3 random_in_unit_sphere(const Rng& g, uint32_t bounce, T& tally) {
  V3 v = mk(0.f, 0.f, 0.f);
  // one block per trip: the loop vectoriser would otherwise run several Philox blocks ahead of the
  // exit test and pay for them in registers
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
  for (uint32_t blk = 0; blk < RT_MAX_TRIALS / 2; ++blk) {
    uint32_t o[4];
    philox(g, blk, bounce, P_SPHERE, o);
    V3 v0 = cube_point(o[0], o[1]), v1 = cube_point(o[2], o[3]);
    bool ok0 = len2(v0) >= 1.0f, ok1 = len2(v1) >= 1.0f;
    tally.template add<RT_EV_SPHERE_TRIAL>(ok0 ? 1 : 2);
    v = ok0 ? v0 : v1;
    if (ok0 | ok1) break;
  }
  return v;
}
template <class T>
RT_DEV V3 random_unit_vector(const Rng& g, uint32_t bounce, T& tally) {  // base.clj:196-197
  return unit(random_in_unit_sphere(g, bounce, tally));
}

// The same draw for a whole wave at once: every lane with `need` gets what
// random_in_unit_sphere(g, bounce) would return; ALL 64 lanes must call it, converged.
// The plain loop runs until the unluckiest lane accepts (about four Philox blocks per wave and
// bounce, against one needed per lane on average) with most lanes idle. Here every lane first
// tries its own block 0; the lanes still without a vector (27 %) then publish their key in LDS and
// the WHOLE wave works for them: lane l evaluates block next + (l mod h) of the (l / h)-th
// unlucky lane, h = 16 down to 1 helpers each (as many as 64 lanes allow), so one more block time covers 2h further
// candidates per unlucky lane. A ballot tells each unlucky lane which of its helpers accepted first
// (helpers are in block order, the two candidates of a block in order: the reference's first
// accepted draw) and it pulls that vector across lanes.
#define RT_LDS __attribute__((address_space(3)))
RT_DEV uint32_t wave_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
RT_DEV float lane_pull(float v, uint32_t src_lane) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __float_as_int(v)));
}
template <bool ROWS = false, class T>
RT_DEV V3 wave_random_in_unit_sphere(bool need, const Rng& g, uint32_t bounce, uint32_t RT_LDS* xch0, uint32_t xs, T& tally) {
  // entry e of the exchange area: four words at (e / 16) * xs + (e % 16) * 4. xs = 64 is an area of its own, 1 KiB in
  // one piece; kernels that walk trees lay the area over the first four levels of the wave's own traversal stacks
  // (idle while the wave draws), whose levels lie xs = blockDim words apart.
  // (ROWS = false: the area in one piece, xs not looked at — the form the kernels without trees compile, whose
  // issue slots are all taken: the general form costs the Cornell box 0.9 %)
  auto entry = [&](uint32_t e) { return ROWS ? xch0 + (e >> 4) * xs + (e & 15u) * 4u : xch0 + e * 4u; };
  const uint32_t lane = wave_lane();
  uint32_t o[4];
  philox(g, 0, bounce, P_SPHERE, o);
  V3 v0 = cube_point(o[0], o[1]), v1 = cube_point(o[2], o[3]);
  bool ok0 = len2(v0) >= 1.0f, ok1 = len2(v1) >= 1.0f;
  V3 v = ok0 ? v0 : v1;
  bool found = !need || ok0 || ok1;
  if (need) tally.template add<RT_EV_SPHERE_TRIAL>(ok0 ? 1 : 2);
  uint32_t next_blk = 1;
  unsigned long long unlucky = __ballot(!found);
  while (unlucky != 0ull) {  // wave-uniform
    const uint32_t n = (uint32_t)__popcll(unlucky);
    // helpers each: as many as 64 lanes allow, not only powers of two (17 to 21 unlucky lanes get three helpers
    // each, six candidates, where two would leave one lane in thirteen for a further round); hm = ceil(2^16 / h),
    // so that lane / h = (lane * hm) >> 16 for every lane below 64
    uint32_t h, hm;
    if (n <= 4) { h = 16u; hm = 4096u; }
    else if (n <= 8) { h = 8u; hm = 8192u; }
    else if (n <= 9) { h = 7u; hm = 9363u; }
    else if (n <= 10) { h = 6u; hm = 10923u; }
    else if (n <= 12) { h = 5u; hm = 13108u; }
    else if (n <= 16) { h = 4u; hm = 16384u; }
    else if (n <= 21) { h = 3u; hm = 21846u; }
    else if (n <= 32) { h = 2u; hm = 32768u; }
    else { h = 1u; hm = 65536u; }
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(unlucky >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)unlucky, 0u));
    if (!found) {
      uint32_t RT_LDS* xch = entry(rank);
      xch[0] = g.k0; xch[1] = g.k1; xch[2] = bounce; xch[3] = next_blk;
    }
    const uint32_t slot = (lane * hm) >> 16;
    const bool helper = slot < n;
    const uint32_t rs = helper ? slot : 0u;
    const uint32_t RT_LDS* xr = entry(rs);
    Rng hg{xr[0], xr[1], g.s0, g.s1};
    const uint32_t hb = xr[2];
    const uint32_t blk = xr[3] + (lane - slot * h);
    philox(hg, blk, hb, P_SPHERE, o);
    v0 = cube_point(o[0], o[1]); v1 = cube_point(o[2], o[3]);
    ok0 = len2(v0) >= 1.0f; ok1 = len2(v1) >= 1.0f;
    const V3 hv = ok0 ? v0 : v1;
    const unsigned long long accepted = __ballot(helper && (ok0 || ok1) && blk < RT_MAX_TRIALS / 2);
    // this lane's helpers are lanes [rank * h, rank * h + h): the lowest set bit is the first accepted block
    const uint32_t bits = (uint32_t)(accepted >> ((rank * h) & 63u)) & ((1u << h) - 1u);
    const bool got = !found && bits != 0u;
    const uint32_t first = (uint32_t)__builtin_ctz(bits | 0x10000u);
    const uint32_t src = (rank * h + first) & 63u;
    const float px = lane_pull(hv.x, src), py = lane_pull(hv.y, src), pz = lane_pull(hv.z, src);
    if constexpr (T::on) {
      const int first_cand = __builtin_amdgcn_ds_bpermute((int)(src << 2), ok0 ? 1 : 2);
      if (!found) tally.template add<RT_EV_SPHERE_TRIAL>(got ? first * 2u + (uint32_t)first_cand : h * 2u);
    }
    if (got) { v = mk(px, py, pz); found = true; }
    if (!found) {
      next_blk += h;
      if (next_blk >= RT_MAX_TRIALS / 2) found = true;  // give up like the plain loop (never in practice)
    }
    unlucky = __ballot(!found);
  }
  return v;
}

// ------------------------------------------------------------------------ scene context

template <uint32_t FEAT>
struct Ctx {
  static constexpr uint32_t feat = FEAT;
  static constexpr bool has(uint32_t f) { return (FEAT & f) != 0; }
  static constexpr bool rects() { return (FEAT & F_RECTS) != 0; }
  DScene S;
  const DPrimInfo RT_LDS* lds_info;    // F_SMALL: the scene's DPrimInfo table in LDS
  const DChainRec RT_LDS* lds_chains;  // F_SMALL: its chain records in LDS
  const float* lds_perlin_vec;      // n_perlin * 1024 floats (LDS)
  const uint8_t* lds_perlin_perm;   // n_perlin * 768 bytes (LDS)
  uint32_t RT_LDS* lds_xch;         // this wave's 1 KiB exchange area (wave_random_in_unit_sphere)
  uint32_t xch_stride;              // ... and how its 16-entry rows are spaced (see there)
  int* lds_stack;                   // BVH stack base of this lane (stride = blockDim.x)
  int stack_stride;
  typedef uint32_t node_q __attribute__((ext_vector_type(4)));
  const node_q RT_LDS* lds_node_half;  // second halves of the packed tree nodes in LDS (null: read them from memory)
#ifdef RT_STAMPS
  unsigned long long* st_leaf;      // diagnostic build: wave cycles spent in BVH leaf tests,
  unsigned long long* st_walk;      //   in the walk between them, and the number of leaf rounds
  unsigned long long* st_rounds;
  unsigned long long* st_grp;       //   wave cycles per surface group (8 slots) and, in slot 7, in the media
#endif
};

// --------------------------------------------------------------------------- primitives
struct Best {
  float t;   // closest so far (the reference's closest-so-far, base.clj:94-97)
  int prim;  // device primitive index; -1 = none; <= -2 = medium (-2 - m)
  double root;  // a sphere's winning root in FP64, for its record (base.clj:243-246 take the FP64 root)
};

RT_DEV int prim_type(int type_group) { return type_group & 15; }
RT_DEV int prim_group(int type_group) { return type_group >> 4; }
RT_DEV bool is_rect(int type) { return type >= RT_PRIM_XY_RECT; }

// Per (ray, group) constants.
struct RayX {
  V3 o, d;
  V3 inv;        // 1/d
  V3 o_lo, d_lo; // FP64 residuals of o and d (zero unless the group asks for an FP64 ray)
  float tm;
};

RT_DEV RayX make_rayx(V3 o, V3 d, float tm) {
  RayX r;
  r.o = o; r.d = d; r.tm = tm;
  r.inv = mk(rcp(d.x), rcp(d.y), rcp(d.z));
  r.o_lo = r.d_lo = mk(0.f, 0.f, 0.f);
  return r;
}
// LO = false where the residuals are known to be zero (instantiations without chains): x + 0.0 is
// not x for x = -0.0, so the compiler would keep those additions.
template <bool LO>
RT_DEV D3 rayx_o64(const RayX& r) {
  if constexpr (!LO) return tod(r.o);
  else return mkd((double)r.o.x + (double)r.o_lo.x, (double)r.o.y + (double)r.o_lo.y, (double)r.o.z + (double)r.o_lo.z);
}
template <bool LO>
RT_DEV D3 rayx_d64(const RayX& r) {
  if constexpr (!LO) return tod(r.d);
  else return mkd((double)r.d.x + (double)r.d_lo.x, (double)r.d.y + (double)r.d_lo.y, (double)r.d.z + (double)r.d_lo.z);
}
// How an instantiation intersects spheres (template argument of sphere_roots / sphere_t).
enum { SPH_ONE_RCP = 1, SPH_LO = 2, SPH_STATIC = 4 };  // SPH_STATIC: the world has no moving sphere
// SURF: the ray of a surface group. Those never hold spheres below a chain (the host carries such spheres into world
// space), so their rays have no residuals: six registers fewer across a walk and six FP64 additions fewer per sphere.
template <class X, bool SURF = false>
constexpr int sphere_mode() { return SPH_ONE_RCP | ((X::has(F_CHAINS) && !SURF) ? SPH_LO : 0) | (X::has(F_MOVING) ? 0 : SPH_STATIC); }

// rectangles.clj:20-30 (XY), :57-67 (XZ), :94-104 (YZ): t from the constant axis, reject iff
// t < t-min or t > t-max (so equality with the closest-so-far is accepted and a later rect
// wins an exact tie), then the inclusive in-plane bounds. Written without branches: the six
// comparisons keep the reference's polarity (a NaN passes them, as it does there).
RT_DEV bool rect_ok_axes(float a0, float a1, float b0, float b1, float k, float oc, float ic, float oa, float da,
                         float ob, float db, float tmin, float tmax, float& t) {
  t = (k - oc) * ic;
  float a = fmaf(t, da, oa), b = fmaf(t, db, ob);
  // One signed "outside distance" instead of six compares chained through scalar masks: each of
  // the six terms is > 0 exactly when the reference's rejecting comparison holds; fmaxf drops a
  // NaN term, as the reference's comparisons let a NaN pass.
  float m = fmaxf(fmaxf(fmaxf(a0 - a, a - a1), fmaxf(b0 - b, b - b1)), fmaxf(tmin - t, t - tmax));
  return !(m > 0.0f);
}

template <bool STATIC = false, class PRIM>
RT_DEV D3 sphere_center(const PRIM& P, int type, float tm) {
  if (STATIC || type == RT_PRIM_SPHERE) return mkd(P.d[0], P.d[1], P.d[2]);
  // base.clj:298-302
  double k = ((double)tm - P.d[6]) / (P.d[7] - P.d[6]);
  return mkd(P.d[0] + (P.d[3] - P.d[0]) * k, P.d[1] + (P.d[4] - P.d[1]) * k,
             P.d[2] + (P.d[5] - P.d[2]) * k);
}
template <bool STATIC = false, class PRIM>
RT_DEV double sphere_radius(const PRIM& P, int type) { return (STATIC || type == RT_PRIM_SPHERE) ? P.d[3] : P.d[8]; }

// n / d in FP64, correctly rounded, given r = RN(1/d) (Markstein): q = RN(n r), then one FMA
// residual and one FMA correction. Three instructions instead of the ~12 of a full division;
// d = 0 takes the real division and its IEEE result.
RT_DEV double exact_div(double n, double d, double r) {
  if (d == 0.0) return n / d;
  double q = n * r;
  return fma(fma(-q, d, n), r, q);
}

// base.clj:234-241,247 / :274-281,287 in FP64: both candidate roots. The second candidate is
// (-half_b + root1) / a, as the reference computes it (it adds the first root, not
// sqrt(discriminant)). False when the discriminant is not positive.
// SPH_ONE_RCP: both roots from one reciprocal of a and two corrected products (same values, fewer
// instructions, two more live registers).
// SPH_LO: the ray carries FP64 residuals.
template <int MODE, class PRIM, class T>
RT_DEV bool sphere_roots(const PRIM& P, int type, const RayX& r, double& root1, double& root2, T& tally) {
  if (type == RT_PRIM_MOVING_SPHERE) tally.template add<RT_EV_MSPHERE_CENTER>();
  D3 c = sphere_center<(MODE & SPH_STATIC) != 0>(P, type, r.tm);
  double radius = sphere_radius<(MODE & SPH_STATIC) != 0>(P, type);
  D3 o = rayx_o64<(MODE & SPH_LO) != 0>(r), d = rayx_d64<(MODE & SPH_LO) != 0>(r);
  D3 oc = mkd(o.x - c.x, o.y - c.y, o.z - c.z);
  double a = ddot(d, d);
  double half_b = ddot(oc, d);
  double cc = fma(-radius, radius, ddot(oc, oc));
  double disc = fma(half_b, half_b, -(a * cc));
  tally.template add<RT_EV_SPHERE_TEST>();
  if (!(disc > 0.0)) return false;
  double sqrtd = sqrt(disc);
  if constexpr ((MODE & SPH_ONE_RCP) != 0) {
    const double ra = 1.0 / a;
    root1 = exact_div(-half_b - sqrtd, a, ra);
    root2 = exact_div(-half_b + root1, a, ra);
  } else {
    root1 = (-half_b - sqrtd) / a;
    root2 = (-half_b + root1) / a;
  }
  return true;
}

// The range tests of base.clj:242,248: t_min < root (FP64) and (float)root < t_max (strict, so
// the earlier of two coincident spheres keeps an exact tie).
template <int MODE, class PRIM, class T>
RT_DEV bool sphere_t(const PRIM& P, int type, const RayX& r, double tmin, float tmax, float& t_out, double& root_out,
                     T& tally) {
  double root1, root2;
  if (!sphere_roots<MODE>(P, type, r, root1, root2, tally)) return false;
  tally.template add<RT_EV_SPHERE_ROOT>();
  if (root1 > tmin && (float)root1 < tmax) { t_out = (float)root1; root_out = root1; return true; }
  tally.template add<RT_EV_SPHERE_ROOT>();
  if (root2 > tmin && (float)root2 < tmax) { t_out = (float)root2; root_out = root2; return true; }
  return false;
}

// Exact-tie rule of the reference's linear scan (base.clj:92-101) for a traversal that visits
// primitives out of order: indices are scan-order ranks. A rect accepts t == closest, a sphere
// does not, so on a tie a rect beats any earlier hit and a sphere never displaces one.
RT_DEV bool wins_tie(int cand_type, int cand_idx, int best_type, int best_idx) {
  if (is_rect(cand_type)) return !is_rect(best_type) || cand_idx > best_idx;
  return !is_rect(best_type) && cand_idx < best_idx;
}

typedef float f2 __attribute__((ext_vector_type(2)));
struct BoxSlab;
RT_DEV BoxSlab box_slab6(float x0, float x1, float y0, float y1, float z0, float z1, const RayX& r);
RT_DEV bool open_box_accept(const BoxSlab& s, float pm, float tmin, float tmax, float& t);
template <class X, class T>
RT_DEV void test_lidbox(const DScene& S, int idx, const RayX& r, float tmin, Best& best, T& tally);

// One primitive visited OUT of scan order (BVH leaves): explicit tie rule on the ranks.
// `leaf` is what the tree stores: index | type << RT_LEAF_TYPE_SHIFT.
template <class X, bool SURF = false, class T>
RT_DEV void test_prim_any_order(const DScene& S, int leaf, const RayX& r, float tmin, Best& best, T& tally, double tmin64) {
  const DPrim RT_CONST* prims = cptr(S.prims);
  const int idx = leaf & RT_LEAF_INDEX_MASK, type = leaf >> RT_LEAF_TYPE_SHIFT;
  const DPrim RT_CONST& P = prims[idx];
  float t;
  double root = 0.0;
  if constexpr (X::has(F_LIDBOX)) {
    if (type == RT_PRIM_LIDBOX) {
      test_lidbox<X>(S, idx, r, tmin, best, tally);
      return;
    }
  }
  if (X::rects() && is_rect(type)) {
    tally.template add<RT_EV_RECT_TEST>();
    // The orientation differs per lane here: select the ray's components. The copies are made
    // opaque first; selecting between loads of one struct would otherwise be rewritten into a
    // load from a selected address, which parks the whole ray in scratch memory.
    float ox = r.o.x, oy = r.o.y, oz = r.o.z, dx = r.d.x, dy = r.d.y, dz = r.d.z, ix = r.inv.x, iy = r.inv.y, iz = r.inv.z;
    asm volatile("" : "+v"(ox), "+v"(oy), "+v"(oz), "+v"(dx), "+v"(dy), "+v"(dz), "+v"(ix), "+v"(iy), "+v"(iz));
    const bool xy = type == RT_PRIM_XY_RECT, xz = type == RT_PRIM_XZ_RECT;
    const float oc = xy ? oz : (xz ? oy : ox), ic = xy ? iz : (xz ? iy : ix);
    const float oa = (xy | xz) ? ox : oy, da = (xy | xz) ? dx : dy;
    const float ob = xy ? oy : oz, db = xy ? dy : dz;
    const bool ok = rect_ok_axes(P.f[0], P.f[1], P.f[2], P.f[3], P.f[4], oc, ic, oa, da, ob, db, tmin, best.t, t);
    if (!ok) return;
    if (t == best.t && best.prim >= 0) {
      const int bi = best.prim;
      if (!wins_tie(type, idx, prim_type(prims[bi].type), bi)) return;
    }
  } else {
    if constexpr (X::has(F_SPHERES)) {
      if (!sphere_t<sphere_mode<X, SURF>()>(P, type, r, tmin64, __builtin_inff(), t, root, tally)) return;
      if (t > best.t) return;
      if (t == best.t) {
        if (best.prim < 0) return;  // t == t_max with nothing hit yet: the strict test fails
        const int bi = best.prim;
        if (!wins_tie(type, idx, prim_type(prims[bi].type), bi)) return;
      }
    } else {
      return;
    }
  }
  best.t = t;
  best.prim = idx;
  best.root = root;
}

// ------------------------------------------------------------------------------- chains
// The ray of a group: its chain's collapsed rigid motion applied to the world ray
// (translate.clj:11, rotate.clj:18-27). Groups with spheres below a chain get the motion in FP64
// on the exact FP32 world ray, split into an FP32 value and an FP32 residual, so the FP64 sphere
// test sees what the reference sees. G is wave-uniform in the traversal loops (SGPR operands).
template <class X, bool SURF = false, class T>
RT_DEV RayX group_ray(const DScene& S, int g, int gflags, float rc, float rs, float bx, float by, float bz,
                      const Ray& wr, T& tally) {
  if constexpr (!X::has(F_CHAINS)) {
    return make_rayx(wr.o, wr.d, wr.tm);
  } else {
    if (!SURF && X::has(F_SPHERES) && (gflags & DGROUP_SPHERES_UNDER_CHAIN)) {
      const DGroup64 RT_CONST& H = cptr(S.groups64)[g];
      const double c = H.rc, s = H.rs;
      D3 wo = tod(wr.o), wd = tod(wr.d);
      D3 o = mkd(fma(c, wo.x, fma(-s, wo.z, H.bx)), wo.y + H.by, fma(s, wo.x, fma(c, wo.z, H.bz)));
      D3 d = mkd(fma(c, wd.x, -(s * wd.z)), wd.y, fma(s, wd.x, c * wd.z));
      tally.template add<RT_EV_ROTATE_RAY>();
      RayX r = make_rayx(tof(o), tof(d), wr.tm);
      r.o_lo = mk((float)(o.x - (double)r.o.x), (float)(o.y - (double)r.o.y), (float)(o.z - (double)r.o.z));
      r.d_lo = mk((float)(d.x - (double)r.d.x), (float)(d.y - (double)r.d.y), (float)(d.z - (double)r.d.z));
      return r;
    }
    const float c = rc, s = rs;
    // explicit FMAs: every inlined copy must round identically
    V3 o = mk(fmaf(c, wr.o.x, fmaf(-s, wr.o.z, bx)), wr.o.y + by, fmaf(s, wr.o.x, fmaf(c, wr.o.z, bz)));
    V3 d = mk(fmaf(c, wr.d.x, -(s * wr.d.z)), wr.d.y, fmaf(s, wr.d.x, c * wr.d.z));
    tally.template add<RT_EV_ROTATE_RAY>();
    return make_rayx(o, d, wr.tm);
  }
}

// ---------------------------------------------------------------------------------- BVH
// One child box of a packed node against the ray: lo_hi = (min | max << 16 ... as stored), the
// six bounds arrive as FP16 halves and go into the FMAs as they are (v_fma_mix_f32).
// t = bound * (1/d) - o/d: less exact than (bound - o) * (1/d) when the origin is far from the box
// relative to its size, by a few ulps of o/d; the boxes' outward rounding covers that many times.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
// The box arrives as centre and half-extent (FP16): the ray's parameter at the centre plane of an axis is
// tc = c * (1/d) - o/d and the slab is tc -+ h * |1/d|, so the near and the far parameter need no min/max to
// tell them apart: nine FMAs per box instead of six FMAs and six min/max.
RT_DEV bool slab16(_Float16 cx, _Float16 cy, _Float16 cz, _Float16 hx, _Float16 hy, _Float16 hz, V3 inv, V3 oi,
                   float tmin, float tmax, float& tnear) {
  const float tcx = fmaf((float)cx, inv.x, -oi.x), tcy = fmaf((float)cy, inv.y, -oi.y), tcz = fmaf((float)cz, inv.z, -oi.z);
  const float ax = fabsf(inv.x), ay = fabsf(inv.y), az = fabsf(inv.z);
  // (the half-extent is the operand that is negated: with -|1/d| the compiler keeps three more registers)
  const float n_x = fmaf(-(float)hx, ax, tcx), f_x = fmaf((float)hx, ax, tcx);
  const float n_y = fmaf(-(float)hy, ay, tcy), f_y = fmaf((float)hy, ay, tcy);
  const float n_z = fmaf(-(float)hz, az, tcz), f_z = fmaf((float)hz, az, tcz);
  const float tn = fmaxf(fmaxf(n_x, n_y), fmaxf(n_z, tmin));
  const float tf = fminf(fminf(f_x, f_y), fminf(f_z, tmax));
  tnear = tn;
  return tn <= tf;
}

// Closest hit below BVH node `root`. Two nested loops: a lane walks internal nodes (near child
// first, far child on its LDS stack) until it HOLDS a leaf, and only when every lane of the wave
// holds a leaf or has finished are the leaves intersected, together. Testing a leaf where it is
// met would make every step of the walk pay for a primitive test that a few lanes need.
// A node reference is >= 0 for an internal node, ~primitive for a leaf, RT_BVH_DONE when the lane
// has nothing left. Leaves are visited out of scan order: test_prim_any_order applies the
// reference's tie rule explicitly.
#define RT_BVH_DONE 0x7fffffff
template <bool LDSQ, bool SURF, class X, class T>
RT_DEV void bvh_traverse(const X& ctx, int root, const RayX& r, float tmin, Best& best, T& tally, double tmin64) {
  const DScene& S = ctx.S;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const u4 RT_CONST* nodes4 = reinterpret_cast<const u4 RT_CONST*>(cptr(S.nodes));
  // 1/d clamped to +-1e30: with an infinite reciprocal (a ray along an axis) bound * inf - o * inf
  // is inf - inf; with the clamp the two parameters of that axis are huge and of opposite sign when
  // the origin lies between the bounds, of the same sign when it does not, as they should be.
  const V3 inv = mk(__builtin_amdgcn_fmed3f(r.inv.x, -1e30f, 1e30f), __builtin_amdgcn_fmed3f(r.inv.y, -1e30f, 1e30f),
                    __builtin_amdgcn_fmed3f(r.inv.z, -1e30f, 1e30f));
  const V3 oi = mk(r.o.x * inv.x, r.o.y * inv.y, r.o.z * inv.z);
  // The node table's address stays in registers across the walk (the register allocator would
  // otherwise re-read it from the kernel arguments, a scalar-memory round trip, in every step).
  asm volatile("" : "+s"(nodes4));
  // The stack index is kept as an element offset (entry i of this lane is lds_stack[i * stride]).
  const int stride = ctx.stack_stride;
  int sp = 0;
  int node = root;
#ifdef RT_STAMPS
  unsigned long long st_a, st_b, st_c;
#endif
  // one step of the walk from the two quarters of a node record
  auto step = [&](const u4 q0, const u4 q1) {
    // halves in storage order: left centre xyz, left half-extent xyz, right centre xyz, right half-extent xyz
    // (plain scalars first: a bit cast applied to a vector element reads the vector's first one)
    const uint32_t w0 = q0.x, w1 = q0.y, w2 = q0.z, w3 = q0.w, w4 = q1.x, w5 = q1.y;
    const h2 a0 = __builtin_bit_cast(h2, w0), a1 = __builtin_bit_cast(h2, w1), a2 = __builtin_bit_cast(h2, w2);
    const h2 a3 = __builtin_bit_cast(h2, w3), a4 = __builtin_bit_cast(h2, w4), a5 = __builtin_bit_cast(h2, w5);
    const int left = (int)q1.z, right = (int)q1.w;
    float tl, tr;
    tally.template add<RT_EV_AABB_TEST>(2);
    const bool hl = slab16(a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, inv, oi, tmin, best.t, tl);
    const bool hr = slab16(a3.x, a3.y, a4.x, a4.y, a5.x, a5.y, inv, oi, tmin, best.t, tr);
    if (hl && hr) {
      const bool swap = tr < tl;
      ctx.lds_stack[sp] = swap ? left : right;
      sp += stride;
      node = swap ? right : left;
    } else if (hl || hr) {
      node = hl ? left : right;
    } else if (sp > 0) {
      sp -= stride;
      node = ctx.lds_stack[sp];
    } else {
      node = RT_BVH_DONE;
    }
  };
  // The root is the same record for every lane: its step reads it through the scalar cache (the index is
  // wave-uniform), one vector-memory round trip less per walk.
  step(nodes4[root * 2 + 0], nodes4[root * 2 + 1]);
  while (node != RT_BVH_DONE) {
#ifdef RT_STAMPS
    __builtin_amdgcn_sched_barrier(0); st_a = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
    // A step's 32 bytes per lane arrive over two pipes when the trees fit: the first half of the record
    // through the vector memory pipe, the second half from its copy in LDS. The walk is held by the
    // rate at which the memory pipe returns 16 bytes per lane (its data path is busy nine cycles in ten
    // with both halves on it), not by instruction issue.
    if constexpr (LDSQ) {
      while ((unsigned)node < (unsigned)RT_BVH_DONE) step(nodes4[node * 2 + 0], ctx.lds_node_half[node]);  // internal
    } else {
#ifdef RT_EXTRA_LOAD
      // diagnostic build (tools/ab_build.sh): one more 16-byte fetch per lane and step, its value unused and off the
      // walk's dependency chain: what it costs is the share of the vector memory pipe in the walk's time
      // (needs an even number of nodes: it reads the sibling record)
      while ((unsigned)node < (unsigned)RT_BVH_DONE) {
        const u4 extra = reinterpret_cast<const u4*>(S.nodes)[(node * 2 + 1) ^ 2];
        asm volatile("" :: "v"(extra.x), "v"(extra.y), "v"(extra.z), "v"(extra.w));
        step(nodes4[node * 2 + 0], nodes4[node * 2 + 1]);
      }
#else
      while ((unsigned)node < (unsigned)RT_BVH_DONE) step(nodes4[node * 2 + 0], nodes4[node * 2 + 1]);  // internal
#endif
    }
#ifdef RT_STAMPS
    __builtin_amdgcn_sched_barrier(0); st_b = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
    if (node != RT_BVH_DONE) {  // a leaf
      test_prim_any_order<X, SURF>(S, ~node, r, tmin, best, tally, tmin64);
      if (sp > 0) {
        sp -= stride;
        node = ctx.lds_stack[sp];
      } else {
        node = RT_BVH_DONE;
      }
    }
#ifdef RT_STAMPS
    __builtin_amdgcn_sched_barrier(0); st_c = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
    // (per-lane sums of wave-level times: the reader takes the maximum over the wave's lanes = the lane that stayed longest)
    if (ctx.st_leaf) { *ctx.st_walk += st_b - st_a; *ctx.st_leaf += st_c - st_b; *ctx.st_rounds += 1; }
#endif
  }
}

// Two rects of one orientation against the ray, in packed FP32 (one v_pk_* instruction serves
// both). rectangles.clj:20-30 / :57-67 / :94-104: t from the constant axis, then reject iff
// t < t-min, t > t-max, or the in-plane point lies outside [a0,a1] x [b0,b1] — all six bounds are
// accepted inclusively, so each of the six differences below is >= 0 (sign bit clear) exactly when
// its comparison accepts; x - x is +0, so equality accepts, and a later rect wins an exact tie
// with the closest-so-far. OR-ing the six sign bits replaces six compares. Slot 0 is tested before
// slot 1 against the closest-so-far, as the reference's scan would.
template <int TYPE>
RT_DEV void rect_pair(f2 k, f2 a0, f2 a1, f2 b0, f2 b1, int i0, int i1, const RayX& r, float tmin, Best& best) {
  float oc, ic, oa, da, ob, db;
  if (TYPE == RT_PRIM_XY_RECT) { oc = r.o.z; ic = r.inv.z; oa = r.o.x; da = r.d.x; ob = r.o.y; db = r.d.y; }
  else if (TYPE == RT_PRIM_XZ_RECT) { oc = r.o.y; ic = r.inv.y; oa = r.o.x; da = r.d.x; ob = r.o.z; db = r.d.z; }
  else { oc = r.o.x; ic = r.inv.x; oa = r.o.y; da = r.d.y; ob = r.o.z; db = r.d.z; }
  const f2 t = (k - (f2){oc, oc}) * (f2){ic, ic};
  const f2 a = __builtin_elementwise_fma(t, (f2){da, da}, (f2){oa, oa});
  const f2 b = __builtin_elementwise_fma(t, (f2){db, db}, (f2){ob, ob});
  const f2 e1 = a - a0, e2 = a1 - a, e3 = b - b0, e4 = b1 - b, e5 = t - (f2){tmin, tmin};
  {
    const float e6 = best.t - t.x;
    const int s = __float_as_int(e1.x) | __float_as_int(e2.x) | __float_as_int(e3.x) | __float_as_int(e4.x) |
                  __float_as_int(e5.x) | __float_as_int(e6);
    const bool ok = s >= 0;
    best.t = ok ? t.x : best.t;
    best.prim = ok ? i0 : best.prim;
  }
  {
    const float e6 = best.t - t.y;
    const int s = __float_as_int(e1.y) | __float_as_int(e2.y) | __float_as_int(e3.y) | __float_as_int(e4.y) |
                  __float_as_int(e5.y) | __float_as_int(e6);
    const bool ok = s >= 0;
    best.t = ok ? t.y : best.t;
    best.prim = ok ? i1 : best.prim;
  }
}

// One rect alone (a run of odd length begins with a record whose two slots hold the same rect): the arithmetic of
// one slot of rect_pair in plain FP32, the same operations on the same values, without the packed operands' set-up.
template <int TYPE>
RT_DEV void rect_single(float k, float a0, float a1, float b0, float b1, int idx, const RayX& r, float tmin, Best& best) {
  float oc, ic, oa, da, ob, db;
  if (TYPE == RT_PRIM_XY_RECT) { oc = r.o.z; ic = r.inv.z; oa = r.o.x; da = r.d.x; ob = r.o.y; db = r.d.y; }
  else if (TYPE == RT_PRIM_XZ_RECT) { oc = r.o.y; ic = r.inv.y; oa = r.o.x; da = r.d.x; ob = r.o.z; db = r.d.z; }
  else { oc = r.o.x; ic = r.inv.x; oa = r.o.y; da = r.d.y; ob = r.o.z; db = r.d.z; }
  const float t = (k - oc) * ic;
  const float a = fmaf(t, da, oa), b = fmaf(t, db, ob);
  const float e1 = a - a0, e2 = a1 - a, e3 = b - b0, e4 = b1 - b, e5 = t - tmin, e6 = best.t - t;
  const int sg = __float_as_int(e1) | __float_as_int(e2) | __float_as_int(e3) | __float_as_int(e4) | __float_as_int(e5) | __float_as_int(e6);
  const bool ok = sg >= 0;
  best.t = ok ? t : best.t;
  best.prim = ok ? idx : best.prim;
}

// Pair records fetched per scalar-memory round trip. Fewer means more trips and fewer SGPRs held
// across them; measured: while the Cornell room was loose rects, 2 (+2 % over 3, 4 is 4 % slower); with the room
// and the boxes as slab tests, 1 everywhere (0.8 % over 2 on the Cornell box, +2.5 % on random-scene).
template <class X>
constexpr int pair_batch() { return (X::has(F_SMALL) || X::has(F_NOISE)) ? RT_PAIR_BATCH : 1; }

// The rects of a brute-force group as pair records (DScene.brute_rects), XY pairs first, then XZ,
// then YZ. BATCH records are fetched per trip with unconditional scalar loads (the table
// carries that much slack) and pinned in SGPRs before any of them is used, so a box costs one
// scalar-memory round trip.
template <int BATCH, class T>
RT_DEV void brute_rects(const float RT_CONST* R, int n_xy, int n_xz, int n_yz, const RayX& r, float tmin, Best& best,
                        T& tally) {
  static_assert(BATCH >= 1 && BATCH <= RT_PAIR_SLACK, "the table carries RT_PAIR_SLACK records of slack");
  const int p_xy = (n_xy + 1) >> 1, p_xz = p_xy + ((n_xz + 1) >> 1), np = p_xz + ((n_yz + 1) >> 1);
  for (int base = 0; base < np; base += BATCH) {
    float v[BATCH * RT_PAIR_STRIDE];
#pragma unroll
    for (int q = 0; q < BATCH * RT_PAIR_STRIDE; ++q) v[q] = R[base * RT_PAIR_STRIDE + q];
    // (all of them pinned in SGPRs before the first use)
    asm volatile("" ::"s"(v[0]), "s"(v[1]), "s"(v[2]), "s"(v[3]), "s"(v[4]), "s"(v[5]), "s"(v[6]), "s"(v[7]), "s"(v[8]), "s"(v[9]), "s"(v[10]), "s"(v[11]));
    if constexpr (BATCH >= 2) asm volatile("" ::"s"(v[12]), "s"(v[13]), "s"(v[14]), "s"(v[15]), "s"(v[16]), "s"(v[17]), "s"(v[18]), "s"(v[19]), "s"(v[20]), "s"(v[21]), "s"(v[22]), "s"(v[23]));
    if constexpr (BATCH >= 3) asm volatile("" ::"s"(v[24]), "s"(v[25]), "s"(v[26]), "s"(v[27]), "s"(v[28]), "s"(v[29]), "s"(v[30]), "s"(v[31]), "s"(v[32]), "s"(v[33]), "s"(v[34]), "s"(v[35]));
    if constexpr (BATCH >= 4) asm volatile("" ::"s"(v[36]), "s"(v[37]), "s"(v[38]), "s"(v[39]), "s"(v[40]), "s"(v[41]), "s"(v[42]), "s"(v[43]), "s"(v[44]), "s"(v[45]), "s"(v[46]), "s"(v[47]));
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int q = base + j;
      if (q < np) {  // wave-uniform, as is the orientation of the pair
        const float* w = v + j * RT_PAIR_STRIDE;
        const f2 k = {w[0], w[1]}, a0 = {w[2], w[3]}, a1 = {w[4], w[5]}, b0 = {w[6], w[7]}, b1 = {w[8], w[9]};
        const int i0 = __float_as_int(w[10]), i1 = __float_as_int(w[11]);
        tally.template add<RT_EV_RECT_TEST>(i0 == i1 ? 1 : 2);
        if (i0 == i1) {  // (wave-uniform) one rect: the lights of the reference's worlds
          if (q < p_xy) rect_single<RT_PRIM_XY_RECT>(w[1], w[3], w[5], w[7], w[9], i1, r, tmin, best);
          else if (q < p_xz) rect_single<RT_PRIM_XZ_RECT>(w[1], w[3], w[5], w[7], w[9], i1, r, tmin, best);
          else rect_single<RT_PRIM_YZ_RECT>(w[1], w[3], w[5], w[7], w[9], i1, r, tmin, best);
        }
        else if (q < p_xy) rect_pair<RT_PRIM_XY_RECT>(k, a0, a1, b0, b1, i0, i1, r, tmin, best);
        else if (q < p_xz) rect_pair<RT_PRIM_XZ_RECT>(k, a0, a1, b0, b1, i0, i1, r, tmin, best);
        else rect_pair<RT_PRIM_YZ_RECT>(k, a0, a1, b0, b1, i0, i1, r, tmin, best);
      }
    }
  }
}

// The six rects of a closed box (box.clj:7-20) as one slab test. Each face is a plane hit at
// t = (k - o) / d that lies inside the face exactly when t is inside the slabs of the other two
// axes, so the faces the reference's scan accepts are the entry point max(near) and the exit point
// min(far) whenever max(near) <= min(far), and its closest-so-far rule keeps the first of them that
// is >= t-min. t is computed as the rect test computes it; the in-face test is made on parameters
// instead of points, which differs only where a ray grazes an edge within rounding. Where two
// faces tie (an edge), the later one in scan order (XY, XZ, YZ faces) wins, as `<=` makes it in
// the scan.
struct BoxSlab {
  f2 tx, ty, tz;   // plane parameters of the faces x0 x1, y0 y1, z0 z1
  float tn, tf;    // entry = max(near), exit = min(far)
};
// w = x0 x1 y0 y1 z0 z1 (wave-uniform)
RT_DEV BoxSlab box_slab6(float x0, float x1, float y0, float y1, float z0, float z1, const RayX& r);
RT_DEV BoxSlab box_slab(const float* w, const RayX& r) { return box_slab6(w[0], w[1], w[2], w[3], w[4], w[5], r); }
RT_DEV BoxSlab box_slab6(float x0, float x1, float y0, float y1, float z0, float z1, const RayX& r) {
  BoxSlab s;
  s.tx = ((f2){x0, x1} - (f2){r.o.x, r.o.x}) * (f2){r.inv.x, r.inv.x};
  s.ty = ((f2){y0, y1} - (f2){r.o.y, r.o.y}) * (f2){r.inv.y, r.inv.y};
  s.tz = ((f2){z0, z1} - (f2){r.o.z, r.o.z}) * (f2){r.inv.z, r.inv.z};
  // fminf/fmaxf drop a NaN operand (0 * inf when the origin lies on a face's plane)
  const float nx = fminf(s.tx.x, s.tx.y), fx = fmaxf(s.tx.x, s.tx.y);
  const float ny = fminf(s.ty.x, s.ty.y), fy = fmaxf(s.ty.x, s.ty.y);
  const float nz = fminf(s.tz.x, s.tz.y), fz = fmaxf(s.tz.x, s.tz.y);
  s.tn = fmaxf(fmaxf(nx, ny), nz);
  s.tf = fminf(fminf(fx, fy), fz);
  return s;
}
// The box's answer to `hit` with range [tmin, tmax]: the entry point if it is >= tmin, else the
// exit point; accepted iff tn <= tf, t >= tmin and t <= tmax — the OR of the three differences'
// sign bits, as in rect_pair (x - x is +0, so every equality accepts).
RT_DEV bool box_accept(const BoxSlab& s, float tmin, float tmax, float& t) {
  t = s.tn >= tmin ? s.tn : s.tf;
  const int sgn = __float_as_int(s.tf - s.tn) | __float_as_int(t - tmin) | __float_as_int(tmax - t);
  return sgn >= 0;
}
// The same for a box with ONE face missing (five rects that are faces of one axis-aligned box: the
// walls of the Cornell room, scenes.clj:76-81). The scan still accepts exactly the faces the ray
// pierces, the entry and the exit of the slab, minus the missing one; pm is the plane parameter
// of the missing face. (Where the missing plane ties with a neighbour's, on the rim of the opening,
// the reference's scan would take the neighbour; such a ray is lost here.)
RT_DEV bool open_box_accept(const BoxSlab& s, float pm, float tmin, float tmax, float& t) {
  const bool entry = s.tn >= tmin && s.tn != pm;
  t = entry ? s.tn : s.tf;
  const int sgn = __float_as_int(s.tf - s.tn) | __float_as_int(t - tmin) | __float_as_int(tmax - t);
  return sgn >= 0 && (entry || s.tf != pm);
}
// A RT_PRIM_LIDBOX leaf (see rt_scene.h): the floor and the four walls as the slab test of a box whose y1 face is
// missing, then the lid as the XZ rect it is. Both offers are made as the scan would make them, floor and walls
// first, against the closest hit so far; a leaf is met out of scan order, so an exact tie with what is held goes by
// the ranks of the rects (test_prim_any_order's rule), and the lid, later in the scan than the z faces and the
// floor and earlier than the x faces, ties with them accordingly.
template <class X, class T>
RT_DEV void test_lidbox(const DScene& S, int idx, const RayX& r, float tmin, Best& best, T& tally) {
  const DPrim RT_CONST* prims = cptr(S.prims);
  const DPrim RT_CONST& P = prims[idx];
  const float x0 = P.f[0], x1 = P.f[1], y0 = P.f[2], y1 = P.f[3], z0 = P.f[4], z1 = P.f[5];
  const float lid_k = P.f[6];
  const int f_z0 = __float_as_int(P.f[7]), f_z1 = __float_as_int(P.f[8]), f_y0 = __float_as_int(P.f[9]);
  const int f_x0 = __float_as_int(P.f[10]), f_x1 = __float_as_int(P.f[11]), f_lid = __float_as_int(P.f[12]);
  tally.template add<RT_EV_BOX_SLAB>();
  const BoxSlab s = box_slab6(x0, x1, y0, y1, z0, z1, r);
  float t;
  if (open_box_accept(s, s.ty.y, tmin, best.t, t)) {
    // the face whose plane gave t; where two planes tie (an edge) the later one in scan order wins (z0 z1 y0 x0 x1)
    int face = f_z0;
    face = s.tz.y == t ? f_z1 : face;
    face = s.ty.x == t ? f_y0 : face;
    face = s.tx.x == t ? f_x0 : face;
    face = s.tx.y == t ? f_x1 : face;
    bool take = true;
    if (t == best.t && best.prim >= 0) take = wins_tie(RT_PRIM_XY_RECT, face, prim_type(prims[best.prim].type), best.prim);
    if (take) { best.t = t; best.prim = face; best.root = 0.0; }
  }
  tally.template add<RT_EV_RECT_TEST>();
  // (opaque copies, as in test_prim_any_order: beside that function's per-lane selects the optimiser would
  // otherwise fetch the ray's components from a selected address and park the ray in scratch memory)
  float ox = r.o.x, oy = r.o.y, oz = r.o.z, dx = r.d.x, dz = r.d.z, iy = r.inv.y;
  asm volatile("" : "+v"(ox), "+v"(oy), "+v"(oz), "+v"(dx), "+v"(dz), "+v"(iy));
  if (rect_ok_axes(x0, x1, z0, z1, lid_k, oy, iy, ox, dx, oz, dz, tmin, best.t, t)) {
    bool take = true;
    if (t == best.t && best.prim >= 0) take = wins_tie(RT_PRIM_XZ_RECT, f_lid, prim_type(prims[best.prim].type), best.prim);
    if (take) { best.t = t; best.prim = f_lid; best.root = 0.0; }
  }
}

template <class T>
RT_DEV void brute_boxes(const float RT_CONST* B, int n_box, const RayX& r, float tmin, Best& best, T& tally) {
  for (int b = 0; b < n_box; ++b) {
    float w[RT_BOX_STRIDE];
#pragma unroll
    for (int q = 0; q < RT_BOX_STRIDE; ++q) w[q] = B[b * RT_BOX_STRIDE + q];
    // the whole record in SGPRs before any of it is used: one scalar-memory round trip per box
    asm volatile("" ::"s"(w[0]), "s"(w[1]), "s"(w[2]), "s"(w[3]), "s"(w[4]), "s"(w[5]), "s"(w[6]), "s"(w[7]), "s"(w[8]),
                 "s"(w[9]), "s"(w[10]), "s"(w[11]));
    tally.template add<RT_EV_BOX_SLAB>();
    const BoxSlab s = box_slab(w, r);
    float t;
    const bool ok = box_accept(s, tmin, best.t, t);
    // the face whose plane gave t; where two planes tie (an edge) the later one in scan order
    // (z0 z1 y0 y1 x0 x1) wins, as the scan's `<=` makes it
    int face = __float_as_int(w[6]);
    face = s.tz.y == t ? __float_as_int(w[7]) : face;
    face = s.ty.x == t ? __float_as_int(w[8]) : face;
    face = s.ty.y == t ? __float_as_int(w[9]) : face;
    face = s.tx.x == t ? __float_as_int(w[10]) : face;
    face = s.tx.y == t ? __float_as_int(w[11]) : face;
    best.t = ok ? t : best.t;
    best.prim = ok ? face : best.prim;
  }
}
// The records with one face missing (its index is -1), a loop of their own so that the closed boxes'
// loop stays as lean as it is.
template <class T>
RT_DEV void brute_open_boxes(const float RT_CONST* B, int n_open, const RayX& r, float tmin, Best& best, T& tally) {
  for (int b = 0; b < n_open; ++b) {
    float w[RT_BOX_STRIDE];
#pragma unroll
    for (int q = 0; q < RT_BOX_STRIDE; ++q) w[q] = B[b * RT_BOX_STRIDE + q];
    asm volatile("" ::"s"(w[0]), "s"(w[1]), "s"(w[2]), "s"(w[3]), "s"(w[4]), "s"(w[5]), "s"(w[6]), "s"(w[7]), "s"(w[8]),
                 "s"(w[9]), "s"(w[10]), "s"(w[11]));
    tally.template add<RT_EV_BOX_SLAB>();
    const BoxSlab s = box_slab(w, r);
    float pm = s.tz.x;  // (the conditions are wave-uniform)
    pm = __float_as_int(w[7]) < 0 ? s.tz.y : pm;
    pm = __float_as_int(w[8]) < 0 ? s.ty.x : pm;
    pm = __float_as_int(w[9]) < 0 ? s.ty.y : pm;
    pm = __float_as_int(w[10]) < 0 ? s.tx.x : pm;
    pm = __float_as_int(w[11]) < 0 ? s.tx.y : pm;
    float t;
    const bool ok = open_box_accept(s, pm, tmin, best.t, t);
    int face = __float_as_int(w[6]);
    face = s.tz.y == t ? __float_as_int(w[7]) : face;
    face = s.ty.x == t ? __float_as_int(w[8]) : face;
    face = s.ty.y == t ? __float_as_int(w[9]) : face;
    face = s.tx.x == t ? __float_as_int(w[10]) : face;
    face = s.tx.y == t ? __float_as_int(w[11]) : face;
    best.t = ok ? t : best.t;
    best.prim = ok ? face : best.prim;
  }
}

// The brute-force loops of group G (rect pairs, closed boxes, spheres in scan order) for the
// group-space ray r.
template <class X, bool SURF = false, class T>
RT_DEV void brute_group_hit(const DScene& S, const DGroup RT_CONST& G, const RayX& r, float tmin, Best& best, T& tally,
                            double tmin64) {
  const DPrim RT_CONST* prims = cptr(S.prims);
  int first = G.first_prim;
  const int loose = G.loose;
  if constexpr (X::rects()) {
  brute_rects<pair_batch<X>()>(cptr(S.brute_rects) + G.first_rect * RT_PAIR_STRIDE, loose & 255, (loose >> 8) & 255, (loose >> 16) & 255,
              r, tmin, best, tally);
  brute_boxes(cptr(S.brute_boxes) + G.first_box * RT_BOX_STRIDE, loose >> 24, r, tmin, best, tally);
  brute_open_boxes(cptr(S.brute_boxes) + (G.first_box + (loose >> 24)) * RT_BOX_STRIDE, (G.flags >> DGROUP_OPEN_SHIFT) & 255, r, tmin, best, tally);
  }
  first += G.n_xy + G.n_xz + G.n_yz;
  if constexpr (X::has(F_SPHERES)) {
    const int n_sph = G.n_sph;
    for (int i = 0; i < n_sph; ++i) {
      const DPrim RT_CONST& P = prims[first + i];
      float t = 0.f;
      double root = 0.0;
      bool ok = sphere_t<sphere_mode<X, SURF>()>(P, prim_type(P.type), r, tmin64, best.t, t, root, tally);
      best.t = ok ? t : best.t;
      best.prim = ok ? first + i : best.prim;
      best.root = ok ? root : best.root;
    }
  }
}

// Closest hit of group g for the world ray wr with t in (tmin, best.t].
// tmin64: the lower bound the spheres compare their FP64 roots with (the FP32 bound widened, unless
// the caller holds a more exact one).
template <bool SURF = false, class X, class T>
RT_DEV void group_hit(const X& ctx, int g, const Ray& wr, float tmin, Best& best, T& tally, double tmin64) {
  const DScene& S = ctx.S;
  const DGroup RT_CONST& G = cptr(S.groups)[g];
  const RayX r = group_ray<X, SURF>(S, g, G.flags, G.rc, G.rs, G.bx, G.by, G.bz, wr, tally);
  const int bvh_root = G.bvh_root;
  if (X::has(F_BVH) && bvh_root >= 0) {
    if constexpr (X::has(F_BVH)) {
      if (ctx.lds_node_half) bvh_traverse<true, SURF>(ctx, bvh_root, r, tmin, best, tally, tmin64);  // scene-uniform
      else bvh_traverse<false, SURF>(ctx, bvh_root, r, tmin, best, tally, tmin64);
    }
  } else {
    brute_group_hit<X, SURF>(S, G, r, tmin, best, tally, tmin64);
  }
}

// ------------------------------------------------------------------------------ records
struct Rec {  // base.clj:169, plus the fields of its material (mat-ptr dereferenced once)
  V3 p, n;
  float t, u, v;
  int mat;
  bool front;
  int mtype, mflags, mtex;
  V3 rgb;
  float mx;
};

template <class PTR>
RT_DEV DPrimInfo fetch_info(PTR p) {
  DPrimInfo I;
  I.type_group = p->type_group; I.mat = p->mat; I.chain = p->chain; I.gflags = p->gflags;
  I.a0 = p->a0; I.a1 = p->a1; I.b0 = p->b0; I.b1 = p->b1;
  I.k = p->k; I.rc = p->rc; I.rs = p->rs; I.bx = p->bx; I.by = p->by; I.bz = p->bz;
  I.mtype = p->mtype; I.mflags = p->mflags; I.mtex = p->mtex;
  I.r = p->r; I.g = p->g; I.b = p->b; I.x = p->x;
  I.ox = p->ox; I.oy = p->oy; I.oz = p->oz;
  return I;
}
// Everything about primitive `prim` that its record and scatter need: one round trip, to LDS
// for small scenes.
template <class X>
RT_DEV DPrimInfo load_info(const X& ctx, int prim) {
  if constexpr (X::has(F_SMALL)) return fetch_info(ctx.lds_info + prim);
  else return fetch_info(cptr(ctx.S.prim_info) + prim);
}

// base.clj:222-227. Only the two subtractions that cancel near the poles are FP64:
// theta = acos(-y) = atan2(sqrt((1 - c)(1 + c)), c) with c = -y, the product formed in FP64 (exact to
// 1e-16 relative wherever c lies), the square root and the arctangent in FP32, where both are
// well-conditioned; |y| > 1 (the non-unit "normal" of a second-root hit) makes the product negative
// and theta NaN, as acos does. phi = atan2(-z, x) + pi does not depend on the vector's length.
// Absolute error of u and v below 3e-7 against the FP64 formulas.
RT_DEV void sphere_uv(D3 n, float& u, float& v) {
  const float PI_F = 3.14159274f;
  const double c = -n.y;
  const float s2 = (float)((1.0 - c) * (1.0 + c));
  const float theta = atan2f(sqrtf(s2), (float)c);
  const float phi = atan2f((float)(-n.z), (float)n.x) + PI_F;
  u = phi * 0.159154937f;  // 1 / (2 pi)
  v = theta * 0.318309873f;  // 1 / pi
}

// The HitRecord of primitive `prim` (info I) for the (chain-space) ray r at parameter t.
template <class X, class T>
RT_DEV void prim_record(const DScene& S, int prim, const DPrimInfo& I, const RayX& r, float t, double root,
                        bool want_uv, Rec& rec, T& tally) {
  const int type = prim_type(I.type_group);
  rec.mat = I.mat;
  rec.mtype = I.mtype; rec.mflags = I.mflags; rec.mtex = I.mtex;
  rec.rgb = mk(I.r, I.g, I.b); rec.mx = I.x;
  rec.t = t;
  rec.u = rec.v = 0.f;
  if (!X::has(F_SPHERES) || (X::rects() && is_rect(type))) {
    // rectangles.clj:31-37; the in-plane point is o + t d, the constant coordinate is set to k
    // (in FP64 the reference's o + t d lands on k to 1e-13; FP32 rounding would not).
    tally.template add<RT_EV_RECT_RECORD>();
    const float k = I.k;
    V3 p = mk(fmaf(t, r.d.x, r.o.x), fmaf(t, r.d.y, r.o.y), fmaf(t, r.d.z, r.o.z));
    // Without a per-lane switch on the orientation: `out` is the unit vector of the constant axis,
    // so both dot products below pick one component exactly, k - p.c is exact for a hit point (it
    // lies within rounding of the plane) and p.c + (k - p.c) is k.
    const V3 out = mk(I.ox, I.oy, I.oz);
    const float dn = fmaf(r.d.x, out.x, fmaf(r.d.y, out.y, r.d.z * out.z));
    const float off = k - fmaf(p.x, out.x, fmaf(p.y, out.y, p.z * out.z));
    p = mk(fmaf(out.x, off, p.x), fmaf(out.y, off, p.y), fmaf(out.z, off, p.z));
    if (want_uv) {
      const float a = (type == RT_PRIM_YZ_RECT) ? p.y : p.x, b = (type == RT_PRIM_XY_RECT) ? p.y : p.z;
      rec.u = (a - I.a0) / (I.a1 - I.a0);
      rec.v = (b - I.b0) / (I.b1 - I.b0);
    }
    rec.front = dn < 0.0f;  // set-face-normal, base.clj:171-174
    rec.n = rec.front ? out : neg(out);
    rec.p = p;
  } else {
    if constexpr (X::has(F_SPHERES)) {
      // base.clj:243-246 in FP64 from the FP64 root the winning test kept
      tally.template add<RT_EV_SPHERE_RECORD>();
      const DPrim RT_CONST& P = cptr(S.prims_rec)[prim];
      D3 c = sphere_center<!X::has(F_MOVING)>(P, type, r.tm);
      // (a static sphere carries 1 / r, the same correctly rounded quotient the division gives)
      double inv_r = P.d[4];
      if constexpr (X::has(F_MOVING)) {
        if (type != RT_PRIM_SPHERE) inv_r = 1.0 / P.d[8];  // (rare: a branch, so that static spheres do not pay the division)
      }
      D3 o64 = rayx_o64<X::has(F_CHAINS)>(r), d64 = rayx_d64<X::has(F_CHAINS)>(r);
      D3 pd = mkd(fma(root, d64.x, o64.x), fma(root, d64.y, o64.y), fma(root, d64.z, o64.z));
      D3 out = mkd((pd.x - c.x) * inv_r, (pd.y - c.y) * inv_r, (pd.z - c.z) * inv_r);
      if (want_uv) sphere_uv(out, rec.u, rec.v);
      rec.front = ddot(d64, out) < 0.0;
      rec.n = rec.front ? tof(out) : neg(tof(out));
      rec.p = tof(pd);
    }
  }
}

// Record fix-ups on the way back up a chain, rotate.clj:31-47 and translate.clj:13-17 for every
// wrapper at once (see DChainRec): the composed point map and the face tests of the two outermost
// wrappers. CHAIN points at the chain's record (LDS for small scenes); the chain differs per lane.
RT_DEV V3 turn_y(float s, float c, V3 d) {  // rotate.clj:14-24, the ray side
  return mk(fmaf(c, d.x, -(s * d.z)), d.y, fmaf(s, d.x, c * d.z));
}
RT_DEV V3 map_rec(float A, float B, V3 n) {  // rotate.clj:32-42 composed, the record side
  return mk(fmaf(A, n.x, B * n.z), n.y, fmaf(B, n.x, A * n.z));
}
template <class X, class CHAIN, class T>
RT_DEV void chain_unwind_rec(CHAIN C, V3 world_d, Rec& rec, T& tally) {
  const int n_ops = C->n_ops;
  if (n_ops == 0) return;
  if constexpr (T::on) {
    const int n_rot = C->n_rotate;
    tally.template add<RT_EV_ROTATE_REC>((uint32_t)n_rot);
    tally.template add<RT_EV_TRANSLATE_REC>((uint32_t)(n_ops - n_rot));
  }
  const float A = C->A, B = C->B, s0 = C->s0, c0 = C->c0;
  const V3 p = rec.p;
  rec.p = mk(fmaf(A, p.x, fmaf(B, p.z, C->tx)), p.y + C->ty, fmaf(B, p.x, fmaf(A, p.z, C->tz)));
  // inside the outermost wrapper: the normal as the second-outermost wrapper oriented it
  V3 n1 = map_rec(C->A1, C->B1, rec.n);
  const bool keep = dot(turn_y(C->s01, C->c01, world_d), n1) < 0.0f || C->inner == 0;
  n1 = keep ? n1 : neg(n1);
  // the outermost wrapper
  const V3 nn = map_rec(c0, s0, n1);
  rec.front = dot(turn_y(s0, c0, world_d), nn) < 0.0f;
  rec.n = rec.front ? nn : neg(nn);
}

// The same for a rect, from the constants its DPrimInfo carries (see there): on entry rec.front says how the rect's own
// face test oriented its axis normal (rectangles.clj:37), and every wrapper can only flip the mapped normal.
template <class INFO, class T>
RT_DEV void rect_unwind(INFO F, V3 world_d, Rec& rec, T& tally) {
  const V3 p = rec.p;
  rec.p = mk(fmaf(F->fA, p.x, fmaf(F->fB, p.z, F->ftx)), p.y + F->fty, fmaf(F->fB, p.x, fmaf(F->fA, p.z, F->ftz)));
  const V3 nn = mk(F->fnx, F->fny, F->fnz);
  const float q0 = dot(world_d, mk(F->g0x, F->g0y, F->g0z));
  const float q1 = dot(world_d, mk(F->g1x, F->g1y, F->g1z));
  const bool sm = rec.front;  // the sign the innermost test gave the normal (true: as stored)
  // the second-outermost wrapper keeps what it is handed when that faces its ray, else flips it (rotate.clj:43,
  // translate.clj:13); a single wrapper has no such inner test
  const bool keep = F->finner == 0 || (sm ? q1 < 0.0f : q1 > 0.0f);
  const bool s1 = keep ? sm : !sm;
  const bool front = s1 ? q0 < 0.0f : q0 > 0.0f;
  rec.front = front;
  rec.n = (front == s1) ? nn : neg(nn);
}

template <class X, class T>
RT_DEV void chain_unwind(const X& ctx, int chain, V3 world_d, Rec& rec, T& tally) {
  if constexpr (X::has(F_CHAINS)) {
    if (chain == 0) return;
    if constexpr (X::has(F_SMALL)) chain_unwind_rec<X>(ctx.lds_chains + chain, world_d, rec, tally);
    else chain_unwind_rec<X>(cptr(ctx.S.chain_recs) + chain, world_d, rec, tally);
  }
}

// -------------------------------------------------------------------------------- media
// t + 0.0001 (constant_medium.clj:13) for an FP32 t: where the sum is absorbed (|t| >= 2048) the next
// float above t, so that the first boundary hit itself stays excluded as it is in FP64.
RT_DEV float after_t(float t) {
  const float s = t + 0.0001f;
  const float up = __int_as_float(__float_as_int(t) + (t > 0.0f ? 1 : -1));
  return s > t ? s : up;
}
// constant_medium.clj:11-38 for medium m; xi is the uniform of line 32 (scripted when have_xi,
// else word m & 3 of `block`, the keyed stream's block m >> 2 for this bounce). True with the hit parameter when the free-flight distance
// ends inside the boundary, t in [t1, t2] within (tmin, tmax]. The boundary groups carry their
// full chains, so they take the world ray; ray length and t do not depend on the chain.
template <class X, class T>
RT_DEV bool medium_t(const X& ctx, int m, const Ray& wr, float tmin, float tmax, float xi, bool have_xi,
                     const uint32_t* block, float& t_out, T& tally) {
  const DScene& S = ctx.S;
  const DMedium RT_CONST& M = cptr(S.media)[m];
  const float INF = __builtin_inff();
  const int first_group = M.first_group, n_groups = M.n_groups;
  // rec1 = hit boundary (-inf, +inf); rec2 = hit boundary (rec1.t + 0.0001, +inf)
  Best b1{INF, -1, 0.0}, b2{INF, -1, 0.0};
  bool single_sphere = false;
  if constexpr (X::has(F_SPHERES)) {
    // A boundary that is one sphere (every medium the reference ships): both queries read the
    // same two roots, so the quadratic is solved once and the range tests of sphere_t are
    // applied twice, exactly as two calls would.
    const DGroup RT_CONST& G = cptr(S.groups)[first_group];
    single_sphere = n_groups == 1 && G.bvh_root < 0 && G.n_sph == 1 && (G.n_xy | G.n_xz | G.n_yz) == 0;
    if (single_sphere) {
      const RayX r = group_ray<X>(S, first_group, G.flags, G.rc, G.rs, G.bx, G.by, G.bz, wr, tally);
      const DPrim RT_CONST& P = cptr(S.prims)[G.first_prim];
      double root1, root2;
      if (!sphere_roots<sphere_mode<X>()>(P, prim_type(P.type), r, root1, root2, tally)) return false;
      const double NINF = -(double)INF;
      double first;
      if (root1 > NINF && (float)root1 < INF) first = root1;
      else if (root2 > NINF && (float)root2 < INF) first = root2;
      else return false;
      b1.t = (float)first;
      // constant_medium.clj:13 adds 0.0001 to the FP64 root: in FP32 the sum is absorbed from
      // |t| = 2048 on and the entry root would be accepted again as the exit
      const double tmin2 = first + 0.0001;
      if (root1 > tmin2 && (float)root1 < INF) b2.t = (float)root1;
      else if (root2 > tmin2 && (float)root2 < INF) b2.t = (float)root2;
      else return false;
      b1.prim = b2.prim = G.first_prim;
    }
  }
  // A boundary that is one box: both queries read the same slab test (the second one's entry
  // point can only be the first one's, which its t-min excludes, so it answers with the exit).
  bool single_box = false;
  {
    const DGroup RT_CONST& G = cptr(S.groups)[first_group];
    single_box = n_groups == 1 && G.bvh_root < 0 && G.n_sph == 0 && G.loose == (1 << 24);
    if (single_box) {
      const RayX r = group_ray<X>(S, first_group, G.flags, G.rc, G.rs, G.bx, G.by, G.bz, wr, tally);
      const float RT_CONST* B = cptr(S.brute_boxes) + G.first_box * RT_BOX_STRIDE;
      float w[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) w[q] = B[q];
      tally.template add<RT_EV_BOX_SLAB>();
      const BoxSlab s = box_slab(w, r);
      if (!box_accept(s, -INF, INF, b1.t)) return false;
      if (!box_accept(s, after_t(b1.t), INF, b2.t)) return false;
      b1.prim = b2.prim = G.first_prim;
    }
  }
  if (!single_sphere && !single_box) {
    // any other boundary: two closest-hit queries over its groups. Only the instantiations for
    // worlds that have such a medium carry these loops (the host sets F_MEDIA_ANY).
    if constexpr (X::has(F_MEDIA_ANY)) {
      for (int gi = 0; gi < n_groups; ++gi) group_hit(ctx, first_group + gi, wr, -INF, b1, tally, -(double)INF);
      if (b1.prim < 0) return false;
      // (spheres compare against the winner's FP64 root + 0.0001 when a sphere won)
      const double tmin2 = b1.prim >= 0 && !is_rect(prim_type(cptr(S.prims)[b1.prim].type)) ? b1.root + 0.0001 : (double)after_t(b1.t);
      for (int gi = 0; gi < n_groups; ++gi) group_hit(ctx, first_group + gi, wr, after_t(b1.t), b2, tally, tmin2);
      if (b2.prim < 0) return false;
    } else {
      return false;  // a medium the root never reaches (no boundary groups)
    }
  }
  float t1 = (b1.t < tmin) ? tmin : b1.t;
  float t2 = (b2.t > tmax) ? tmax : b2.t;
  if (!(t1 < t2)) return false;
  if (t1 < 0.0f) t1 = 0.0f;
  if (!have_xi) {
    const uint32_t w = (m & 3) == 0 ? block[0] : (m & 3) == 1 ? block[1] : (m & 3) == 2 ? block[2] : block[3];
    xi = u24(w);
  }
  tally.template add<RT_EV_MEDIUM_SAMPLE>();
  float ray_length = sqrtf(len2(wr.d));
  float distance_inside = (t2 - t1) * ray_length;
  float hit_distance = M.neg_inv_density * logf(xi);
  if (!(hit_distance <= distance_inside)) return false;
  t_out = t1 + hit_distance / ray_length;
  return true;
}

// ---------------------------------------------------------------------------- closest hit
// (hit (->HittableList world) r t-min t-max {}) of core.clj:23 over the flattened world.
// medium_xi: scripted uniforms per medium (parity hook) or nullptr to draw from `g`;
// skip_media drops MEDIUM nodes altogether.
// The constant media of the world against the closest surface hit so far (constant_medium.clj:11-44
// for each, in medium order).
template <class X, class T>
RT_DEV void media_hit(const X& ctx, const Ray& wr, float tmin, const Rng& g, uint32_t bounce, const float* medium_xi,
                      Best& best, T& tally) {
  if constexpr (X::has(F_MEDIA)) {
    const int n_media = ctx.S.n_media;
    // one block of the keyed stream serves four media: drawn here for every lane, once, instead
    // of inside each medium for the lanes that cross it
    uint32_t block[4] = {0u, 0u, 0u, 0u};
    for (int m = 0; m < n_media; ++m) {
      if (medium_xi == nullptr && (m & 3) == 0) philox(g, (uint32_t)m >> 2, bounce, P_MEDIUM, block);
      float tm;
      if (medium_t(ctx, m, wr, tmin, best.t, medium_xi ? medium_xi[m] : 0.f, medium_xi != nullptr, block,
                   tm, tally)) {
        best.t = tm; best.prim = -2 - m;
      }
    }
  }
}

// The HitRecord of the winner `best` (best.prim != -1) for the world ray wr.
template <class X, class T>
RT_DEV void hit_record(const X& ctx, const Ray& wr, const Best& best, bool want_uv_always, Rec& rec, T& tally) {
  const DScene& S = ctx.S;
  if (!X::has(F_MEDIA) || best.prim >= 0) {
    const int prim = best.prim;
    const DPrimInfo I = load_info(ctx, prim);
    RayX rx = group_ray<X>(S, prim_group(I.type_group), I.gflags, I.rc, I.rs, I.bx, I.by, I.bz, wr, tally);
    bool want_uv = want_uv_always;
    if constexpr (X::has(F_IMAGE)) want_uv = want_uv || (I.mflags & DMAT_NEEDS_UV) != 0;
    prim_record<X>(S, prim, I, rx, best.t, best.root, want_uv, rec, tally);
    RT_MARK(unwind);
    if constexpr (X::has(F_CHAINS) && X::rects()) {
      // a rect below a chain takes its record back up from constants of its own (one lookup, no chain record)
      if ((!X::has(F_SPHERES) || is_rect(prim_type(I.type_group))) && I.chain != 0) {
        if constexpr (T::on) {
          const int n_ops = cptr(S.chain_recs)[I.chain].n_ops, n_rot = cptr(S.chain_recs)[I.chain].n_rotate;
          tally.template add<RT_EV_ROTATE_REC>((uint32_t)n_rot);
          tally.template add<RT_EV_TRANSLATE_REC>((uint32_t)(n_ops - n_rot));
        }
        if constexpr (X::has(F_SMALL)) rect_unwind(ctx.lds_info + prim, wr.d, rec, tally);
        else rect_unwind(cptr(S.prim_info) + prim, wr.d, rec, tally);
      } else {
        chain_unwind(ctx, I.chain, wr.d, rec, tally);
      }
    } else {
      chain_unwind(ctx, I.chain, wr.d, rec, tally);
    }
  } else {
    if constexpr (X::has(F_MEDIA)) {
      // constant_medium.clj:38-44: p = ray-at r t, normal [1 0 0], front-face true, no u/v
      const DMedium RT_CONST& M = cptr(S.media)[-2 - best.prim];
      const int outer = M.outer_chain;
      // the medium's own space: its outer chain's motion, kept in a one-group header per medium
      const DGroup RT_CONST& OG = cptr(S.groups)[M.outer_group];
      RayX mr = group_ray<X>(S, M.outer_group, 0, OG.rc, OG.rs, OG.bx, OG.by, OG.bz, wr, tally);
      rec.t = best.t;
      rec.p = mk(fmaf(best.t, mr.d.x, mr.o.x), fmaf(best.t, mr.d.y, mr.o.y), fmaf(best.t, mr.d.z, mr.o.z));
      rec.n = mk(1.f, 0.f, 0.f);
      rec.front = true;
      rec.mat = M.phase_mat;
      const DMat RT_CONST& PM = cptr(S.mats)[M.phase_mat];
      rec.mtype = PM.type; rec.mflags = PM.flags; rec.mtex = PM.tex;
      rec.rgb = mk(PM.r, PM.g, PM.b); rec.mx = PM.x;
      rec.u = rec.v = 0.f;
      chain_unwind(ctx, outer, wr.d, rec, tally);
    }
  }
}

template <class X, class T>
RT_DEV bool closest_hit(const X& ctx, const Ray& wr, float tmin, float tmax, const Rng& g, uint32_t bounce,
                        const float* medium_xi, bool skip_media, bool want_uv_always, Rec& rec, T& tally,
                        unsigned long long* g_stamp_inner = nullptr) {
  const DScene& S = ctx.S;
  Best best{tmax, -1, 0.0};
  const int n_surface = S.n_surface_groups;
#ifdef RT_STAMPS
  for (int gi = 0; gi < n_surface; ++gi) {
    __builtin_amdgcn_sched_barrier(0); const unsigned long long g0 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
    group_hit<true>(ctx, gi, wr, tmin, best, tally, (double)tmin);
    __builtin_amdgcn_sched_barrier(0); const unsigned long long g1 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
    if (ctx.st_grp) ctx.st_grp[gi < 6 ? gi : 6] += g1 - g0;
  }
#else
  for (int gi = 0; gi < n_surface; ++gi) group_hit<true>(ctx, gi, wr, tmin, best, tally, (double)tmin);
#endif
#ifdef RT_STAMPS
  if (g_stamp_inner) { __builtin_amdgcn_sched_barrier(0); *g_stamp_inner = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#endif
  RT_MARK(media);
#ifdef RT_STAMPS
  __builtin_amdgcn_sched_barrier(0); const unsigned long long m0 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#endif
  if (!skip_media) media_hit(ctx, wr, tmin, g, bounce, medium_xi, best, tally);
#ifdef RT_STAMPS
  __builtin_amdgcn_sched_barrier(0); const unsigned long long m1 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
  if (ctx.st_grp) ctx.st_grp[7] += m1 - m0;
#endif
  if (best.prim == -1) return false;
  RT_MARK(record);
  hit_record(ctx, wr, best, want_uv_always, rec, tally);
  return true;
}

// ------------------------------------------------------------------------------- textures
RT_DEV float hermite(float x) { return x * x * (3.0f - 2.0f * x); }

// noise.clj:39-58 + :21-33. Both Hermite passes and the OR of the three permutation entries are the
// reference's. Tables are read from LDS.
// The point is an FP32 hit point times a power of two (perlin_turb), so cell and fraction come out of FP32
// exactly as they would out of FP64: the floor is exact, and p - floor(p) is one correctly rounded subtraction
// of two exact operands either way (the FP64 difference is exact, or within 2^-29 of 1 and rounds to 1 like
// the FP32 one). Sixty FP64-rate instructions per octave (three of them conversions to 64-bit integers)
// become twelve FP32 ones with the same results bit for bit.
template <class X, class T>
RT_DEV float perlin_noise(const X& ctx, int pid, V3 p, T& tally) {
  tally.template add<RT_EV_PERLIN_NOISE>();
  const float fx = __builtin_floorf(p.x), fy = __builtin_floorf(p.y), fz = __builtin_floorf(p.z);
  float u = hermite(p.x - fx), v = hermite(p.y - fy), w = hermite(p.z - fz);
  int i = (int)fx, j = (int)fy, k = (int)fz;
  const uint8_t* perm = ctx.lds_perlin_perm + pid * 768;
  // (a gradient is three floats at a 16-byte stride: three registers per corner in flight, not four)
  struct G3 { float x, y, z, pad; };
  const G3* vec = reinterpret_cast<const G3*>(ctx.lds_perlin_vec) + pid * 256;
  float uu = hermite(u), vv = hermite(v), ww = hermite(w);
  float acc = 0.f;
#pragma unroll
  for (int di = 0; di < 2; ++di) {
    int px = perm[(i + di) & 255];
    float bu = di ? uu : 1.0f - uu;
#pragma unroll
    for (int dj = 0; dj < 2; ++dj) {
      int py = perm[256 + ((j + dj) & 255)];
      float bv = dj ? vv : 1.0f - vv;
#pragma unroll
      for (int dk = 0; dk < 2; ++dk) {
        int pz = perm[512 + ((k + dk) & 255)];
        float bw = dk ? ww : 1.0f - ww;
        const G3* g = vec + (px | py | pz);
        const float cx = g->x, cy = g->y, cz = g->z;
        float dp = fmaf(cx, u - di, fmaf(cy, v - dj, cz * (w - dk)));
        acc = fmaf((bu * bv) * bw, dp, acc);
      }
    }
  }
  return acc;
}

// noise.clj:59-70
template <class X, class T>
RT_DEV float perlin_turb(const X& ctx, int pid, V3 p, int depth, T& tally) {
  float acc = 0.f, weight = 1.f;
  V3 tp = p;
  for (int i = 0; i < depth; ++i) {
    acc = fmaf(weight, perlin_noise(ctx, pid, tp, tally), acc);
    weight *= 0.5f;
    tp = mk(tp.x * 2.0f, tp.y * 2.0f, tp.z * 2.0f);  // exact
  }
  return fabsf(acc);
}

// sign of sin(x) for FP64 x = 10 * coordinate: negative iff floor(x / pi) is odd.
RT_DEV bool sin_negative(double x) {
  const double q = floor(x * 0.3183098861837907);
  // q odd: half of it is not a whole number (a conversion to a 64-bit integer is a dozen instructions here)
  const double h = q * 0.5;
  return h != floor(h);
}

// A hit point ON a coordinate plane through the origin (a rect with k = 0): the reference's point o + t d
// (FP64, ray-at) misses that plane by a rounding error or lands on it, and a checker reads the SIGN of
// sin(10 coordinate) there (texture.clj:25-30) - the sign of that rounding error, a speckle of both colours.
// The same three FP64 operations on the incoming ray give a sign of the same kind: +-1e-30, or 0.
RT_DEV float plane_residual(float o, float d) {
  const double oc = (double)o, dc = (double)d;
  const double t = (0.0 - oc) / dc;
  const double pc = oc + t * dc;
  return pc > 0.0 ? 1e-30f : (pc < 0.0 ? -1e-30f : 0.0f);
}

// The residual for the first coordinate of p that is exactly 0 (0 when none is).
RT_DEV float zero_plane_residual(V3 p, V3 ro, V3 rd) {
  if (!(p.x == 0.0f || p.y == 0.0f || p.z == 0.0f)) return 0.0f;
  const bool zx = p.x == 0.0f, zy = !zx && p.y == 0.0f;
  return plane_residual(zx ? ro.x : (zy ? ro.y : ro.z), zx ? rd.x : (zy ? rd.y : rd.z));
}

// Where the residual is worked out: at the head of `scatter` (one live value through the material code) or here,
// inside the checker's branch (the incoming ray stays live instead). Which is cheaper is the register allocator's
// call and differs by kernel: the noise-only kernels measured 1-2 % faster with the second, the others with the first.
template <class X>
constexpr bool residual_in_checker() { return X::has(F_NOISE) && !X::has(F_BVH); }

// texture.clj: value of texture `tex` at (u, v, p); resid = zero_plane_residual of the hit (ro, rd: its ray).
template <class X, class T>
RT_DEV V3 texture_value(const X& ctx, int tex, float u, float v, V3 p, float resid, V3 ro, V3 rd, T& tally) {
  const DScene& S = ctx.S;
  const DTex RT_CONST* texs = cptr(S.texs);
  // a texture's whole record in one fetch (its fields are not waited for one after the other)
  struct TexRec { int type, i0, i1; float r, g, b; int w, h; };
  auto fetch = [&](int t) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 RT_CONST* q = reinterpret_cast<const u4 RT_CONST*>(texs + t);
    const u4 a = q[0], b = q[1];
    return TexRec{(int)a.x, (int)a.y, (int)a.z, __uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y), (int)b.z, (int)b.w};
  };
  TexRec R = fetch(tex);
  if (R.type == RT_TEX_CHECKER) {
    if constexpr (residual_in_checker<X>()) resid = zero_plane_residual(p, ro, rd);
    if (resid != 0.0f) {
      // (one coordinate: a point on two such planes at once is an edge)
      const bool zx = p.x == 0.0f, zy = !zx && p.y == 0.0f, zz = !zx && !zy && p.z == 0.0f;
      p = mk(zx ? resid : p.x, zy ? resid : p.y, zz ? resid : p.z);
    }
  }
  for (int depth = 0; R.type == RT_TEX_CHECKER && depth < 16; ++depth) {  // :25-30
    tally.template add<RT_EV_CHECKER>();
    bool nx = sin_negative(10.0 * (double)p.x), ny = sin_negative(10.0 * (double)p.y),
         nz = sin_negative(10.0 * (double)p.z);
    // the product of the three sines is negative iff an odd number of them is, and is +-0 (not
    // negative) when a coordinate is exactly 0
    bool negative = ((nx != ny) != nz) && p.x != 0.0f && p.y != 0.0f && p.z != 0.0f;
    R = fetch(negative ? R.i0 : R.i1);
  }
  if (R.type == RT_TEX_SOLID) return mk(R.r, R.g, R.b);  // :15-17
  if constexpr (X::has(F_NOISE)) {
    if (R.type == RT_TEX_NOISE) {  // :41-44
      tally.template add<RT_EV_NOISE_TEX>();
      float k = 0.5f * (1.0f + sinf(fmaf((float)R.i1, p.z, 10.0f * perlin_turb(ctx, R.i0, p, 7, tally))));
      return mk(k, k, k);
    }
  }
  if (X::has(F_IMAGE) && R.type == RT_TEX_IMAGE) {  // :55-66
    tally.template add<RT_EV_IMAGE_TEX>();
    if (R.i0 < 0) return mk(0.f, 1.f, 1.f);
    const int w = R.w, h = R.h, first_texel = R.i1;
    float uu = (0.0f > u) ? 0.0f : (u > 1.0f ? 1.0f : u);   // clamp, base.clj:307-310 (NaN stays)
    float vc = (0.0f > v) ? 0.0f : (v > 1.0f ? 1.0f : v);
    float vv = 1.0f - vc;
    float fi = uu * (float)w, fj = vv * (float)h;
    int i = (fi != fi) ? 0 : (int)fi;  // (int NaN) = 0 on the JVM
    int j = (fj != fj) ? 0 : (int)fj;
    if (i >= w) i = w - 1;
    if (j >= h) j = h - 1;
    uint32_t px = cptr(S.texels)[first_texel + j * w + i];
    const float cs = 1.0f / 255.0f;
    return mk((float)(px & 255u) * cs, (float)((px >> 8) & 255u) * cs, (float)((px >> 16) & 255u) * cs);
  }
  return mk(0.f, 0.f, 0.f);
}

// The colour of a textured material: solid colours are folded into the material record.
template <class X, class T>
RT_DEV V3 material_colour(const X& ctx, int flags, int tex, V3 solid, const Rec& rec, float resid, const Ray& r, T& tally) {
  if constexpr (X::has(F_TEXTURES)) {
    if (!(flags & DMAT_SOLID)) return texture_value(ctx, tex, rec.u, rec.v, rec.p, resid, r.o, r.d, tally);
  }
  return solid;
}

// ------------------------------------------------------------------------------ materials
RT_DEV V3 reflect(V3 v, V3 n) { return madd(n, -2.0f * dot(v, n), v); }  // base.clj:199-200
RT_DEV V3 refract(V3 uv, V3 n, float eta) {                           // base.clj:202-209
  float cos_theta = fminf(dot(neg(uv), n), 1.0f);
  V3 perp = madd(n, cos_theta, uv) * eta;
  return madd(n, -sqrtf(fabsf(1.0f - len2(perp))), perp);
}
RT_DEV float reflectance(float cosine, float ref_idx) {  // base.clj:215-220
  float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
  r0 = r0 * r0;
  float m = 1.0f - cosine;
  float m2 = m * m;
  return fmaf(1.0f - r0, m2 * m2 * m, r0);
}

// (scatter mat r rec) and (emitted mat u v p) of materials.clj. Returns ok; on ok the ray is
// replaced by the scattered ray. `emitted` is only meaningful when !ok (core.clj:26-29).
RT_DEV bool needs_sphere_draw(int mtype) {
  return mtype == RT_MAT_LAMBERTIAN || mtype == RT_MAT_METAL || mtype == RT_MAT_ISOTROPIC;
}
// With PRE, `sv` is random-in-unit-sphere for this bounce, drawn by the caller when
// needs_sphere_draw(mtype); without it the material draws its own.
// The branches only produce the scattered DIRECTION; the one assignment of the scattered ray (origin
// = the hit point, time kept) follows them. (With the origin assigned inside every branch the
// compiler kept it in a register that the dielectric branch's random draw reused: reflected glass
// bounces came out with a garbage origin in one kernel instantiation.)
template <bool PRE, class X, class T>
RT_DEV bool scatter(const X& ctx, const Rec& rec, Ray& r, const Rng& g, uint32_t bounce, V3 sv, V3& attenuation,
                    V3& emitted, T& tally) {
  const int type = rec.mtype, flags = rec.mflags, tex = rec.mtex;
  const V3 rgb = rec.rgb;
  emitted = mk(0.f, 0.f, 0.f);
  attenuation = mk(0.f, 0.f, 0.f);
  V3 dir = r.d;
  bool ok = false;
  float resid = 0.0f;  // see plane_residual: only a textured material on a plane through the origin has one
  if constexpr (X::has(F_TEXTURES) && !residual_in_checker<X>()) {
    if (!(flags & DMAT_SOLID)) resid = zero_plane_residual(rec.p, r.o, r.d);
  }
  if (type == RT_MAT_LAMBERTIAN) {  // :16-23
    tally.template add<RT_EV_LAMBERT>();
    dir = rec.n + unit(PRE ? sv : random_in_unit_sphere(g, bounce, tally));  // random-unit-vector, base.clj:196-197
    attenuation = material_colour(ctx, flags, tex, rgb, rec, resid, r, tally);
    ok = true;
  } else if (type == RT_MAT_DIFFUSE_LIGHT) {  // :69-78 (the unused scatter direction is not drawn)
    tally.template add<RT_EV_LIGHT>();
    emitted = material_colour(ctx, flags, tex, rgb, rec, resid, r, tally);
    attenuation = emitted;
  } else if (X::has(F_SPECULAR) && type == RT_MAT_METAL) {  // :33-39
    if constexpr (X::has(F_SPECULAR)) {
      tally.template add<RT_EV_METAL>();
      V3 reflected = reflect(unit(r.d), rec.n);
      dir = madd(PRE ? sv : random_in_unit_sphere(g, bounce, tally), rec.mx, reflected);
      attenuation = rgb;
      ok = dot(dir, rec.n) > 0.0f;
    }
  } else if (X::has(F_SPECULAR) && type == RT_MAT_DIELECTRIC) {  // :47-61
    if constexpr (X::has(F_SPECULAR)) {
      tally.template add<RT_EV_DIELECTRIC>();
      float ir = rec.rgb.x;
      float ratio = rec.front ? (1.0f / ir) : ir;
      V3 ud = unit(r.d);
      float cos_theta = fminf(dot(neg(ud), rec.n), 1.0f);
      float sin_theta = sqrtf(fmaf(-cos_theta, cos_theta, 1.0f));
      bool reflect_it = ratio * sin_theta > 1.0f;
      if (!reflect_it) {  // `or` short-circuits: the uniform is drawn only here
        uint32_t o4[4];
        philox(g, 0, bounce, P_DIEL, o4);
        reflect_it = reflectance(cos_theta, ratio) > u24(o4[0]);
      }
      dir = reflect_it ? reflect(ud, rec.n) : refract(ud, rec.n, ratio);
      attenuation = mk(1.f, 1.f, 1.f);
      ok = true;
    }
  } else if (X::has(F_MEDIA) && type == RT_MAT_ISOTROPIC) {  // :87-90 — direction not normalised
    if constexpr (X::has(F_MEDIA)) {
      tally.template add<RT_EV_ISOTROPIC>();
      dir = PRE ? sv : random_in_unit_sphere(g, bounce, tally);
      attenuation = material_colour(ctx, flags, tex, rgb, rec, resid, r, tally);
      ok = true;
    }
  }
  // (any other type: the host picks an instantiation that covers the scene)
  r.o = rec.p;   // the scattered ray starts at the hit point; a caller only reads it when ok
  r.d = dir;
  return ok;
}

// --------------------------------------------------------------------------------- camera
// core.clj:96-102 + ray.clj:14-26 in FP64 (once per path), rounded to the FP32 path state
// (the divisions by W-1 and H-1 through exact_div and the reciprocals the host rounded).
// `camp` points at a DCamera in any address space (the kernarg segment for the render kernel).
template <class CAMP, class T>
RT_DEV Ray camera_ray(CAMP camp, int x, int y, const Rng& g, T& tally) {
  const auto& cam = *camp;
  tally.template add<RT_EV_CAMERA_RAY>();
  uint32_t o4[4];
  philox(g, 0, 0, P_CAM, o4);
  if (cam.pinhole) {  // wave-uniform
    const double U = 5.9604644775390625e-8;  // 2^-24
    const double sx = fma((double)(o4[0] >> 8), U, (double)x), ty = fma((double)(o4[1] >> 8), U, (double)y);
    Ray r;
    r.o = mk(cam.fO[0], cam.fO[1], cam.fO[2]);
    r.d = mk((float)fma(cam.dV[0], ty, fma(cam.dH[0], sx, cam.dC[0])), (float)fma(cam.dV[1], ty, fma(cam.dH[1], sx, cam.dC[1])),
             (float)fma(cam.dV[2], ty, fma(cam.dH[2], sx, cam.dC[2])));
    r.tm = (float)fma(cam.time1 - cam.time0, (double)u24(o4[2]), cam.time0);
    return r;
  }
  double s = exact_div((double)x + (double)u24(o4[0]), cam.wm1, cam.inv_wm1);
  double t = exact_div((double)y + (double)u24(o4[1]), cam.hm1, cam.inv_hm1);
  double time_ = fma(cam.time1 - cam.time0, (double)u24(o4[2]), cam.time0);
  // random-in-unit-disk (base.clj:188-194): draw 0 never tested, first later draw inside the disk.
  // The reference draws it even when the lens radius is 0 (ray.clj:18); the result is then
  // multiplied by 0, so the addressed draws are simply not generated.
  double rx = 0.0, ry = 0.0;
  if (cam.lens_radius != 0.0) {
    double dx = 0.0, dy = 0.0;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (uint32_t k = 1; k < RT_MAX_TRIALS; ++k) {
      uint32_t d4[4];
      philox(g, k >> 1, 0, P_DISK, d4);
      float ux = u24((k & 1) ? d4[2] : d4[0]), uy = u24((k & 1) ? d4[3] : d4[1]);
      dx = -1.0 + 2.0 * (double)ux;
      dy = -1.0 + 2.0 * (double)uy;
      tally.template add<RT_EV_DISK_TRIAL>();
      if (fma(dx, dx, dy * dy) < 1.0) break;
    }
    rx = dx * cam.lens_radius;
    ry = dy * cam.lens_radius;
  }
  double off[3], org[3], dir[3];
  if (cam.lens_radius != 0.0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      off[a] = fma(cam.u[a], rx, cam.v[a] * ry);
      org[a] = cam.origin[a] + off[a];
      dir[a] = (fma(cam.vertical[a], t, fma(cam.horizontal[a], s, cam.llc[a])) - cam.origin[a]) - off[a];
    }
  } else {
    // a pinhole (ray.clj:18-21 with lens-radius 0): the offset is u * 0 + v * 0 = +0, or -0 where an axis component
    // of u or v is negative; origin + (+-0) is the origin and x - (+-0) is x, unless x itself is a zero: the sign of
    // a zero direction component is not observable in what follows (its reciprocal is clamped, products with it are 0)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      org[a] = cam.origin[a];
      dir[a] = fma(cam.vertical[a], t, fma(cam.horizontal[a], s, cam.llc[a])) - cam.origin[a];
    }
  }
  Ray r;
  r.o = mk((float)org[0], (float)org[1], (float)org[2]);
  r.d = mk((float)dir[0], (float)dir[1], (float)dir[2]);
  r.tm = (float)time_;
  return r;
}

// core.clj:106-108,118,121-122 + java.awt.Color(float,float,float): in FP64 from the FP32 sums,
// so that the packed pixel is a pure function of the sums.
RT_DEV uint32_t pack_pixel(float r, float g, float b, int spp) {
  double scale = 1.0 / (double)spp;
  const double in[3] = {(double)r, (double)g, (double)b};
  uint32_t out[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double c = sqrt(in[i] * scale);
    c = (0.0 > c) ? 0.0 : (c > 0.999 ? 0.999 : c);
    float f = (float)c;
    double scaled = (double)(f * 255.0f) + 0.5;
    int iv = (scaled != scaled) ? 0 : (int)scaled;
    out[i] = (uint32_t)iv & 0xFFu;
  }
  return 0xFF000000u | (out[0] << 16) | (out[1] << 8) | out[2];
}

}  // namespace rt
#endif  // RT_DEVICE_HPP
