// rt_device.hpp — device functions of the path tracer (gfx950, wave64).
//
// Everything the reference's `ray-color` calls, restated for one lane of a wavefront:
// intersection of the flattened world (rt_scene.h), the five materials, the four textures,
// Perlin noise, the samplers and the camera. Arithmetic is FP32 except where the reference's
// FP64 results cannot be reached otherwise (sphere quadratic, sphere u/v, camera ray, Perlin
// cell fractions, checker sign). Each function cites the reference lines it implements
// (paths relative to src/raytracing/).
#ifndef RT_DEVICE_HPP
#define RT_DEVICE_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtmi.h"
#include "rt_scene.h"

namespace rt {

#define RT_DEV __device__ __forceinline__

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
RT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV float len2(V3 a) { return dot(a, a); }
RT_DEV V3 unit(V3 a) { return a * (1.0f / sqrtf(len2(a))); }  // base.clj:39-40 (v * 1/|v|)

struct D3 {
  double x, y, z;
};
RT_DEV D3 mkd(double x, double y, double z) { return D3{x, y, z}; }
RT_DEV D3 tod(V3 a) { return mkd(a.x, a.y, a.z); }
RT_DEV V3 tof(D3 a) { return mk((float)a.x, (float)a.y, (float)a.z); }
RT_DEV double ddot(D3 a, D3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

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
  template <int E>
  RT_DEV void add(uint32_t = 1) {}
};
template <>
struct Tally<true> {
  uint32_t ev[RT_EV_COUNT];
  RT_DEV void clear() {
#pragma unroll
    for (int i = 0; i < RT_EV_COUNT; ++i) ev[i] = 0;
  }
  template <int E>
  RT_DEV void add(uint32_t n = 1) { ev[E] += n; }
};

// ---------------------------------------------------------------------------------- RNG
// Philox-4x32-10 (Salmon et al., SC'11). A draw is addressed, not sequenced:
//   key     = (pixel index y*W+x, global sample index)
//   counter = (block, bounce << 4 | purpose, seed_lo, seed_hi)
// so its value does not depend on evaluation order or on which GPU runs the sample.
// Uniforms are the top 24 bits / 2^24: exactly representable in FP32 and FP64.
enum { P_CAM = 0, P_DISK = 1, P_SPHERE = 2, P_DIEL = 3, P_MEDIUM = 4 };
#define RT_MAX_TRIALS 256  // rejection loops give up here so every lane reaches the exit

struct Rng {
  uint32_t k0, k1, s0, s1;
};

RT_DEV void philox(const Rng& g, uint32_t blk, uint32_t bounce, uint32_t purpose, uint32_t out[4]) {
  uint32_t c0 = blk, c1 = (bounce << 4) | purpose, c2 = g.s0, c3 = g.s1;
  uint32_t k0 = g.k0, k1 = g.k1;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
RT_DEV float u24(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }

// base.clj:182-186: the reduce seeds its accumulator with draw 0 and never tests it, then
// keeps the first later draw with |v|^2 >= 1 (the inverse of the textbook test). Draw 0 is
// addressed but never generated here.
template <class T>
RT_DEV V3 random_in_unit_sphere(const Rng& g, uint32_t bounce, T& tally) {
  V3 v = mk(0.f, 0.f, 0.f);
  for (uint32_t k = 1; k < RT_MAX_TRIALS; ++k) {
    uint32_t o[4];
    philox(g, k, bounce, P_SPHERE, o);
    v = mk(fmaf(2.0f, u24(o[0]), -1.0f), fmaf(2.0f, u24(o[1]), -1.0f), fmaf(2.0f, u24(o[2]), -1.0f));
    tally.template add<RT_EV_SPHERE_TRIAL>();
    if (len2(v) >= 1.0f) break;
  }
  return v;
}
template <class T>
RT_DEV V3 random_unit_vector(const Rng& g, uint32_t bounce, T& tally) {  // base.clj:196-197
  return unit(random_in_unit_sphere(g, bounce, tally));
}

// ------------------------------------------------------------------------ scene context
struct Ctx {
  DScene S;
  const float* lds_perlin_vec;      // n_perlin * 1024 floats (LDS), or nullptr -> global tables
  const uint8_t* lds_perlin_perm;   // n_perlin * 768 bytes (LDS)
  int* lds_stack;                   // BVH stack base for this lane (stride = blockDim.x)
  int stack_stride;
};

// --------------------------------------------------------------------------- primitives
struct Best {
  float t;     // closest so far (the reference's closest-so-far, base.clj:94-97)
  int prim;    // device primitive index, -1 = none, <= -2 = medium (-2 - m)
  int group;   // group of the winner; bit 30 set = a sphere won with its second candidate root
};
#define RT_SECOND_ROOT (1 << 30)

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
  r.inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  r.o_lo = r.d_lo = mk(0.f, 0.f, 0.f);
  return r;
}
RT_DEV D3 rayx_o64(const RayX& r) { return mkd((double)r.o.x + (double)r.o_lo.x, (double)r.o.y + (double)r.o_lo.y, (double)r.o.z + (double)r.o_lo.z); }
RT_DEV D3 rayx_d64(const RayX& r) { return mkd((double)r.d.x + (double)r.d_lo.x, (double)r.d.y + (double)r.d_lo.y, (double)r.d.z + (double)r.d_lo.z); }

RT_DEV bool is_rect(int type) { return type >= RT_PRIM_XY_RECT; }

// rectangles.clj:20-30 (XY), :57-67 (XZ), :94-104 (YZ): t from the constant axis, reject iff
// t < t-min or t > t-max (so equality with the closest-so-far is accepted and a later rect
// wins an exact tie), then the inclusive in-plane bounds.
RT_DEV bool rect_t(const DPrim& P, const RayX& r, float tmin, float tmax, float& t_out) {
  float oc, ic, oa, da, ob, db;
  if (P.type == RT_PRIM_XY_RECT) { oc = r.o.z; ic = r.inv.z; oa = r.o.x; da = r.d.x; ob = r.o.y; db = r.d.y; }
  else if (P.type == RT_PRIM_XZ_RECT) { oc = r.o.y; ic = r.inv.y; oa = r.o.x; da = r.d.x; ob = r.o.z; db = r.d.z; }
  else { oc = r.o.x; ic = r.inv.x; oa = r.o.y; da = r.d.y; ob = r.o.z; db = r.d.z; }
  float t = (P.f[4] - oc) * ic;
  if (t < tmin || t > tmax) return false;
  float a = fmaf(t, da, oa), b = fmaf(t, db, ob);
  if (a < P.f[0] || a > P.f[1] || b < P.f[2] || b > P.f[3]) return false;
  t_out = t;
  return true;
}

RT_DEV D3 sphere_center(const DPrim& P, float tm) {
  if (P.type == RT_PRIM_SPHERE) return mkd(P.d[0], P.d[1], P.d[2]);
  // base.clj:298-302
  double k = ((double)tm - P.d[6]) / (P.d[7] - P.d[6]);
  return mkd(P.d[0] + (P.d[3] - P.d[0]) * k, P.d[1] + (P.d[4] - P.d[1]) * k,
             P.d[2] + (P.d[5] - P.d[2]) * k);
}
RT_DEV double sphere_radius(const DPrim& P) { return P.type == RT_PRIM_SPHERE ? P.d[3] : P.d[8]; }

// base.clj:234-241,247 / :274-281,287 in FP64: both candidate roots. The second candidate is
// (-half_b + root1) / a, as the reference computes it (it adds the first root, not
// sqrt(discriminant)). False when the discriminant is not positive.
template <class T>
RT_DEV bool sphere_roots(const DPrim& P, const RayX& r, double& root1, double& root2, T& tally) {
  if (P.type == RT_PRIM_MOVING_SPHERE) tally.template add<RT_EV_MSPHERE_CENTER>();
  D3 c = sphere_center(P, r.tm);
  double radius = sphere_radius(P);
  D3 o = rayx_o64(r), d = rayx_d64(r);
  D3 oc = mkd(o.x - c.x, o.y - c.y, o.z - c.z);
  double a = ddot(d, d);
  double half_b = ddot(oc, d);
  double cc = ddot(oc, oc) - radius * radius;
  double disc = half_b * half_b - a * cc;
  tally.template add<RT_EV_SPHERE_TEST>();
  if (!(disc > 0.0)) return false;
  double sqrtd = sqrt(disc);
  root1 = (-half_b - sqrtd) / a;
  root2 = (-half_b + root1) / a;
  return true;
}

// The range tests of base.clj:242,248: t_min < root (FP64) and (float)root < t_max (strict, so
// the earlier of two coincident spheres keeps an exact tie).
template <class T>
RT_DEV bool sphere_t(const DPrim& P, const RayX& r, double tmin, float tmax, float& t_out, bool& second, T& tally) {
  double root1, root2;
  if (!sphere_roots(P, r, root1, root2, tally)) return false;
  tally.template add<RT_EV_SPHERE_ROOT>();
  if (root1 > tmin && (float)root1 < tmax) { t_out = (float)root1; second = false; return true; }
  tally.template add<RT_EV_SPHERE_ROOT>();
  if (root2 > tmin && (float)root2 < tmax) { t_out = (float)root2; second = true; return true; }
  return false;
}

// Exact-tie rule of the reference's linear scan (base.clj:92-101) for a traversal that visits
// primitives out of order: indices are scan-order ranks. A rect accepts t == closest, a sphere
// does not, so on a tie a rect beats any earlier hit and a sphere never displaces one.
RT_DEV bool wins_tie(int cand_type, int cand_idx, int best_type, int best_idx) {
  if (is_rect(cand_type)) return !is_rect(best_type) || cand_idx > best_idx;
  return !is_rect(best_type) && cand_idx < best_idx;
}

// One primitive against the running best. IN_ORDER = the caller visits primitives in scan order,
// where the comparisons of the reference give the tie rule by themselves.
template <bool IN_ORDER, class T>
RT_DEV void test_prim(const DScene& S, int idx, const DPrim& P, const RayX& r, float tmin, Best& best,
                      int group, T& tally) {
  float t;
  bool second = false;
  if (is_rect(P.type)) {
    tally.template add<RT_EV_RECT_TEST>();
    if (!rect_t(P, r, tmin, best.t, t)) return;
    if (!IN_ORDER && t == best.t && best.prim >= 0 &&
        !wins_tie(P.type, idx, S.prims[best.prim].type, best.prim)) return;
  } else {
    if (!sphere_t(P, r, (double)tmin, IN_ORDER ? best.t : __builtin_inff(), t, second, tally)) return;
    if (!IN_ORDER) {
      if (t > best.t) return;
      if (t == best.t) {
        if (best.prim < 0) return;  // t == t_max with nothing hit yet: the strict test fails
        if (!wins_tie(P.type, idx, S.prims[best.prim].type, best.prim)) return;
      }
    }
  }
  best.t = t; best.prim = idx; best.group = second ? (group | RT_SECOND_ROOT) : group;
}

// ------------------------------------------------------------------------------- chains
template <class T>
RT_DEV void chain_apply(const DScene& S, int chain, V3& o, V3& d, T& tally) {
  const DChain C = S.chains[chain];
  for (int i = 0; i < C.n_ops; ++i) {
    const DChainOp op = S.ops[C.first_op + i];
    if (op.kind == DOP_TRANSLATE) {  // translate.clj:11
      o = mk(o.x - op.a, o.y - op.b, o.z - op.c);
      tally.template add<RT_EV_TRANSLATE_RAY>();
    } else {  // rotate.clj:18-27
      // explicit FMAs: every inlined copy must round identically
      float s = op.a, c = op.b;
      o = mk(fmaf(c, o.x, -(s * o.z)), o.y, fmaf(s, o.x, c * o.z));
      d = mk(fmaf(c, d.x, -(s * d.z)), d.y, fmaf(s, d.x, c * d.z));
      tally.template add<RT_EV_ROTATE_RAY>();
    }
  }
}

// The ray of group g: its chain applied to the world ray. Groups with spheres below a chain get
// the transform in FP64 (translate.clj:11, rotate.clj:18-27 on the exact FP32 world ray), split
// into an FP32 value and an FP32 residual, so the FP64 sphere test sees what the reference sees.
template <class T>
RT_DEV RayX group_ray(const DScene& S, const DGroup& G, const Ray& wr, T& tally) {
  if (!(G.flags & DGROUP_SPHERES_UNDER_CHAIN)) {
    V3 o = wr.o, d = wr.d;
    chain_apply(S, G.chain, o, d, tally);
    return make_rayx(o, d, wr.tm);
  }
  D3 o = tod(wr.o), d = tod(wr.d);
  const DChain C = S.chains[G.chain];
  for (int i = 0; i < C.n_ops; ++i) {
    const DChainOp op = S.ops[C.first_op + i];
    if (op.kind == DOP_TRANSLATE) {
      o = mkd(o.x - op.da, o.y - op.db, o.z - op.dc);
      tally.template add<RT_EV_TRANSLATE_RAY>();
    } else {
      double s = op.da, c = op.db;
      o = mkd(c * o.x - s * o.z, o.y, s * o.x + c * o.z);
      d = mkd(c * d.x - s * d.z, d.y, s * d.x + c * d.z);
      tally.template add<RT_EV_ROTATE_RAY>();
    }
  }
  RayX r = make_rayx(tof(o), tof(d), wr.tm);
  r.o_lo = mk((float)(o.x - (double)r.o.x), (float)(o.y - (double)r.o.y), (float)(o.z - (double)r.o.z));
  r.d_lo = mk((float)(d.x - (double)r.d.x), (float)(d.y - (double)r.d.y), (float)(d.z - (double)r.d.z));
  return r;
}

// ---------------------------------------------------------------------------------- BVH
RT_DEV bool slab(const float* mn, const float* mx, const RayX& r, float tmin, float tmax, float& tnear) {
  float t0x = (mn[0] - r.o.x) * r.inv.x, t1x = (mx[0] - r.o.x) * r.inv.x;
  float t0y = (mn[1] - r.o.y) * r.inv.y, t1y = (mx[1] - r.o.y) * r.inv.y;
  float t0z = (mn[2] - r.o.z) * r.inv.z, t1z = (mx[2] - r.o.z) * r.inv.z;
  // fminf/fmaxf drop a NaN operand (0 * inf when the origin lies on a slab plane)
  float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
  float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
  tnear = tn;
  return tn <= tf;
}

template <class T>
RT_DEV void bvh_traverse(const Ctx& X, int root, const RayX& r, float tmin, Best& best, int group, T& tally) {
  const DScene& S = X.S;
  int sp = 0;
  int node = root;
  for (;;) {
    const DBvhNode* N = &S.nodes[node];
    const float4 q0 = reinterpret_cast<const float4*>(N)[0];
    const float4 q1 = reinterpret_cast<const float4*>(N)[1];
    const float4 q2 = reinterpret_cast<const float4*>(N)[2];
    const float4 q3 = reinterpret_cast<const float4*>(N)[3];
    const float lmin[3] = {q0.x, q0.y, q0.z}, lmax[3] = {q0.w, q1.x, q1.y};
    const float rmin[3] = {q1.z, q1.w, q2.x}, rmax[3] = {q2.y, q2.z, q2.w};
    int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
    float tl, tr;
    tally.template add<RT_EV_AABB_TEST>(2);
    bool hl = slab(lmin, lmax, r, tmin, best.t, tl);
    bool hr = slab(rmin, rmax, r, tmin, best.t, tr);
    // leaves are tested on the spot; internal children are visited near child first
    if (hl && left < 0) { test_prim<false>(S, ~left, S.prims[~left], r, tmin, best, group, tally); hl = false; }
    if (hr && right < 0) {
      if (tr <= best.t) test_prim<false>(S, ~right, S.prims[~right], r, tmin, best, group, tally);
      hr = false;
    }
    if (hl && tl > best.t) hl = false;
    if (hr && tr > best.t) hr = false;
    if (hl && hr) {
      int near = left, far = right;
      if (tr < tl) { near = right; far = left; }
      X.lds_stack[sp * X.stack_stride] = far;
      ++sp;
      node = near;
    } else if (hl) {
      node = left;
    } else if (hr) {
      node = right;
    } else {
      if (sp == 0) break;
      --sp;
      node = X.lds_stack[sp * X.stack_stride];
    }
  }
}

// Closest hit of one group (its chain already applied to r) with t in (tmin, best.t].
template <class T>
RT_DEV void group_hit(const Ctx& X, int g, const RayX& r, float tmin, Best& best, T& tally) {
  const DScene& S = X.S;
  const DGroup G = S.groups[g];
  if (G.bvh_root >= 0) {
    bvh_traverse(X, G.bvh_root, r, tmin, best, g, tally);
  } else {
    for (int i = 0; i < G.n_prims; ++i) {
      int idx = G.first_prim + i;
      test_prim<true>(S, idx, S.prims[idx], r, tmin, best, g, tally);
    }
  }
}

// ------------------------------------------------------------------------------ records
struct Rec {  // base.clj:169
  V3 p, n;
  float t, u, v;
  int mat;
  bool front;
};

// base.clj:222-227 in FP64 (acos is ill-conditioned at the poles).
RT_DEV void sphere_uv(D3 n, float& u, float& v) {
  const double PI = 3.141592653589793;
  double theta = acos(-n.y);
  double phi = atan2(-n.z, n.x) + PI;
  u = (float)(phi / (2 * PI));
  v = (float)(theta / PI);
}

// The HitRecord of primitive `prim` for the (chain-space) ray r at parameter t.
template <class T>
RT_DEV void prim_record(const DScene& S, int prim, const RayX& r, float t, bool second_root, bool want_uv,
                        Rec& rec, T& tally) {
  const DPrim& P = S.prims[prim];
  rec.mat = P.mat;
  rec.t = t;
  rec.u = rec.v = 0.f;
  if (is_rect(P.type)) {
    // rectangles.clj:31-37; the in-plane point is o + t d, the constant coordinate is set to k
    // (in FP64 the reference's o + t d lands on k to 1e-13; FP32 rounding would not).
    tally.template add<RT_EV_RECT_RECORD>();
    V3 p = mk(fmaf(t, r.d.x, r.o.x), fmaf(t, r.d.y, r.o.y), fmaf(t, r.d.z, r.o.z));
    float a, b, dn;
    V3 out;
    if (P.type == RT_PRIM_XY_RECT) { p.z = P.f[4]; a = p.x; b = p.y; dn = r.d.z; out = mk(0, 0, 1); }
    else if (P.type == RT_PRIM_XZ_RECT) { p.y = P.f[4]; a = p.x; b = p.z; dn = r.d.y; out = mk(0, 1, 0); }
    else { p.x = P.f[4]; a = p.y; b = p.z; dn = r.d.x; out = mk(1, 0, 0); }
    rec.u = (a - P.f[0]) / (P.f[1] - P.f[0]);
    rec.v = (b - P.f[2]) / (P.f[3] - P.f[2]);
    rec.front = dn < 0.0f;  // set-face-normal, base.clj:171-174
    rec.n = rec.front ? out : neg(out);
    rec.p = p;
  } else {
    // base.clj:243-246 in FP64 from the FP64 root (recomputed: only the winner pays for it)
    tally.template add<RT_EV_SPHERE_RECORD>();
    Tally<false> quiet;
    double root1 = (double)t, root2 = (double)t;
    sphere_roots(P, r, root1, root2, quiet);
    double root = second_root ? root2 : root1;
    D3 c = sphere_center(P, r.tm);
    double inv_r = 1.0 / sphere_radius(P);
    D3 o64 = rayx_o64(r), d64 = rayx_d64(r);
    D3 pd = mkd(o64.x + root * d64.x, o64.y + root * d64.y, o64.z + root * d64.z);
    D3 out = mkd((pd.x - c.x) * inv_r, (pd.y - c.y) * inv_r, (pd.z - c.z) * inv_r);
    if (want_uv) sphere_uv(out, rec.u, rec.v);
    rec.front = ddot(d64, out) < 0.0;
    rec.n = rec.front ? tof(out) : neg(tof(out));
    rec.p = tof(pd);
  }
}

// Record fix-ups on the way back up a chain: rotate.clj:31-47 (both components use +sin, and the
// face test is against the ROTATED ray) then translate.clj:13-17 (face test against the moved
// ray, whose direction is the outer one).
template <class T>
RT_DEV void chain_unwind(const DScene& S, int chain, V3 world_d, Rec& rec, T& tally) {
  const DChain C = S.chains[chain];
  if (C.n_ops == 0) return;
  V3 dirs[RT_MAX_CHAIN_OPS + 1];
  dirs[0] = world_d;
#pragma unroll
  for (int i = 0; i < RT_MAX_CHAIN_OPS; ++i) {
    V3 d = dirs[i];
    if (i < C.n_ops) {
      const DChainOp op = S.ops[C.first_op + i];
      if (op.kind == DOP_ROTATE_Y) d = mk(op.b * d.x - op.a * d.z, d.y, op.a * d.x + op.b * d.z);
    }
    dirs[i + 1] = d;
  }
#pragma unroll
  for (int i = RT_MAX_CHAIN_OPS - 1; i >= 0; --i) {
    if (i < C.n_ops) {
      const DChainOp op = S.ops[C.first_op + i];
      V3 d_inner = dirs[i + 1];
      if (op.kind == DOP_ROTATE_Y) {
        float s = op.a, c = op.b;
        V3 p = rec.p, n = rec.n;
        rec.p = mk(c * p.x + s * p.z, p.y, s * p.x + c * p.z);
        V3 nn = mk(c * n.x + s * n.z, n.y, s * n.x + c * n.z);
        rec.front = dot(d_inner, nn) < 0.0f;
        rec.n = rec.front ? nn : neg(nn);
        tally.template add<RT_EV_ROTATE_REC>();
      } else {
        rec.front = dot(d_inner, rec.n) < 0.0f;
        rec.n = rec.front ? rec.n : neg(rec.n);
        rec.p = mk(rec.p.x + op.a, rec.p.y + op.b, rec.p.z + op.c);
        tally.template add<RT_EV_TRANSLATE_REC>();
      }
    }
  }
}

// -------------------------------------------------------------------------------- media
// constant_medium.clj:11-38 for medium m; xi is the uniform of line 32. Returns the hit
// parameter or a negative value for "no hit"; t in [t1, t2] within (tmin, tmax].
template <class T>
RT_DEV bool medium_t(const Ctx& X, int m, const Ray& wr, float tmin, float tmax, float xi, bool have_xi,
                     const Rng& g, uint32_t bounce, float& t_out, T& tally) {
  const DScene& S = X.S;
  const DMedium M = S.media[m];
  const float INF = __builtin_inff();
  // rec1 = hit boundary (-inf, +inf); rec2 = hit boundary (rec1.t + 0.0001, +inf)
  Best b1{INF, -1, -1};
  for (int gi = 0; gi < M.n_groups; ++gi) {
    int gidx = M.first_group + gi;
    group_hit(X, gidx, group_ray(S, S.groups[gidx], wr, tally), -INF, b1, tally);
  }
  if (b1.prim < 0) return false;
  Best b2{INF, -1, -1};
  for (int gi = 0; gi < M.n_groups; ++gi) {
    int gidx = M.first_group + gi;
    group_hit(X, gidx, group_ray(S, S.groups[gidx], wr, tally), b1.t + 0.0001f, b2, tally);
  }
  if (b2.prim < 0) return false;
  float t1 = (b1.t < tmin) ? tmin : b1.t;
  float t2 = (b2.t > tmax) ? tmax : b2.t;
  if (!(t1 < t2)) return false;
  if (t1 < 0.0f) t1 = 0.0f;
  if (!have_xi) {
    uint32_t o4[4];
    philox(g, (uint32_t)m >> 2, bounce, P_MEDIUM, o4);
    uint32_t w = (m & 3) == 0 ? o4[0] : (m & 3) == 1 ? o4[1] : (m & 3) == 2 ? o4[2] : o4[3];
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
template <class T>
RT_DEV bool closest_hit(const Ctx& X, const Ray& wr, float tmin, float tmax, const Rng& g, uint32_t bounce,
                        const float* medium_xi, bool skip_media, bool want_uv_always, Rec& rec, T& tally) {
  const DScene& S = X.S;
  Best best{tmax, -1, -1};
  for (int gi = 0; gi < S.n_surface_groups; ++gi)
    group_hit(X, gi, group_ray(S, S.groups[gi], wr, tally), tmin, best, tally);
  if (!skip_media) {
    for (int m = 0; m < S.n_media; ++m) {
      // the medium lives under its own outer chain; lengths and t are chain-invariant
      V3 o = wr.o, d = wr.d;
      chain_apply(S, S.media[m].outer_chain, o, d, tally);
      Ray mr{o, d, wr.tm};
      float tm;
      if (medium_t(X, m, mr, tmin, best.t, medium_xi ? medium_xi[m] : 0.f, medium_xi != nullptr, g, bounce,
                   tm, tally)) {
        best.t = tm; best.prim = -2 - m; best.group = -1;
      }
    }
  }
  if (best.prim == -1) return false;
  if (best.prim >= 0) {
    const bool second_root = (best.group & RT_SECOND_ROOT) != 0;
    best.group &= ~RT_SECOND_ROOT;
    int chain = S.groups[best.group].chain;
    RayX rx = group_ray(S, S.groups[best.group], wr, tally);
    bool want_uv = want_uv_always || S.mats[S.prims[best.prim].mat].needs_uv != 0;
    prim_record(S, best.prim, rx, best.t, second_root, want_uv, rec, tally);
    chain_unwind(S, chain, wr.d, rec, tally);
  } else {
    // constant_medium.clj:38-44: p = ray-at r t, normal [1 0 0], front-face true, no u/v
    const DMedium M = S.media[-2 - best.prim];
    V3 o = wr.o, d = wr.d;
    chain_apply(S, M.outer_chain, o, d, tally);
    rec.t = best.t;
    rec.p = mk(fmaf(best.t, d.x, o.x), fmaf(best.t, d.y, o.y), fmaf(best.t, d.z, o.z));
    rec.n = mk(1.f, 0.f, 0.f);
    rec.front = true;
    rec.mat = M.phase_mat;
    rec.u = rec.v = 0.f;
    chain_unwind(S, M.outer_chain, wr.d, rec, tally);
  }
  return true;
}

// ------------------------------------------------------------------------------- textures
RT_DEV float hermite(float x) { return x * x * (3.0f - 2.0f * x); }

// noise.clj:39-58 + :21-33. The cell index and fraction come from FP64 (p * 2^k is exact there),
// the rest is FP32. Both Hermite passes and the OR of the three permutation entries are the
// reference's.
template <class T>
RT_DEV float perlin_noise(const Ctx& X, int pid, D3 p, T& tally) {
  tally.template add<RT_EV_PERLIN_NOISE>();
  double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
  float u = hermite((float)(p.x - fx)), v = hermite((float)(p.y - fy)), w = hermite((float)(p.z - fz));
  int i = (int)(long long)fx, j = (int)(long long)fy, k = (int)(long long)fz;
  const uint8_t* perm;
  const float4* vec;
  if (X.lds_perlin_vec) {
    perm = X.lds_perlin_perm + pid * 768;
    vec = reinterpret_cast<const float4*>(X.lds_perlin_vec) + pid * 256;
  } else {
    perm = X.S.perlin_perm + pid * 768;
    vec = reinterpret_cast<const float4*>(X.S.perlin_vec) + pid * 256;
  }
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
        float4 c = vec[px | py | pz];
        float dp = c.x * (u - di) + c.y * (v - dj) + c.z * (w - dk);
        acc += ((bu * bv) * bw) * dp;
      }
    }
  }
  return acc;
}

// noise.clj:59-70
template <class T>
RT_DEV float perlin_turb(const Ctx& X, int pid, V3 p, int depth, T& tally) {
  float acc = 0.f, weight = 1.f;
  D3 tp = tod(p);
  for (int i = 0; i < depth; ++i) {
    acc += weight * perlin_noise(X, pid, tp, tally);
    weight *= 0.5f;
    tp = mkd(tp.x * 2.0, tp.y * 2.0, tp.z * 2.0);
  }
  return fabsf(acc);
}

// sign of sin(x) for FP64 x = 10 * coordinate: negative iff floor(x / pi) is odd.
RT_DEV bool sin_negative(double x) {
  double q = floor(x * 0.3183098861837907);
  return ((long long)q & 1LL) != 0;
}

// texture.clj: value of texture `tex` at (u, v, p).
template <class T>
RT_DEV V3 texture_value(const Ctx& X, int tex, float u, float v, V3 p, T& tally) {
  const DScene& S = X.S;
  DTex t = S.texs[tex];
  for (int depth = 0; t.type == RT_TEX_CHECKER && depth < 16; ++depth) {  // :25-30
    tally.template add<RT_EV_CHECKER>();
    bool nx = sin_negative(10.0 * (double)p.x), ny = sin_negative(10.0 * (double)p.y),
         nz = sin_negative(10.0 * (double)p.z);
    // the product of the three sines is negative iff an odd number of them is, and is +-0 (not
    // negative) when a coordinate is exactly 0
    bool negative = ((nx != ny) != nz) && p.x != 0.0f && p.y != 0.0f && p.z != 0.0f;
    t = S.texs[negative ? t.i0 : t.i1];
  }
  if (t.type == RT_TEX_SOLID) return mk(t.r, t.g, t.b);  // :15-17
  if (t.type == RT_TEX_NOISE) {                          // :41-44
    tally.template add<RT_EV_NOISE_TEX>();
    float k = 0.5f * (1.0f + sinf((float)t.i1 * p.z + 10.0f * perlin_turb(X, t.i0, p, 7, tally)));
    return mk(k, k, k);
  }
  if (t.type == RT_TEX_IMAGE) {  // :55-66
    tally.template add<RT_EV_IMAGE_TEX>();
    if (t.i0 < 0) return mk(0.f, 1.f, 1.f);
    const DImage im = S.images[t.i0];
    float uu = (0.0f > u) ? 0.0f : (u > 1.0f ? 1.0f : u);   // clamp, base.clj:307-310 (NaN stays)
    float vc = (0.0f > v) ? 0.0f : (v > 1.0f ? 1.0f : v);
    float vv = 1.0f - vc;
    float fi = uu * (float)im.w, fj = vv * (float)im.h;
    int i = (fi != fi) ? 0 : (int)fi;  // (int NaN) = 0 on the JVM
    int j = (fj != fj) ? 0 : (int)fj;
    if (i >= im.w) i = im.w - 1;
    if (j >= im.h) j = im.h - 1;
    uint32_t px = S.texels[im.first_texel + j * im.w + i];
    const float cs = 1.0f / 255.0f;
    return mk((float)(px & 255u) * cs, (float)((px >> 8) & 255u) * cs, (float)((px >> 16) & 255u) * cs);
  }
  return mk(0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------ materials
RT_DEV V3 reflect(V3 v, V3 n) { return v - n * (2.0f * dot(v, n)); }  // base.clj:199-200
RT_DEV V3 refract(V3 uv, V3 n, float eta) {                           // base.clj:202-209
  float cos_theta = fminf(dot(neg(uv), n), 1.0f);
  V3 perp = (uv + n * cos_theta) * eta;
  V3 par = n * (-sqrtf(fabsf(1.0f - len2(perp))));
  return perp + par;
}
RT_DEV float reflectance(float cosine, float ref_idx) {  // base.clj:215-220
  float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
  r0 = r0 * r0;
  float m = 1.0f - cosine;
  float m2 = m * m;
  return r0 + (1.0f - r0) * (m2 * m2 * m);
}

// (scatter mat r rec) and (emitted mat u v p) of materials.clj. Returns ok; on ok the ray is
// replaced by the scattered ray. `emitted` is only meaningful when !ok (core.clj:26-29).
template <class T>
RT_DEV bool scatter(const Ctx& X, const Rec& rec, Ray& r, const Rng& g, uint32_t bounce, V3& attenuation,
                    V3& emitted, T& tally) {
  const DMat M = X.S.mats[rec.mat];
  emitted = mk(0.f, 0.f, 0.f);
  switch (M.type) {
    case RT_MAT_LAMBERTIAN: {  // :16-23
      tally.template add<RT_EV_LAMBERT>();
      V3 dir = rec.n + random_unit_vector(g, bounce, tally);
      attenuation = texture_value(X, M.tex, rec.u, rec.v, rec.p, tally);
      r.o = rec.p; r.d = dir;
      return true;
    }
    case RT_MAT_METAL: {  // :33-39
      tally.template add<RT_EV_METAL>();
      V3 reflected = reflect(unit(r.d), rec.n);
      V3 dir = reflected + random_in_unit_sphere(g, bounce, tally) * M.x;
      attenuation = mk(M.r, M.g, M.b);
      r.o = rec.p; r.d = dir;
      return dot(dir, rec.n) > 0.0f;
    }
    case RT_MAT_DIELECTRIC: {  // :47-61
      tally.template add<RT_EV_DIELECTRIC>();
      float ir = M.r;
      float ratio = rec.front ? (1.0f / ir) : ir;
      V3 ud = unit(r.d);
      float cos_theta = fminf(dot(neg(ud), rec.n), 1.0f);
      float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
      bool reflect_it = ratio * sin_theta > 1.0f;
      if (!reflect_it) {  // `or` short-circuits: the uniform is drawn only here
        uint32_t o4[4];
        philox(g, 0, bounce, P_DIEL, o4);
        reflect_it = reflectance(cos_theta, ratio) > u24(o4[0]);
      }
      V3 dir = reflect_it ? reflect(ud, rec.n) : refract(ud, rec.n, ratio);
      attenuation = mk(1.f, 1.f, 1.f);
      r.o = rec.p; r.d = dir;
      return true;
    }
    case RT_MAT_DIFFUSE_LIGHT: {  // :69-78 (the unused scatter direction is not drawn)
      tally.template add<RT_EV_LIGHT>();
      emitted = texture_value(X, M.tex, rec.u, rec.v, rec.p, tally);
      attenuation = emitted;
      return false;
    }
    default: {  // RT_MAT_ISOTROPIC :87-90 — direction not normalised
      tally.template add<RT_EV_ISOTROPIC>();
      V3 dir = random_in_unit_sphere(g, bounce, tally);
      attenuation = texture_value(X, M.tex, rec.u, rec.v, rec.p, tally);
      r.o = rec.p; r.d = dir;
      return true;
    }
  }
}

// --------------------------------------------------------------------------------- camera
// core.clj:96-102 + ray.clj:14-26 in FP64 (once per path), rounded to the FP32 path state.
template <class T>
RT_DEV Ray camera_ray(const DCamera& cam, const DRender& R, int x, int y, const Rng& g, T& tally) {
  tally.template add<RT_EV_CAMERA_RAY>();
  uint32_t o4[4];
  philox(g, 0, 0, P_CAM, o4);
  double s = ((double)x + (double)u24(o4[0])) / (double)(R.width - 1);
  double t = ((double)y + (double)u24(o4[1])) / (double)(R.height - 1);
  double time_ = cam.time0 + (cam.time1 - cam.time0) * (double)u24(o4[2]);
  // random-in-unit-disk (base.clj:188-194): draw 0 never tested, first later draw inside the disk.
  double dx = 0.0, dy = 0.0;
  for (uint32_t k = 1; k < RT_MAX_TRIALS; ++k) {
    uint32_t d4[4];
    philox(g, k >> 1, 0, P_DISK, d4);
    float ux = u24((k & 1) ? d4[2] : d4[0]), uy = u24((k & 1) ? d4[3] : d4[1]);
    dx = -1.0 + 2.0 * (double)ux;
    dy = -1.0 + 2.0 * (double)uy;
    tally.template add<RT_EV_DISK_TRIAL>();
    if (dx * dx + dy * dy < 1.0) break;
  }
  double rx = dx * cam.lens_radius, ry = dy * cam.lens_radius;
  double off[3], org[3], dir[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    off[a] = cam.u[a] * rx + cam.v[a] * ry;
    org[a] = cam.origin[a] + off[a];
    dir[a] = (((cam.llc[a] + cam.horizontal[a] * s) + cam.vertical[a] * t) - cam.origin[a]) - off[a];
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
