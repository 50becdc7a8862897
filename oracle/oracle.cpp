// oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A CPU, FP64 restatement of the reference's path-tracing sample loop
// (MartinPuda/raytracing, Clojure), one function per reference function, quirks
// included. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may load this; the shipped library (librtmi.so) never does.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
// (test/raytracing/core_test.clj:1-3 is an empty namespace) and no JVM exists in
// the build container, so this restatement is pinned only by known-answer tests
// derived by hand from the cited source lines (tests/test_oracle_kat.py).
//
// Every function cites the reference file:line it follows (paths relative to
// src/raytracing/). Arithmetic keeps the reference's operation order where that
// changes results: `vd` multiplies by a reciprocal, `length-squared` and `dot`
// sum left to right, the sphere's second root reuses the first root, RotateY's
// record transform uses +sin in z, and so on.
//
// The one deliberate departure is the random source: the reference draws from one
// process-global java.util.Random in evaluation order, which no second
// implementation can follow. Draws here go through an Rng interface with two
// back ends: Scripted (a caller-supplied sequence consumed in the reference's
// evaluation order, for known-answer tests) and Keyed (counter-based
// Philox-4x32-7 addressed by purpose, so the value of a draw does not depend on
// evaluation order and the device kernel can reproduce it).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../include/rtmi.h"

namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();
constexpr double kPi = 3.141592653589793;  // clojure.math/PI = Math/PI

// ---------------------------------------------------------------- base.clj:11-46
struct V3 {
  double x, y, z;
};
inline double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }  // base.clj:11-13
inline V3 vmul(V3 v, double k) { return {v.x * k, v.y * k, v.z * k}; }         // v*  :15-18
inline V3 vadd(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }       // v+  :20-23
inline V3 vsub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }       // v-  :25-28
inline V3 vd(V3 v, double k) { return vmul(v, 1.0 / k); }                      // vd  :30-31
inline double length_squared(V3 v) { return (v.x * v.x + v.y * v.y) + v.z * v.z; }  // :33-34
inline double length(V3 v) { return std::sqrt(length_squared(v)); }                 // :36-37
inline V3 unit_vector(V3 v) { return vd(v, length(v)); }                            // :39-40
inline V3 cross(V3 u, V3 v) {                                                       // :42-46
  return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}
inline V3 vneg(V3 v) { return {-v.x, -v.y, -v.z}; }  // (map - v)
inline double clampd(double x, double mn, double mx) {  // base.clj:307-310 (NaN falls through)
  if (mn > x) return mn;
  if (x > mx) return mx;
  return x;
}

struct Ray {  // ray.clj:8
  V3 orig, dir;
  double time_;
};
inline V3 ray_at(const Ray& r, double t) { return vadd(r.orig, vmul(r.dir, t)); }  // base.clj:166-167

struct HitRecord {  // base.clj:169
  V3 p, normal;
  int mat_ptr;
  double t, u, v;
  bool front_face;
  bool has_uv;  // ConstantMedium records carry no :u/:v (constant_medium.clj:39-44)
};

// ------------------------------------------------------------------ random source
// Philox-4x32 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11). The keyed
// stream of this project uses 7 rounds (kPhiloxRounds): the fewest for which the paper reports
// the generator Crush-resistant (its default of 10 adds safety margin the renderer does not need).
constexpr int kPhiloxRounds = 7;
inline void philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], int rounds = kPhiloxRounds) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < rounds; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
inline double u24(uint32_t x) { return (double)(x >> 8) * (1.0 / 16777216.0); }

enum Purpose { P_CAM = 0, P_DISK = 1, P_SPHERE = 2, P_DIEL = 3, P_MEDIUM = 4 };
constexpr int kMaxTrials = 256;  // rejection loops give up here (probability < 1e-70)

// A draw is named by what the reference uses it for; `bounce` is the index of the
// ray-color invocation (0 = the camera ray's own invocation).
struct Rng {
  // Scripted back end: reference evaluation order.
  const double* script = nullptr;
  int script_n = 0, script_pos = 0;
  bool script_overrun = false;
  // Keyed back end: a draw is Philox(counter = (pixel index, sample index, block, bounce << 4 | purpose),
  // key = (seed_lo, seed_hi)). `key` holds what names the path, the pixel and the sample index.
  uint32_t key[2] = {0, 0};
  uint32_t seed_lo = 0, seed_hi = 0;
  // Scripted per-medium uniforms (the rt_hit_batch parity hook); NULL = not used.
  const float* medium_xi = nullptr;

  bool scripted() const { return script != nullptr; }
  double next_script() {
    if (script_pos >= script_n) {
      script_overrun = true;
      return 0.5;
    }
    return script[script_pos++];
  }
  void block(int purpose, int bounce, uint32_t blk, uint32_t out[4]) const {
    uint32_t ctr[4] = {key[0], key[1], blk, ((uint32_t)bounce << 4) | (uint32_t)purpose};
    uint32_t seed[2] = {seed_lo, seed_hi};
    philox4x32(ctr, seed, out);
  }
  // (rand) (rand) at core.clj:99,101, then (random-double time0 time1)'s uniform at ray.clj:26.
  void camera(double& jx, double& jy) {
    if (scripted()) {
      jx = next_script();
      jy = next_script();
      return;
    }
    uint32_t o[4];
    block(P_CAM, 0, 0, o);
    jx = u24(o[0]);
    jy = u24(o[1]);
  }
  double camera_time() {
    if (scripted()) return next_script();
    uint32_t o[4];
    block(P_CAM, 0, 0, o);
    return u24(o[2]);
  }
  // The k-th element of (repeatedly #(vector (random-double -1 1) (random-double -1 1) 0)).
  void disk_trial(int k, double& x, double& y) {
    double ux, uy;
    if (scripted()) {
      ux = next_script();
      uy = next_script();
    } else {
      uint32_t o[4];
      block(P_DISK, 0, (uint32_t)(k >> 1), o);
      ux = u24(o[(k & 1) * 2]);
      uy = u24(o[(k & 1) * 2 + 1]);
    }
    x = -1.0 + (1.0 - -1.0) * ux;  // random-double mn mx, base.clj:128
    y = -1.0 + (1.0 - -1.0) * uy;
  }
  // The k-th element of (repeatedly #(random-vec3 -1 1)) at invocation `bounce`. Keyed stream:
  // draw 0 is never tested by the reference's reduce, so it is not addressed; draw k >= 1 is
  // candidate j = k - 1, two candidates per Philox block (block j >> 1, half j & 1), three 16-bit
  // uniforms each: x, y = the upper and the lower half of the pair's first word, z = the upper half of its second.
  V3 sphere_trial(int bounce, int k) {
    double ux, uy, uz;
    if (scripted()) {
      ux = next_script();
      uy = next_script();
      uz = next_script();
    } else {
      uint32_t o[4];
      int j = k - 1;
      block(P_SPHERE, bounce, (uint32_t)(j >> 1), o);
      uint32_t wa = o[(j & 1) * 2], wb = o[(j & 1) * 2 + 1];
      const double S = 1.0 / 65536.0;  // 2^-16; the uniform is the centre of its cell, (k + 1/2) / 2^16
      ux = ((double)(wa >> 16) + 0.5) * S;
      uy = ((double)(wa & 0xFFFFu) + 0.5) * S;
      uz = ((double)(wb >> 16) + 0.5) * S;
    }
    return {-1.0 + 2.0 * ux, -1.0 + 2.0 * uy, -1.0 + 2.0 * uz};
  }
  double dielectric(int bounce) {  // (rand) at materials.clj:56
    if (scripted()) return next_script();
    uint32_t o[4];
    block(P_DIEL, bounce, 0, o);
    return u24(o[0]);
  }
  double medium(int bounce, int m) {  // (random-double) at constant_medium.clj:32
    if (medium_xi) return (double)medium_xi[m];
    if (scripted()) return next_script();
    uint32_t o[4];
    block(P_MEDIUM, bounce, (uint32_t)(m >> 2), o);
    return u24(o[m & 3]);
  }
};

// base.clj:182-186. (reduce f coll) without an init value seeds the accumulator with the
// first element and never tests it; the predicate keeps the first LATER draw whose squared
// length is >= 1 (the book rejects those).
V3 random_in_unit_sphere(Rng& rng, int bounce) {
  V3 v{0.0, 0.0, 0.0};
  if (rng.scripted()) v = rng.sphere_trial(bounce, 0);  // drawn, never tested; keyed draws are
                                                        // addressed, so skipping it changes nothing
  for (int k = 1; k < kMaxTrials; ++k) {
    v = rng.sphere_trial(bounce, k);
    if (length_squared(v) >= 1.0) return v;
  }
  return v;
}
// base.clj:188-194: same shape, predicate (< |v|^2 1), z = 0.
V3 random_in_unit_disk(Rng& rng) {
  double x = 0.0, y = 0.0;
  if (rng.scripted()) rng.disk_trial(0, x, y);  // drawn, never tested
  for (int k = 1; k < kMaxTrials; ++k) {
    rng.disk_trial(k, x, y);
    V3 v{x, y, 0.0};
    if (length_squared(v) < 1.0) return v;
  }
  return {x, y, 0.0};
}
V3 random_unit_vector(Rng& rng, int bounce) {  // base.clj:196-197
  return unit_vector(random_in_unit_sphere(rng, bounce));
}

// base.clj:199-200
V3 reflect(V3 v, V3 n) { return vsub(v, vmul(n, 2.0 * dot(v, n))); }
// base.clj:202-209
V3 refract(V3 uv, V3 n, double etai_over_etat) {
  double cos_theta = std::fmin(dot(vneg(uv), n), 1.0);
  V3 r_out_perp = vmul(vadd(uv, vmul(n, cos_theta)), etai_over_etat);
  V3 r_out_parallel = vmul(n, -std::sqrt(std::fabs(1.0 - length_squared(r_out_perp))));
  return vadd(r_out_perp, r_out_parallel);
}
// base.clj:215-220
double reflectance(double cosine, double ref_idx) {
  double r0 = std::pow((1.0 - ref_idx) / (1.0 + ref_idx), 2);
  return r0 + (1.0 - r0) * std::pow(1.0 - cosine, 5);
}

// ------------------------------------------------------------------------- scene
struct Scene {
  const rt_scene_desc* d;
};

// base.clj:171-174
inline void set_face_normal(const Ray& r, V3 outward, bool& front, V3& normal) {
  front = dot(r.dir, outward) < 0.0;
  normal = front ? outward : vneg(outward);
}
// base.clj:222-227
inline void get_sphere_uv(V3 p, double& u, double& v) {
  double theta = std::acos(-p.y);
  double phi = std::atan2(-p.z, p.x) + kPi;
  u = phi / (2 * kPi);
  v = theta / kPi;
}

// base.clj:234-252 (Sphere) and :274-292 (MovingSphere): identical but for the centre.
bool sphere_hit(V3 center, double radius, int mat, const Ray& r, double t_min, double t_max,
                HitRecord& rec) {
  V3 oc = vsub(r.orig, center);
  double a = length_squared(r.dir);
  double half_b = dot(oc, r.dir);
  double c = length_squared(oc) - std::pow(radius, 2);
  double discriminant = std::pow(half_b, 2) - a * c;
  if (!(discriminant > 0.0)) return false;  // (pos? discriminant)
  double sqrtd = std::sqrt(discriminant);
  double root = (-half_b - sqrtd) / a;
  if (!(t_max > root && root > t_min)) {
    root = (-half_b + root) / a;  // :247 — adds the FIRST ROOT, not sqrtd
    if (!(t_max > root && root > t_min)) return false;
  }
  V3 p = ray_at(r, root);
  V3 outward = vd(vsub(p, center), radius);
  get_sphere_uv(outward, rec.u, rec.v);
  set_face_normal(r, outward, rec.front_face, rec.normal);
  rec.p = p;
  rec.mat_ptr = mat;
  rec.t = root;
  rec.has_uv = true;
  return true;
}

// base.clj:298-302
V3 moving_center(const double* pd, double time_) {
  V3 c0{pd[0], pd[1], pd[2]}, c1{pd[3], pd[4], pd[5]};
  double time0 = pd[6], time1 = pd[7];
  return vadd(c0, vmul(vsub(c1, c0), (time_ - time0) / (time1 - time0)));
}

// rectangles.clj:15-37 / :52-74 / :89-111. `axis` is the constant axis (2, 1, 0);
// (a, b) are the in-plane axes in the record's field order.
bool rect_hit(int axis, const double* pd, int mat, const Ray& r, double t_min, double t_max,
              HitRecord& rec) {
  const double o[3] = {r.orig.x, r.orig.y, r.orig.z};
  const double dd[3] = {r.dir.x, r.dir.y, r.dir.z};
  int ia, ib;
  V3 outward;
  if (axis == 2) { ia = 0; ib = 1; outward = {0.0, 0.0, 1.0}; }
  else if (axis == 1) { ia = 0; ib = 2; outward = {0.0, 1.0, 0.0}; }
  else { ia = 1; ib = 2; outward = {1.0, 0.0, 0.0}; }
  double a0 = pd[0], a1 = pd[1], b0 = pd[2], b1 = pd[3], k = pd[4];
  double t = (k - o[axis]) / dd[axis];
  if (t < t_min || t > t_max) return false;
  double a = o[ia] + t * dd[ia];
  double b = o[ib] + t * dd[ib];
  if (a < a0 || a > a1 || b < b0 || b > b1) return false;
  rec.u = (a - a0) / (a1 - a0);
  rec.v = (b - b0) / (b1 - b0);
  rec.has_uv = true;
  set_face_normal(r, outward, rec.front_face, rec.normal);
  rec.p = ray_at(r, t);
  rec.mat_ptr = mat;
  rec.t = t;
  return true;
}

// base.clj:59-75
bool aabb_hit(const double* mn, const double* mx, const Ray& r, double t_min, double t_max) {
  const double o[3] = {r.orig.x, r.orig.y, r.orig.z};
  const double dd[3] = {r.dir.x, r.dir.y, r.dir.z};
  for (int a = 0; a < 3; ++a) {
    double invD = 1.0 / dd[a];
    double t0 = invD * (mn[a] - o[a]);
    double t1 = invD * (mx[a] - o[a]);
    double temp = t1;
    t1 = (invD < 0.0) ? t0 : t1;
    t0 = (invD < 0.0) ? temp : t0;
    t_min = (t0 > t_min) ? t0 : t_min;
    t_max = (t1 < t_max) ? t1 : t_max;
    if (t_max <= t_min) return false;
  }
  return true;
}

struct Counters {
  uint64_t rays = 0;
  std::vector<Ray>* trace = nullptr;  // when set, every ray handed to `hit` is recorded
};

bool hit_node(const Scene& s, int node, const Ray& r, double t_min, double t_max, HitRecord& rec,
              Rng& rng, int bounce);

bool hit_prim(const Scene& s, int prim, const Ray& r, double t_min, double t_max, HitRecord& rec) {
  const rt_scene_desc* d = s.d;
  const double* pd = d->prim_d + (size_t)prim * RT_PRIM_STRIDE;
  int mat = d->prim_mat[prim];
  switch (d->prim_type[prim]) {
    case RT_PRIM_SPHERE:
      return sphere_hit({pd[0], pd[1], pd[2]}, pd[3], mat, r, t_min, t_max, rec);
    case RT_PRIM_MOVING_SPHERE:
      return sphere_hit(moving_center(pd, r.time_), pd[8], mat, r, t_min, t_max, rec);
    case RT_PRIM_XY_RECT:
      return rect_hit(2, pd, mat, r, t_min, t_max, rec);
    case RT_PRIM_XZ_RECT:
      return rect_hit(1, pd, mat, r, t_min, t_max, rec);
    case RT_PRIM_YZ_RECT:
      return rect_hit(0, pd, mat, r, t_min, t_max, rec);
  }
  return false;
}

bool hit_node(const Scene& s, int node, const Ray& r, double t_min, double t_max, HitRecord& rec,
              Rng& rng, int bounce) {
  const rt_scene_desc* d = s.d;
  const int32_t* ni = d->node_i + (size_t)node * RT_NODE_ISTRIDE;
  const double* nd = d->node_d + (size_t)node * RT_NODE_DSTRIDE;
  switch (d->node_kind[node]) {
    case RT_NODE_PRIM:
      return hit_prim(s, ni[0], r, t_min, t_max, rec);
    case RT_NODE_LIST: {  // base.clj:92-101
      bool any = false;
      double closest_so_far = t_max;
      for (int i = 0; i < ni[1]; ++i) {
        HitRecord tmp;
        if (hit_node(s, d->node_children[ni[0] + i], r, t_min, closest_so_far, tmp, rng, bounce)) {
          any = true;
          closest_so_far = tmp.t;
          rec = tmp;
        }
      }
      return any;
    }
    case RT_NODE_BVH: {  // base.clj:118-122
      if (!aabb_hit(nd, nd + 3, r, t_min, t_max)) return false;
      HitRecord left, right;
      bool hit_left = hit_node(s, ni[0], r, t_min, t_max, left, rng, bounce);
      bool hit_right = hit_node(s, ni[1], r, t_min, hit_left ? left.t : t_max, right, rng, bounce);
      if (hit_right) { rec = right; return true; }  // (or hit-right hit-left)
      if (hit_left) { rec = left; return true; }
      return false;
    }
    case RT_NODE_TRANSLATE: {  // translate.clj:10-17
      V3 offset{nd[0], nd[1], nd[2]};
      Ray moved_r{vsub(r.orig, offset), r.dir, r.time_};
      if (!hit_node(s, ni[0], moved_r, t_min, t_max, rec, rng, bounce)) return false;
      set_face_normal(moved_r, rec.normal, rec.front_face, rec.normal);
      rec.p = vadd(rec.p, offset);
      return true;
    }
    case RT_NODE_ROTATE_Y: {  // rotate.clj:11-47
      double sin_theta = nd[0], cos_theta = nd[1];
      V3 o = r.orig, dd = r.dir;
      V3 new_origin{cos_theta * o.x - sin_theta * o.z, o.y, sin_theta * o.x + cos_theta * o.z};
      V3 new_direction{cos_theta * dd.x - sin_theta * dd.z, dd.y,
                       sin_theta * dd.x + cos_theta * dd.z};
      Ray rotated_r{new_origin, new_direction, r.time_};
      if (!hit_node(s, ni[0], rotated_r, t_min, t_max, rec, rng, bounce)) return false;
      V3 p = rec.p, n = rec.normal;
      // :32-42 — both components use +sin (the inverse rotation would need -sin in z)
      V3 new_p{cos_theta * p.x + sin_theta * p.z, p.y, sin_theta * p.x + cos_theta * p.z};
      V3 new_normal{cos_theta * n.x + sin_theta * n.z, n.y, sin_theta * n.x + cos_theta * n.z};
      set_face_normal(rotated_r, new_normal, rec.front_face, rec.normal);
      rec.p = new_p;
      return true;
    }
    case RT_NODE_MEDIUM: {  // constant_medium.clj:11-44
      HitRecord rec1, rec2;
      if (!hit_node(s, ni[0], r, -kInf, kInf, rec1, rng, bounce)) return false;
      if (!hit_node(s, ni[0], r, rec1.t + 0.0001, kInf, rec2, rng, bounce)) return false;
      double t1 = (rec1.t < t_min) ? t_min : rec1.t;
      double t2 = (rec2.t > t_max) ? t_max : rec2.t;
      if (!(t1 < t2)) return false;
      if (t1 < 0.0) t1 = 0.0;
      double ray_length = length(r.dir);
      double distance_inside_boundary = (t2 - t1) * ray_length;
      double hit_distance = nd[0] * std::log(rng.medium(bounce, ni[2]));
      if (!(hit_distance <= distance_inside_boundary)) return false;
      double t = t1 + hit_distance / ray_length;
      rec.t = t;
      rec.p = ray_at(r, t);
      rec.normal = {1.0, 0.0, 0.0};
      rec.front_face = true;
      rec.mat_ptr = ni[1];
      rec.u = rec.v = 0.0;
      rec.has_uv = false;
      return true;
    }
  }
  return false;
}

// ---------------------------------------------------------------------- noise.clj
inline double hermite(double x) { return x * x * (3.0 - 2.0 * x); }

// noise.clj:21-33
double perlin_interp(const V3 c[2][2][2], double u, double v, double w) {
  double uu = hermite(u), vv = hermite(v), ww = hermite(w);
  double acc = 0.0;
  bool first = true;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int k = 0; k < 2; ++k) {
        V3 weight_v{u - i, v - j, w - k};
        double term = ((i * uu + (1 - i) * (1 - uu)) * (j * vv + (1 - j) * (1 - vv))) *
                      (k * ww + (1 - k) * (1 - ww)) * dot(c[i][j][k], weight_v);
        acc = first ? term : acc + term;  // (reduce + seq)
        first = false;
      }
  return acc;
}

// noise.clj:39-58
double perlin_noise(const rt_scene_desc* d, int pid, V3 point) {
  const double* ranvec = d->perlin_ranvec + (size_t)pid * 256 * 3;
  const int32_t* px = d->perlin_perm + (size_t)pid * 3 * 256;
  const int32_t* py = px + 256;
  const int32_t* pz = py + 256;
  const double pt[3] = {point.x, point.y, point.z};
  double uvw[3];
  int64_t ijk[3];
  for (int a = 0; a < 3; ++a) {
    double value = pt[a] - std::floor(pt[a]);
    uvw[a] = value * value * (3 - 2.0 * value);  // first Hermite (:40-45)
    ijk[a] = (int64_t)std::floor(pt[a]);
  }
  V3 c[2][2][2];
  for (int di = 0; di < 2; ++di)
    for (int dj = 0; dj < 2; ++dj)
      for (int dk = 0; dk < 2; ++dk) {
        int idx = px[(ijk[0] + di) & 255] | py[(ijk[1] + dj) & 255] | pz[(ijk[2] + dk) & 255];  // bit-or, :53-55
        c[di][dj][dk] = {ranvec[idx * 3], ranvec[idx * 3 + 1], ranvec[idx * 3 + 2]};
      }
  return perlin_interp(c, uvw[0], uvw[1], uvw[2]);  // second Hermite inside
}

// noise.clj:59-70
double perlin_turb(const rt_scene_desc* d, int pid, V3 point, int depth) {
  double acc = 0.0, weight = 1.0;
  V3 temp_p = point;
  for (int i = 0; i < depth; ++i) {
    acc = acc + weight * perlin_noise(d, pid, temp_p);
    weight = weight * 0.5;
    temp_p = vmul(temp_p, 2);
  }
  return std::fabs(acc);
}

// -------------------------------------------------------------------- texture.clj
V3 texture_value(const rt_scene_desc* d, int tex, double u, double v, V3 p) {
  const double* td = d->tex_d + (size_t)tex * RT_TEX_DSTRIDE;
  const int32_t* ti = d->tex_i + (size_t)tex * RT_TEX_ISTRIDE;
  switch (d->tex_type[tex]) {
    case RT_TEX_SOLID:  // :15-17
      return {td[0], td[1], td[2]};
    case RT_TEX_CHECKER: {  // :25-30  (transduce (map sin(10 x)) * p) = ((1*sx)*sy)*sz
      double sines = ((1.0 * std::sin(10 * p.x)) * std::sin(10 * p.y)) * std::sin(10 * p.z);
      return texture_value(d, sines < 0.0 ? ti[0] : ti[1], u, v, p);
    }
    case RT_TEX_NOISE: {  // :41-44 (scale is a long field)
      double scale = (double)ti[1];
      double k = 0.5 * (1.0 + std::sin(scale * p.z + 10 * perlin_turb(d, ti[0], p, 7)));
      return vmul({1, 1, 1}, k);
    }
    case RT_TEX_IMAGE: {  // :55-66
      if (ti[0] < 0) return {0, 1, 1};
      int width = d->img_w[ti[0]], height = d->img_h[ti[0]];
      const uint8_t* data = d->img_rgb8[ti[0]];
      double uu = clampd(u, 0.0, 1.0);
      double vv = 1.0 - clampd(v, 0.0, 1.0);
      double fi = uu * width, fj = vv * height;
      int i = std::isnan(fi) ? 0 : (int)fi;  // (int NaN) = 0 on the JVM
      int j = std::isnan(fj) ? 0 : (int)fj;
      if (i >= width) i = width - 1;
      if (j >= height) j = height - 1;
      double color_scale = 1.0 / 255.0;
      const uint8_t* px = data + ((size_t)j * width + i) * 3;
      return vmul({(double)px[0], (double)px[1], (double)px[2]}, color_scale);
    }
  }
  return {0, 0, 0};
}

// ------------------------------------------------------------------ materials.clj
struct Scatter {
  bool ok;
  V3 attenuation;
  Ray scattered;
};

V3 material_emitted(const rt_scene_desc* d, int mat, const HitRecord& rec) {
  if (d->mat_type[mat] == RT_MAT_DIFFUSE_LIGHT)  // :77-78
    return texture_value(d, d->mat_tex[mat], rec.u, rec.v, rec.p);
  return {0.0, 0.0, 0.0};
}

Scatter material_scatter(const rt_scene_desc* d, int mat, const Ray& r_in, const HitRecord& rec,
                         Rng& rng, int bounce) {
  const double* md = d->mat_d + (size_t)mat * RT_MAT_DSTRIDE;
  Scatter s;
  switch (d->mat_type[mat]) {
    case RT_MAT_LAMBERTIAN: {  // :16-23
      V3 scatter_direction = vadd(rec.normal, random_unit_vector(rng, bounce));
      s.ok = true;
      s.scattered = {rec.p, scatter_direction, r_in.time_};
      s.attenuation = texture_value(d, d->mat_tex[mat], rec.u, rec.v, rec.p);
      return s;
    }
    case RT_MAT_METAL: {  // :33-39
      V3 reflected = reflect(unit_vector(r_in.dir), rec.normal);
      V3 dir = vadd(reflected, vmul(random_in_unit_sphere(rng, bounce), md[3]));
      s.scattered = {rec.p, dir, r_in.time_};
      s.ok = dot(dir, rec.normal) > 0.0;
      s.attenuation = {md[0], md[1], md[2]};
      return s;
    }
    case RT_MAT_DIELECTRIC: {  // :47-61
      double ir = md[0];
      double refraction_ratio = rec.front_face ? (1.0 / ir) : ir;
      V3 unit_direction = unit_vector(r_in.dir);
      double cos_theta = std::fmin(dot(vneg(unit_direction), rec.normal), 1.0);
      double sin_theta = std::sqrt(1.0 - std::pow(cos_theta, 2));
      bool cannot_refract = refraction_ratio * sin_theta > 1.0;
      V3 direction;
      if (cannot_refract || reflectance(cos_theta, refraction_ratio) > rng.dielectric(bounce))
        direction = reflect(unit_direction, rec.normal);
      else
        direction = refract(unit_direction, rec.normal, refraction_ratio);
      s.ok = true;
      s.attenuation = {1.0, 1.0, 1.0};
      s.scattered = {rec.p, direction, r_in.time_};
      return s;
    }
    case RT_MAT_DIFFUSE_LIGHT: {  // :69-76 — draws a direction it never uses
      V3 scatter_direction = vadd(rec.normal, random_unit_vector(rng, bounce));
      s.ok = false;
      s.scattered = {rec.p, scatter_direction, r_in.time_};
      s.attenuation = texture_value(d, d->mat_tex[mat], rec.u, rec.v, rec.p);
      return s;
    }
    case RT_MAT_ISOTROPIC: {  // :87-90 — direction is NOT normalised
      s.ok = true;
      s.scattered = {rec.p, random_in_unit_sphere(rng, bounce), r_in.time_};
      s.attenuation = texture_value(d, d->mat_tex[mat], rec.u, rec.v, rec.p);
      return s;
    }
  }
  s.ok = false;
  s.attenuation = {0, 0, 0};
  s.scattered = r_in;
  return s;
}

// ----------------------------------------------------------------- core.clj:19-30
V3 ray_color(const Scene& s, const Ray& r, V3 background, int depth, Rng& rng, int bounce,
             Counters& cnt) {
  if (0 >= depth) return {0.0, 0.0, 0.0};
  HitRecord rec;
  cnt.rays++;
  if (cnt.trace) cnt.trace->push_back(r);
  if (hit_node(s, s.d->root, r, 0.001, kInf, rec, rng, bounce)) {
    V3 emitted = material_emitted(s.d, rec.mat_ptr, rec);
    Scatter sc = material_scatter(s.d, rec.mat_ptr, r, rec, rng, bounce);
    if (sc.ok) {
      V3 c = ray_color(s, sc.scattered, background, depth - 1, rng, bounce + 1, cnt);
      return {sc.attenuation.x * c.x, sc.attenuation.y * c.y, sc.attenuation.z * c.z};
    }
    return emitted;
  }
  return background;
}

// ------------------------------------------------------------------- ray.clj:14-26
Ray get_ray(const rt_camera& cam, double s, double t, Rng& rng) {
  V3 u{cam.u[0], cam.u[1], cam.u[2]}, v{cam.v[0], cam.v[1], cam.v[2]};
  V3 origin{cam.origin[0], cam.origin[1], cam.origin[2]};
  V3 llc{cam.lower_left_corner[0], cam.lower_left_corner[1], cam.lower_left_corner[2]};
  V3 horizontal{cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]};
  V3 vertical{cam.vertical[0], cam.vertical[1], cam.vertical[2]};
  V3 rd = vmul(random_in_unit_disk(rng), cam.lens_radius);
  V3 offset = vadd(vmul(u, rd.x), vmul(v, rd.y));
  V3 dir = vsub(vsub(vadd(vadd(llc, vmul(horizontal, s)), vmul(vertical, t)), origin), offset);
  double time_ = cam.time0 + (cam.time1 - cam.time0) * rng.camera_time();
  return {vadd(origin, offset), dir, time_};
}

// core.clj:96-105: one sample of pixel (x, y).
V3 sample_pixel(const Scene& s, const rt_camera& cam, const rt_params& prm, int x, int y,
                uint32_t sample_index, Counters& cnt) {
  Rng rng;
  rng.key[0] = (uint32_t)(y * prm.width + x);
  rng.key[1] = sample_index;
  rng.seed_lo = (uint32_t)prm.seed;
  rng.seed_hi = (uint32_t)(prm.seed >> 32);
  double jx, jy;
  rng.camera(jx, jy);
  double su = (x + jx) / (double)(prm.width - 1);
  double sv = (y + jy) / (double)(prm.height - 1);
  Ray r = get_ray(cam, su, sv, rng);
  V3 bg{prm.background[0], prm.background[1], prm.background[2]};
  return ray_color(s, r, bg, prm.max_depth, rng, 0, cnt);
}

// core.clj:106-108,118,121-122 + java.awt.Color(float,float,float).getRGB().
uint32_t pack_pixel(double r, double g, double b, int spp) {
  double scale = 1.0 / spp;
  const double in[3] = {r, g, b};
  uint32_t out[3];
  for (int i = 0; i < 3; ++i) {
    double c = clampd(std::sqrt(in[i] * scale), 0.0, 0.999);
    float f = (float)c;                       // ^float destructuring at core.clj:108
    double scaled = (double)(f * 255.0f) + 0.5;  // Color(float…): (int)(r*255+0.5)
    int iv = std::isnan(scaled) ? 0 : (int)scaled;  // (int)NaN = 0 on the JVM
    out[i] = (uint32_t)iv & 0xFF;
  }
  return 0xFF000000u | (out[0] << 16) | (out[1] << 8) | out[2];
}

bool tile_owned(const rt_params& p, int x, int y) {
  int tiles_x = (p.width + 7) / 8;
  int t = (y / 8) * tiles_x + (x / 8);
  int sc = p.shard_count > 0 ? p.shard_count : 1;
  // the tiles are dealt in groups of sc; the deal moves on by one place with every tile row the groups start in
  int turn = ((t / sc) * sc / tiles_x) % sc;
  return (t % sc + sc - turn) % sc == p.shard_index;
}

}  // namespace

extern "C" {

// Closest hit per ray, as ray-color's `hit` call (core.clj:23) returns it. Output fields are
// doubles: t, p[3], normal[3], u, v, front_face, mat_id, hit (12 per ray).
int orc_hit_batch(const rt_scene_desc* d, int32_t n, const float* orig3, const float* dir3,
                  const float* time, double t_min, double t_max, const float* medium_xi,
                  int32_t skip_media, double* out12) {
  Scene s{d};
  // xi = 0 -> log = -inf -> hit_distance = +inf -> a medium never produces a hit
  std::vector<float> never((size_t)std::max(d->n_media, 1), 0.0f);
  for (int i = 0; i < n; ++i) {
    Ray r{{orig3[3 * i], orig3[3 * i + 1], orig3[3 * i + 2]},
          {dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]},
          time ? (double)time[i] : 0.0};
    Rng rng;
    rng.medium_xi = (skip_media || !medium_xi) ? never.data() : medium_xi + (size_t)i * d->n_media;
    HitRecord rec;
    double* o = out12 + (size_t)i * 12;
    std::memset(o, 0, 12 * sizeof(double));
    if (hit_node(s, d->root, r, t_min, t_max, rec, rng, 0)) {
      o[0] = rec.t;
      o[1] = rec.p.x; o[2] = rec.p.y; o[3] = rec.p.z;
      o[4] = rec.normal.x; o[5] = rec.normal.y; o[6] = rec.normal.z;
      o[7] = rec.has_uv ? rec.u : 0.0;
      o[8] = rec.has_uv ? rec.v : 0.0;
      o[9] = rec.front_face ? 1.0 : 0.0;
      o[10] = (double)rec.mat_ptr;
      o[11] = 1.0;
    }
  }
  return 0;
}

// The render loop of core.clj:96-127 with `nthreads` workers pulling rows (the reference
// starts one thread per row). rgb_sum: W*H*3 doubles, row 0 = y 0 (bottom).
int orc_render(const rt_scene_desc* d, const rt_camera* cam, const rt_params* prm, double* rgb_sum,
               uint32_t* argb, int32_t nthreads, uint64_t* rays_out) {
  Scene s{d};
  const int W = prm->width, H = prm->height;
  std::atomic<int> next_row{0};
  std::atomic<uint64_t> rays{0};
  auto worker = [&]() {
    Counters cnt;
    for (;;) {
      int y = next_row.fetch_add(1);
      if (y >= H) break;
      for (int x = 0; x < W; ++x) {
        V3 acc{0.0, 0.0, 0.0};
        if (tile_owned(*prm, x, y)) {
          for (int sidx = 0; sidx < prm->spp; ++sidx)
            acc = vadd(acc, sample_pixel(s, *cam, *prm, x, y, (uint32_t)(prm->sample_offset + sidx), cnt));
        }
        size_t pi = (size_t)y * W + x;
        if (rgb_sum) { rgb_sum[3 * pi] = acc.x; rgb_sum[3 * pi + 1] = acc.y; rgb_sum[3 * pi + 2] = acc.z; }
        if (argb) argb[pi] = pack_pixel(acc.x, acc.y, acc.z, prm->spp);
      }
    }
    rays += cnt.rays;
  };
  int nt = nthreads < 1 ? 1 : nthreads;
  std::vector<std::thread> th;
  for (int i = 1; i < nt; ++i) th.emplace_back(worker);
  worker();
  for (auto& t : th) t.join();
  if (rays_out) *rays_out = rays.load();
  return 0;
}

// Radiance of single (pixel, sample) units: out_rgb[3*i..] for pixel index px[i], sample smp[i].
int orc_sample_batch(const rt_scene_desc* d, const rt_camera* cam, const rt_params* prm, int32_t n,
                     const int32_t* px, const int32_t* smp, double* out_rgb) {
  Scene s{d};
  Counters cnt;
  for (int i = 0; i < n; ++i) {
    int x = px[i] % prm->width, y = px[i] / prm->width;
    V3 c = sample_pixel(s, *cam, *prm, x, y, (uint32_t)smp[i], cnt);
    out_rgb[3 * i] = c.x; out_rgb[3 * i + 1] = c.y; out_rgb[3 * i + 2] = c.z;
  }
  return 0;
}

// One sample of pixel (x, y) with SCRIPTED uniforms consumed in the reference's evaluation order (core.clj:96-105: the
// two jitters, get-ray's disk draws and its time, then ray-color's draws bounce by bounce, a medium's free-flight draw
// where the scan reaches it). out_rgb3 = the sample's radiance; returns the uniforms consumed, -1 on overrun.
int orc_sample_scripted(const rt_scene_desc* d, const rt_camera* cam, const rt_params* prm, int32_t x, int32_t y,
                        const double* script, int32_t n, double* out_rgb3) {
  Scene s{d};
  Counters cnt;
  Rng rng;
  rng.script = script;
  rng.script_n = n;
  double jx, jy;
  rng.camera(jx, jy);
  double su = (x + jx) / (double)(prm->width - 1);
  double sv = (y + jy) / (double)(prm->height - 1);
  Ray r = get_ray(*cam, su, sv, rng);
  V3 bg{prm->background[0], prm->background[1], prm->background[2]};
  V3 c = ray_color(s, r, bg, prm->max_depth, rng, 0, cnt);
  out_rgb3[0] = c.x; out_rgb3[1] = c.y; out_rgb3[2] = c.z;
  return rng.script_overrun ? -1 : rng.script_pos;
}

// The rays one (pixel, sample) path hands to `hit`, in order: out7[7*k..] = orig dir time (FP64).
// Returns the number of rays (at most max_rays are written).
int orc_path_rays(const rt_scene_desc* d, const rt_camera* cam, const rt_params* prm, int32_t px,
                  int32_t smp, int32_t max_rays, double* out7) {
  Scene s{d};
  Counters cnt;
  std::vector<Ray> rays;
  cnt.trace = &rays;
  sample_pixel(s, *cam, *prm, px % prm->width, px / prm->width, (uint32_t)smp, cnt);
  int n = (int)rays.size();
  for (int i = 0; i < n && i < max_rays; ++i) {
    double* o = out7 + (size_t)i * 7;
    o[0] = rays[i].orig.x; o[1] = rays[i].orig.y; o[2] = rays[i].orig.z;
    o[3] = rays[i].dir.x; o[4] = rays[i].dir.y; o[5] = rays[i].dir.z;
    o[6] = rays[i].time_;
  }
  return n;
}

uint32_t orc_pack_pixel(double r, double g, double b, int32_t spp) { return pack_pixel(r, g, b, spp); }

// The post-process applied to FP32 sums (what the device holds), for bit-exact comparison
// with the device tonemap kernel.
int orc_tonemap_f32(const float* rgb_sum, int32_t n_pixels, int32_t spp, uint32_t* argb) {
  for (int i = 0; i < n_pixels; ++i)
    argb[i] = pack_pixel(rgb_sum[3 * i], rgb_sum[3 * i + 1], rgb_sum[3 * i + 2], spp);
  return 0;
}

// camera.clj:7-33
int orc_camera(const double* lookfrom, const double* lookat, const double* vup, double vfov,
               double aspect_ratio, double aperture, double focus_dist, double time0, double time1,
               rt_camera* out) {
  double theta = vfov * 0.017453292519943295;  // Math/toRadians
  double h = std::tan(theta * 0.5);
  double viewport_height = h * 2.0;
  double viewport_width = aspect_ratio * viewport_height;
  V3 from{lookfrom[0], lookfrom[1], lookfrom[2]}, at{lookat[0], lookat[1], lookat[2]};
  V3 up{vup[0], vup[1], vup[2]};
  V3 w = unit_vector(vsub(from, at));
  V3 u = unit_vector(cross(up, w));
  V3 v = cross(w, u);
  V3 origin = from;
  V3 horizontal = vmul(u, focus_dist * viewport_width);
  V3 vertical = vmul(v, focus_dist * viewport_height);
  V3 llc = vsub(vsub(vsub(origin, vd(horizontal, 2)), vd(vertical, 2)), vmul(w, focus_dist));
  auto put = [](double* dst, V3 a) { dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; };
  put(out->origin, origin);
  put(out->lower_left_corner, llc);
  put(out->horizontal, horizontal);
  put(out->vertical, vertical);
  put(out->u, u);
  put(out->v, v);
  put(out->w, w);
  out->lens_radius = aperture / 2;
  out->time0 = time0;
  out->time1 = time1;
  return 0;
}

// ---- known-answer-test hooks: single reference functions with scripted draws ----
double orc_perlin_noise(const rt_scene_desc* d, int32_t pid, const double* p) {
  return perlin_noise(d, pid, {p[0], p[1], p[2]});
}
double orc_perlin_turb(const rt_scene_desc* d, int32_t pid, const double* p, int32_t depth) {
  return perlin_turb(d, pid, {p[0], p[1], p[2]}, depth);
}
int orc_texture_value(const rt_scene_desc* d, int32_t tex, double u, double v, const double* p,
                      double* out3) {
  V3 c = texture_value(d, tex, u, v, {p[0], p[1], p[2]});
  out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
  return 0;
}
// which: 0 random-in-unit-sphere, 1 random-in-unit-disk, 2 random-unit-vector. Returns the
// number of scripted uniforms consumed (reference evaluation order), or -1 on overrun.
int orc_sampler_scripted(int32_t which, const double* script, int32_t n, double* out3) {
  Rng rng;
  rng.script = script;
  rng.script_n = n;
  V3 v = which == 0 ? random_in_unit_sphere(rng, 0)
         : which == 1 ? random_in_unit_disk(rng)
                      : random_unit_vector(rng, 0);
  out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
  return rng.script_overrun ? -1 : rng.script_pos;
}
// Keyed draws, for statistical tests of the samplers: n vectors for key (i, 0).
int orc_sampler_keyed(int32_t which, uint64_t seed, int32_t n, double* out3n) {
  for (int i = 0; i < n; ++i) {
    Rng rng;
    rng.key[0] = (uint32_t)i;
    rng.seed_lo = (uint32_t)seed;
    rng.seed_hi = (uint32_t)(seed >> 32);
    V3 v = which == 0 ? random_in_unit_sphere(rng, 0)
           : which == 1 ? random_in_unit_disk(rng)
                        : random_unit_vector(rng, 0);
    out3n[3 * i] = v.x; out3n[3 * i + 1] = v.y; out3n[3 * i + 2] = v.z;
  }
  return 0;
}
// scatter + emitted of material `mat` for one ray/record with scripted draws.
// rec9: p[3] normal[3] u v front; out: ok, attenuation[3], scattered orig[3] dir[3] time, emitted[3].
int orc_scatter_scripted(const rt_scene_desc* d, int32_t mat, const double* ray7, const double* rec9,
                         const double* script, int32_t n, double* out14) {
  Rng rng;
  rng.script = script;
  rng.script_n = n;
  Ray r{{ray7[0], ray7[1], ray7[2]}, {ray7[3], ray7[4], ray7[5]}, ray7[6]};
  HitRecord rec;
  rec.p = {rec9[0], rec9[1], rec9[2]};
  rec.normal = {rec9[3], rec9[4], rec9[5]};
  rec.u = rec9[6]; rec.v = rec9[7];
  rec.front_face = rec9[8] != 0.0;
  rec.has_uv = true;
  rec.mat_ptr = mat;
  rec.t = 0;
  V3 em = material_emitted(d, mat, rec);
  Scatter s = material_scatter(d, mat, r, rec, rng, 0);
  out14[0] = s.ok ? 1.0 : 0.0;
  out14[1] = s.attenuation.x; out14[2] = s.attenuation.y; out14[3] = s.attenuation.z;
  out14[4] = s.scattered.orig.x; out14[5] = s.scattered.orig.y; out14[6] = s.scattered.orig.z;
  out14[7] = s.scattered.dir.x; out14[8] = s.scattered.dir.y; out14[9] = s.scattered.dir.z;
  out14[10] = s.scattered.time_;
  out14[11] = em.x; out14[12] = em.y; out14[13] = em.z;
  return rng.script_overrun ? -1 : rng.script_pos;
}
void orc_philox(const uint32_t* ctr4, const uint32_t* key2, int rounds, uint32_t* out4) {
  philox4x32(ctr4, key2, out4, rounds > 0 ? rounds : kPhiloxRounds);
}
double orc_reflectance(double cosine, double ref_idx) { return reflectance(cosine, ref_idx); }
int orc_refract(const double* uv, const double* n, double eta, double* out3) {
  V3 r = refract({uv[0], uv[1], uv[2]}, {n[0], n[1], n[2]}, eta);
  out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
  return 0;
}
int orc_aabb_hit(const double* mn, const double* mx, const double* ray6, double t_min, double t_max) {
  Ray r{{ray6[0], ray6[1], ray6[2]}, {ray6[3], ray6[4], ray6[5]}, 0.0};
  return aabb_hit(mn, mx, r, t_min, t_max) ? 1 : 0;
}

}  // extern "C"
