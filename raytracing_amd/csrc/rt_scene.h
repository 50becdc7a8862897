// rt_scene.h — device-side scene layout shared by the host uploader and the kernels.
//
// The flat description of include/rtmi.h keeps the reference's hittable tree. The uploader
// rewrites it into the form the kernels traverse:
//   * every Translate / RotateY wrapper path from the root becomes an interned "chain" of
//     operations (outermost first). The record transform of RotateY is not the inverse of its
//     ray transform (reference rotate.clj:32-42), so chains are kept as operation lists and
//     never collapsed into a matrix.
//   * all leaves reachable under one chain form a "group": a contiguous run of primitive
//     records in the reference's scan order (so the index doubles as the tie-break rank),
//     intersected by a wave-uniform brute-force loop when small and by a GPU-built BVH
//     otherwise.
//   * every ConstantMedium becomes a medium record pointing at the groups of its boundary.
#ifndef RT_SCENE_H
#define RT_SCENE_H

#include <stdint.h>

#include "../../include/rtmi.h"

#define RT_MAX_CHAIN_OPS 4     // deepest Translate/RotateY nesting supported
#define RT_BRUTE_MAX 16        // groups up to this size skip the BVH
#define RT_MAX_PERLIN_LDS 8    // Perlin table sets staged in LDS
#define RT_MAX_CHUNKS 128      // work items per pixel, at most
#define RT_TILE 8              // pixels per tile edge; one wave64 = one 8x8 tile
#define RT_QUEUE_HEADS 8        // the work-item queue has this many heads (one window of every 8 each)
#define RT_QUEUE_STRIDE 32      // dwords between two heads (each on a 128-byte line of its own)
#define RT_PAIR_STRIDE 12      // dwords per pair record of DScene.brute_rects
#define RT_BOX_STRIDE 12       // dwords per box record of DScene.brute_boxes
#ifndef RT_PAIR_BATCH
#define RT_PAIR_BATCH 1        // pair records the rect kernels' brute-force loop fetches per scalar-memory round trip (1 to 4;
                               // since rooms and boxes became slab tests few rects are left loose: 1 measured 0.8 % faster than 2)
#endif
#define RT_PAIR_SLACK 4        // records of slack after the table: the most a batch reads past its end

// The statistics buffer of a render: samples, rays, the event counters, then four clock words
// (cycle counter at the start and end of one wave, constant-rate timer at the same two moments).
#define RT_STATS_CLOCK (2 + RT_EV_COUNT)
#define RT_STATS_WORDS (2 + RT_EV_COUNT + 4)

enum { DOP_TRANSLATE = 0, DOP_ROTATE_Y = 1 };

// Scene features a kernel instantiation must support (DScene.features, Ctx<FEAT>).
enum : uint32_t {
  F_SPHERES = 1,    // sphere / moving-sphere primitives (FP64 quadratic)
  F_BVH = 2,        // at least one group is traversed through a BVH
  F_MEDIA = 4,      // constant media (and their isotropic phase functions)
  F_CHAINS = 8,     // translate / rotate-y wrappers
  F_TEXTURES = 16,  // non-solid textures: checker (noise and image only with F_NOISE / F_IMAGE)
  F_SPECULAR = 32,  // metal / dielectric
  F_SMALL = 64,     // not a scene feature: the instantiation keeps DPrimInfo in LDS (<= RT_LDS_PRIMS)
  F_NOISE = 128,    // Perlin noise textures: tables in LDS, by far the most registers
  F_IMAGE = 256,    // image textures: the only consumers of u/v (FP64 acos/atan2 for spheres)
  F_LIDBOX = 2048,  // some tree holds RT_PRIM_LIDBOX leaves (only the all-features instantiations know them: the
                    // leaf test's extra branch costs the sphere / BVH kernel 11 % on the random-spheres world)
  F_RECTS = 4096,   // rect primitives (every instantiation but the one for worlds of spheres alone: without the rect
                    // tests and records the sphere / BVH kernel spills nothing and is 8 % faster on the random-spheres world)
  F_MOVING = 8192,  // moving spheres (every instantiation but the one for worlds of static spheres alone)
  F_ALL = 63 | 128 | 256 | 2048 | 4096 | 8192,
  F_DEEP = 512,     // not a scene feature: same code, compiled for more waves per SIMD (BVH worlds)
  F_MEDIA_ANY = 1024  // some medium's boundary is neither one sphere nor one closed box: its two boundary
                      // queries run the general group loops (30 more registers in every kernel that has them)
};

// 80 bytes. Spheres are intersected in FP64 (|oc|^2 - r^2 cancels catastrophically in FP32 for
// the r = 1000 ground spheres), rectangles in FP32.
struct DPrim {
  int32_t type;  // RT_PRIM_* in bits 0-3, index of the owning group in bits 4-31
  int32_t mat;   // material index of the description
  union {
    double d[9];  // sphere: cx cy cz r, then 1 / r | moving sphere: c0 xyz, c1 xyz, t0, t1, r
    float f[18];  // rect: a0 a1 b0 b1 k
  };
};

// A tree leaf that stands for the six rects of one box whose lid is sunk below its rim: box.clj:16-23 builds a box
// as six rects and the ^long field k of rectangles.clj:9,46,83 drops the plane of every rect to a whole number, so a
// box standing on y0 = 0 with a top at y1 = 37.4 has walls that reach 37.4 and a lid at y = 37 (the 400 ground boxes
// of final-scene, scenes.clj:110-120). The walls and the floor are an axis-aligned box with the y1 face missing, the
// lid is one XZ rect inside it. Internal to the device form (never in a description): DPrim.f = x0 x1 y0 y1 z0 z1,
// lid k, then as int bits the device primitives of the faces z0 z1 y0 x0 x1 and of the lid. The six rects keep
// their records behind the group's tested primitives; the leaf test only names the one that was hit.
#define RT_PRIM_LIDBOX 5

struct DChainOp {  // one wrapper of a chain, as the flattener keeps it on the host
  int32_t kind;
  float a, b, c;     // TRANSLATE: offset xyz | ROTATE_Y: sin, cos
  double da, db, dc; // the same in FP64, for groups whose spheres are intersected in FP64
};

// 64 bytes: the record side of a whole chain. On the way back up, every wrapper maps the hit
// point (rotate.clj:32-36: x' = c x + s z, z' = s x + c z, both with +sin; translate.clj:16:
// p + offset) and re-orients the normal against its own ray (rotate.clj:43, translate.clj:13).
// The point maps compose into one, p' = [[A B] [B A]] (x z) + t. A wrapper's normal is
// +-(its map)(inner normal) with the sign picked by ITS face test, whatever the wrappers inside
// decided; front-face says whether that sign kept the inner wrapper's choice. So the outcome needs
// two face tests only: the second-outermost wrapper's (map [[A1 B1] [B1 A1]] = all wrappers but
// the outermost, against the world ray turned by (s01, c01) = the ray inside that wrapper), then
// the outermost's (its own record map (s0, c0), against the world ray turned by (s0, c0); a
// translate keeps the direction: (0, 1)). With a single wrapper the primitive's own orientation
// stands in for the inner test (inner = 0).
// Staged in LDS for small scenes.
struct DChainRec {
  int32_t n_ops;      // 0: identity chain, nothing to do
  int32_t n_rotate;   // how many of the n_ops are rotate-y (event accounting)
  int32_t inner;      // 1 when n_ops >= 2
  int32_t pad0;
  float A, B, tx, ty;
  float tz, s0, c0, A1;
  float B1, s01, c01, pad1;
};
#define RT_LDS_CHAINS 64  // F_SMALL instantiations keep up to this many chain records in LDS

// 64 bytes, read with one scalar load by the brute-force loop. Primitives of a group are stored
// by type (XY rects, XZ rects, YZ rects, spheres), stably, so each type gets a branch-free loop;
// scan order within a type is kept, which preserves the reference's outcome for every tie
// between coincident primitives of one type.
struct DGroup {
  int32_t first_prim;
  int32_t n_xy, n_xz, n_yz, n_sph;
  int32_t bvh_root;  // node index, or -1 for the brute-force loops
  int32_t flags;     // DGROUP_*
  int32_t chain;
  int32_t first_rect;  // first pair record of this group in DScene.brute_rects (brute-force groups)
  // The chain's ray transform collapsed to one rigid motion about y (translate.clj:11 and
  // rotate.clj:18-27 compose exactly): o' = Ry(o) + b, d' = Ry(d), Ry(v) = [c vx - s vz, vy, s vx + c vz].
  // Only the ray side may be collapsed; records go back up the chain operation by operation.
  float rc, rs, bx, by, bz;
  // Brute-force groups: six rects that close an axis-aligned box (box.clj:7-20) are intersected as
  // ONE slab test (DScene.brute_boxes) and left out of the pair records. `loose` packs what the
  // pair records still hold and the number of boxes: xy | xz << 8 | yz << 16 | boxes << 24.
  int32_t loose;
  int32_t first_box;
  int32_t n_lidbox;  // groups with a tree: RT_PRIM_LIDBOX leaves, stored behind the group's spheres
  int32_t pad_[3];
};
struct DGroup64 {  // the same motion in FP64, for DGROUP_SPHERES_UNDER_CHAIN groups
  double rc, rs, bx, by, bz;
  double pad[3];
};
#define DGROUP_OPEN_SHIFT 8            // flags bits 8-15: boxes with one face missing, stored after the
                                      // closed ones in DScene.brute_boxes (the missing face's index is -1)
#define DGROUP_SPHERES_UNDER_CHAIN 1  // spheres below a non-identity chain: the ray is carried into
                                      // chain space in FP64 (as a float hi + lo pair)

// 160 bytes: everything the winner of a traversal needs for its record and its scatter, gathered
// so that the divergent (per-lane) lookup after the loops is ONE memory round trip. Staged in
// LDS for small scenes.
struct DPrimInfo {
  int32_t type_group, mat, chain, gflags;  // as DPrim.type / DPrim.mat / DGroup.chain / DGroup.flags
  float a0, a1, b0, b1;                    // rect bounds (for u, v)
  float k, rc, rs, bx;                     // rect plane; the group's ray motion (DGroup)
  float by, bz;
  int32_t mtype, mflags;                   // DMat.type / DMat.flags of the material
  int32_t mtex;
  float r, g, b;                           // DMat.r g b
  float x;                                 // DMat.x
  float ox, oy, oz;                        // rects: the unit vector of the constant axis (outward normal)
  // Rects below a chain: the chain's record side (DChainRec) applied to THIS rect, whose normal is an axis. The
  // point map (fA fB ftx fty ftz = the chain's A B tx ty tz); the normal every wrapper can only flip, carried through
  // all of them (fn) and through all but the outermost (its image under the ray turns: g1); the outermost's face
  // test as a dot product with the WORLD direction (g0 = the transpose of the outermost ray turn applied to fn,
  // g1 = that of the two outermost turns applied to the inner normal). Formed in FP64 on the host and rounded once:
  // two dot products and a sign replace two maps, two ray turns and two dot products per record (rotate.clj:31-47,
  // translate.clj:13-17), and the record needs no second lookup by chain.
  float fA, fB, ftx, fty;
  float ftz, fnx, fny, fnz;
  float g0x, g0y, g0z, g1x;
  float g1y, g1z;
  int32_t finner;                          // DChainRec.inner
  int32_t fpad;
};
#define RT_LDS_PRIMS 256  // scenes up to this many flattened primitives keep DPrimInfo in LDS

struct DMedium {
  int32_t first_group, n_groups;  // boundary groups (indices into groups[])
  int32_t outer_chain;            // chain the medium itself lives under
  int32_t outer_group;            // an (empty) group that carries that chain's ray motion
  int32_t phase_mat;
  float neg_inv_density;
};

// One BVH node holds the boxes of both children so a traversal step is one 64-byte fetch.
// child < 0 encodes a leaf: primitive index = ~child (index into prims[]).
struct DBvhNode {
  float lmin[3], lmax[3];
  float rmin[3], rmax[3];
  int32_t left, right;
  int32_t pad[2];
};
// What the traversal reads: the same node in 32 bytes, each box as centre and half-extent in FP16 (the
// centre to the nearest half, the half-extent rounded up from what that centre needs to cover both
// bounds), so a step fetches half as much per lane and the slab test takes its operands straight
// from the halves (v_fma_mix_f32, nine per box and no min/max to sort near from far). A box can only
// grow: nothing is culled that the FP32 node would keep. A leaf reference carries the primitive's type
// beside its index (~child = index | RT_PRIM_* << RT_LEAF_TYPE_SHIFT), so the leaf test knows which
// fields to fetch without reading the primitive's type word first: one memory round trip per leaf, not two.
struct DBvhNode16 {
  uint16_t lc[3], lh[3];   // left child: centre, half-extent
  uint16_t rc[3], rh[3];   // right child
  int32_t left, right;
};
#define RT_LEAF_TYPE_SHIFT 28
#define RT_LEAF_INDEX_MASK ((1 << RT_LEAF_TYPE_SHIFT) - 1)

struct DMat {
  int32_t type;   // RT_MAT_*
  int32_t tex;
  int32_t flags;  // DMAT_*
  int32_t pad;
  float r, g, b, x;  // metal: albedo rgb, fuzz | dielectric: r = ir | DMAT_SOLID: the texture's colour
};
#define DMAT_NEEDS_UV 1  // its texture (or a checker branch of it) reads u/v: image textures only
#define DMAT_SOLID 2     // its texture is a solid colour, folded into r g b

struct DTex {
  int32_t type;  // RT_TEX_*
  int32_t i0, i1;  // checker: odd, even | noise: perlin id, scale | image: image id (-1: none), first texel
  float r, g, b;   // solid colour
  int32_t w, h;    // image: its size (a copy of the DImage, so a lookup is record -> texel, not record -> image -> texel)
};

struct DImage {
  int32_t w, h;
  int32_t first_texel;  // into texels[] (RGBA8, top row first)
  int32_t pad;
};

struct DScene {
  const DPrim* prims;            // what the intersection tests read
  const DPrim* prims_rec;        // what a sphere's record reads: the same table, except that a sphere tested in
                                 // world space (lent by a chain to a chain-less group) stands here in its chain's space
  const float* brute_rects;      // the rects of every brute-force group as PAIRS of one orientation, so one
                                 // packed-FP32 instruction serves two rects: k k' a0 a0' a1 a1' b0 b0' b1 b1'
                                 // then the two device primitive indices (int bits). Per group: XY pairs, XZ
                                 // pairs, YZ pairs. A run of odd length starts with a pair whose first slot
                                 // repeats its second (same index): the repeat is tested first and the real
                                 // rect, accepting a tie, overwrites it. RT_PAIR_SLACK records of slack follow.
  const float* brute_boxes;      // the closed boxes of the brute-force groups: x0 x1 y0 y1 z0 z1, then the device
                                 // primitive indices (int bits) of the faces z = z0, z = z1, y = y0, y = y1,
                                 // x = x0, x = x1. One record of slack follows.
  const DPrimInfo* prim_info;
  const DGroup* groups;
  const DGroup64* groups64;
  const DChainRec* chain_recs;
  const DMedium* media;
  const DBvhNode16* nodes;
  const DMat* mats;
  const DTex* texs;
  const float* perlin_vec;       // n_perlin * 256 * 4 (xyz + pad)
  const uint8_t* perlin_perm;    // n_perlin * 3 * 256
  const DImage* images;
  const uint32_t* texels;
  int32_t n_surface_groups;      // groups [0, n_surface_groups) are tested for every ray
  int32_t n_groups;
  int32_t n_chains;
  int32_t n_prims;
  int32_t n_media;
  int32_t n_perlin;
  int32_t bvh_stack;             // entries per lane the traversal stack needs (0 = no BVH)
  const int32_t* bvh_next;       // per group: the next surface group after it that has a tree, or -1
  int32_t first_bvh_group;       // the first surface group with a tree, or -1
  int32_t n_bvh_nodes;
  int32_t xch_on_stacks;         // 1 when the kernels that walk trees lay their exchange area over the traversal stacks
  int32_t lds_nodes;             // n_bvh_nodes when the kernels that walk trees keep the nodes' second halves in LDS, else 0
  uint32_t features;             // F_* mask of what this scene needs from a kernel instantiation
};

struct DCamera {  // the reference's camera map in FP64 (camera.clj:24-33)
  double origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3];
  double lens_radius, time0, time1;
  // core.clj:99,101 divide the jittered pixel coordinate by W-1 and H-1: the divisors and their
  // correctly rounded reciprocals, for an exact division in three FMAs (see camera_ray)
  double wm1, hm1, inv_wm1, inv_hm1;
  // A pinhole camera (lens radius 0, the setting of seven of the reference's eight worlds, core.clj:60-65) on an image
  // of at least 2 x 2 pixels: dir = dC + (x + jitter) dH + (y + jitter) dV with dC = llc - origin,
  // dH = horizontal / (W - 1), dV = vertical / (H - 1): two FP64 FMAs per axis instead of two exact divisions and the
  // lens terms. Against ray.clj:21-25 evaluated in FP64 this re-associates at the 1e-16 level, which the rounding of
  // the ray to FP32 hides (the two agree in all but about one ray in 1e8).
  double dC[3], dH[3], dV[3];
  float fO[3];
  int32_t pinhole;
  int32_t pad_;
};

struct DRender {
  int32_t width, height;
  int32_t spp, sample_offset;
  int32_t max_depth;
  int32_t shard_index, shard_count;
  int32_t tiles_x, tiles_y;
  int32_t n_owned_tiles;   // tiles this call renders
  int32_t n_chunks;        // a pixel's samples are split into n_chunks work items ("chunks")
  int32_t n_items;         // n_owned_tiles * 64 * n_chunks
  float bg[3];
  uint32_t seed_lo, seed_hi;
  int32_t chunk_start[RT_MAX_CHUNKS + 1];  // chunk c covers samples [chunk_start[c], chunk_start[c+1])
};

// Which shard renders which 8x8 tile. The tiles, in row-major order, are dealt in groups of `count` consecutive
// tiles, one to each shard; the deal moves on by one place with every tile row the groups start in, so that no shard
// owns fixed tile columns of the image (columns differ in cost: with fixed columns one shard of eight of the Cornell
// box took 3.5 % longer than another). A shard's k-th tile lies in group k.
#if defined(__HIPCC__)
#define RT_SHARD_FN __host__ __device__ inline
#else
#define RT_SHARD_FN inline
#endif
RT_SHARD_FN int shard_turn(int group, int count, int tiles_x) { return (group * count / tiles_x) % count; }
RT_SHARD_FN int shard_tile(int group, int index, int count, int tiles_x) {
  return group * count + (index + shard_turn(group, count, tiles_x)) % count;
}
RT_SHARD_FN int tile_shard(int tile, int count, int tiles_x) {
  return (tile % count + count - shard_turn(tile / count, count, tiles_x)) % count;
}

#endif  // RT_SCENE_H
