/*
 * rtmi.h — C ABI of librtmi.so, the MI355X (gfx950) implementation of the
 * per-pixel path-tracing sample loop of MartinPuda/raytracing.
 *
 * The reference has no FFI of its own. The seam this library plugs into is the
 * body of `-main` between camera construction and `display-image`
 * (reference src/raytracing/core.clj:96-128): given the world vector built by
 * the `scenes` constructors, the camera map and W, H, spp, depth, background,
 * produce one packed-RGB `int` per pixel (what `setRGB` consumes at
 * core.clj:123-126) and/or the per-pixel radiance sums.
 *
 * Conventions
 *   - plain C types only; every pointer argument is caller-owned and only read
 *     (or written, for outputs) during the call; the library copies what it keeps.
 *   - return value: 0 = ok, negative = error (RT_ERR_*); rt_last_error() returns a
 *     thread-local, library-owned message for the last failing call on this thread.
 *   - no exceptions, no aborts, no callbacks cross this boundary.
 *   - calls on one context must be serialised by the caller, and a context has ONE render in
 *     flight at a time: renders share the context's work queue and partial sums, so a second
 *     rt_render_device must not be enqueued (on any stream) before the first has finished.
 *   - there is no CPU fallback: without a HIP device rt_init fails with RT_ERR_HIP.
 */
#ifndef RTMI_H
#define RTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTMI_ABI_VERSION 2

enum {
  RT_OK = 0,
  RT_ERR_ARG = -1,      /* malformed argument / scene description */
  RT_ERR_HIP = -2,      /* HIP runtime error or no device */
  RT_ERR_UNSUPPORTED = -3, /* valid reference construct this build cannot flatten */
  RT_ERR_OOM = -4
};

/* ---- primitive kinds: one per reference Hittable leaf record ------------------ */
enum {
  RT_PRIM_SPHERE = 0,        /* base.clj:229-258   prim_d: cx cy cz r                      */
  RT_PRIM_MOVING_SPHERE = 1, /* base.clj:263-302   prim_d: c0x c0y c0z c1x c1y c1z t0 t1 r */
  RT_PRIM_XY_RECT = 2,       /* rectangles.clj:8-40    prim_d: x0 x1 y0 y1 k               */
  RT_PRIM_XZ_RECT = 3,       /* rectangles.clj:45-77   prim_d: x0 x1 z0 z1 k               */
  RT_PRIM_YZ_RECT = 4        /* rectangles.clj:82-114  prim_d: y0 y1 z0 z1 k               */
};
#define RT_PRIM_STRIDE 12    /* doubles per primitive in prim_d (unused tail = 0)         */

/* ---- hittable tree nodes: one per reference Hittable aggregate / wrapper ------- */
enum {
  RT_NODE_PRIM = 0,      /* node_i: prim index, -, -, -                                   */
  RT_NODE_LIST = 1,      /* HittableList / Box sides / the world vector
                            (base.clj:90-101, box.clj:8-11). node_i: first, count into
                            node_children (order = the reference's scan order)            */
  RT_NODE_BVH = 2,       /* BvhNode (base.clj:116-124). node_i: left node, right node;
                            node_d: box min xyz, box max xyz                              */
  RT_NODE_TRANSLATE = 3, /* translate.clj:8-17. node_i: child node; node_d: offset xyz    */
  RT_NODE_ROTATE_Y = 4,  /* rotate.clj:9-47.   node_i: child node; node_d: sin, cos       */
  RT_NODE_MEDIUM = 5     /* constant_medium.clj:9-44. node_i: boundary node, phase
                            material, medium index (0..n_media-1, DFS order);
                            node_d: neg_inv_density                                       */
};
#define RT_NODE_ISTRIDE 4
#define RT_NODE_DSTRIDE 6

/* ---- materials (materials.clj) ------------------------------------------------- */
enum {
  RT_MAT_LAMBERTIAN = 0,    /* :14-24  mat_tex = albedo texture                            */
  RT_MAT_METAL = 1,         /* :31-40  mat_d = albedo r g b, fuzz (as stored: the `metal`
                                       constructor's clamp is the caller's job)           */
  RT_MAT_DIELECTRIC = 2,    /* :45-62  mat_d = ir                                          */
  RT_MAT_DIFFUSE_LIGHT = 3, /* :67-78  mat_tex = emit texture                              */
  RT_MAT_ISOTROPIC = 4      /* :85-91  mat_tex = albedo texture                            */
};
#define RT_MAT_DSTRIDE 4

/* ---- textures (texture.clj) ---------------------------------------------------- */
enum {
  RT_TEX_SOLID = 0,   /* :15-17  tex_d = r g b                                             */
  RT_TEX_CHECKER = 1, /* :23-30  tex_i = odd texture, even texture                         */
  RT_TEX_NOISE = 2,   /* :38-44  tex_i = perlin table index, scale (already truncated to
                                 long as the record field does)                            */
  RT_TEX_IMAGE = 3    /* :50-66  tex_i = image index, or -1 for the no-data texture
                                 that returns [0 1 1]                                      */
};
#define RT_TEX_DSTRIDE 4
#define RT_TEX_ISTRIDE 2

/*
 * Flat description of a world: what a host-side flattener produces by walking the
 * reference's records ((:objects …), (:left/:right/:box …), (:ptr …), (:boundary …),
 * (:sides …)). All geometry is FP64 exactly as the reference holds it; the library
 * derives its own device formats.
 */
typedef struct rt_scene_desc {
  int32_t abi_version;          /* RTMI_ABI_VERSION */

  int32_t n_prims;
  const int32_t* prim_type;     /* n_prims */
  const double*  prim_d;        /* n_prims * RT_PRIM_STRIDE */
  const int32_t* prim_mat;      /* n_prims */

  int32_t n_nodes;
  const int32_t* node_kind;     /* n_nodes */
  const int32_t* node_i;        /* n_nodes * RT_NODE_ISTRIDE */
  const double*  node_d;        /* n_nodes * RT_NODE_DSTRIDE */
  int32_t n_children;
  const int32_t* node_children; /* n_children: child node ids of LIST nodes */
  int32_t root;                 /* node id of the world (normally a LIST) */
  int32_t n_media;              /* number of RT_NODE_MEDIUM nodes */

  int32_t n_mats;
  const int32_t* mat_type;      /* n_mats */
  const double*  mat_d;         /* n_mats * RT_MAT_DSTRIDE */
  const int32_t* mat_tex;       /* n_mats (-1 when unused) */

  int32_t n_tex;
  const int32_t* tex_type;      /* n_tex */
  const double*  tex_d;         /* n_tex * RT_TEX_DSTRIDE */
  const int32_t* tex_i;         /* n_tex * RT_TEX_ISTRIDE */

  int32_t n_perlin;             /* noise.clj:72-83: one table set per `perlin` call */
  const double*  perlin_ranvec; /* n_perlin * 256 * 3 */
  const int32_t* perlin_perm;   /* n_perlin * 3 * 256  (perm-x, perm-y, perm-z) */

  int32_t n_images;             /* texture.clj:68-81: decoded on the host */
  const int32_t* img_w;
  const int32_t* img_h;
  const uint8_t* const* img_rgb8; /* per image: h*w*3 bytes, top row first, R G B */
} rt_scene_desc;

/* The camera map, field for field (camera.clj:24-33). */
typedef struct rt_camera {
  double origin[3], lower_left_corner[3], horizontal[3], vertical[3];
  double u[3], v[3], w[3];
  double lens_radius, time0, time1;
} rt_camera;

/*
 * One render request = the constants of `-main` (core.clj:52-59) plus sharding.
 * A unit of work is one (pixel, sample); its random stream depends only on
 * (seed, pixel index y*width+x, global sample index), so any split of the work
 * over devices or calls reproduces the same per-sample radiance.
 */
typedef struct rt_params {
  int32_t width, height;  /* each 1..65535, width*height <= 2^28                       */
  int32_t spp;            /* samples rendered by THIS call per pixel                      */
  int32_t sample_offset;  /* global index of this call's first sample (sample slicing)   */
  int32_t max_depth;      /* core.clj:58 (50)                                             */
  int32_t shard_index;    /* this call renders one 8x8-pixel tile of every group of       */
  int32_t shard_count;    /*   shard_count consecutive tiles (row-major tile order): tile */
                          /*   t belongs to shard (t % n + n - turn) % n, n = shard_count, */
                          /*   turn = ((t / n) * n / tiles_per_row) % n: the deal moves on */
                          /*   by one place per tile row, so no shard owns fixed columns   */
  int32_t flags;          /* RT_FLAG_*                                                    */
  double  background[3];  /* core.clj:30,60-65                                            */
  uint64_t seed;
} rt_params;

enum {
  RT_FLAG_COUNTERS = 1,   /* run the event-counting build of the path kernel and fill
                             rt_stats.events (slower; never use its timing)               */
  RT_FLAG_POOL_PATHS = 2  /* worlds with trees: render on the kernel that keeps a pool of paths per
                             wave in LDS and re-packs them onto the lanes between the stages of a
                             bounce (camera ray, brute-force groups, tree walk, leaf tests,
                             record + scatter) instead of the one in which a lane owns a path from
                             camera ray to termination. Same frame, bit for bit. Its stages run
                             with 85-95 % of the lanes busy (34 % in the default kernel), but the
                             pool's LDS leaves room for two waves per SIMD and the tree walk then
                             waits on its node fetches: it is 2.5 x slower today. Ignored for
                             worlds without trees.                                            */
};

/* Event counters, indexed by RT_EV_*; the unit costs live next to the enum. */
enum {
  RT_EV_AABB_TEST = 0, RT_EV_SPHERE_TEST, RT_EV_SPHERE_ROOT, RT_EV_SPHERE_RECORD,
  RT_EV_MSPHERE_CENTER, RT_EV_RECT_TEST, RT_EV_RECT_RECORD,
  RT_EV_ROTATE_RAY, RT_EV_ROTATE_REC, RT_EV_TRANSLATE_RAY, RT_EV_TRANSLATE_REC,
  RT_EV_MEDIUM_SAMPLE, RT_EV_CAMERA_RAY, RT_EV_SPHERE_TRIAL, RT_EV_DISK_TRIAL,
  RT_EV_LAMBERT, RT_EV_METAL, RT_EV_DIELECTRIC, RT_EV_ISOTROPIC, RT_EV_LIGHT,
  RT_EV_CHECKER, RT_EV_PERLIN_NOISE, RT_EV_NOISE_TEX, RT_EV_IMAGE_TEX,
  RT_EV_THROUGHPUT, RT_EV_ACCUMULATE,
  RT_EV_BOX_SLAB,     /* six rects that close a box, intersected as one slab test (box.clj:8-24):
                         what the kernel executes in place of six RT_EV_RECT_TEST               */
  RT_EV_COUNT
};

typedef struct rt_stats {
  uint64_t samples;              /* camera paths started                                  */
  uint64_t rays;                 /* ray-color invocations that reached `hit` (core.clj:23) */
  uint64_t events[RT_EV_COUNT];  /* only with RT_FLAG_COUNTERS, else 0                    */
  double   kernel_ms;            /* HIP-event time of the path kernel alone, on its stream */
  double   shader_mhz;           /* shader clock DURING the path kernel: one wave reads the cycle
                                    counter and the constant-rate timer when it starts and when
                                    it ends; 0 when the device reports no timer rate            */
} rt_stats;

/* One closest-hit record = the reference's HitRecord (base.clj:169) as `ray-color`
 * would receive it from (hit (->HittableList world) r t-min t-max {}). */
typedef struct rt_hit {
  float t, p[3], normal[3], u, v;
  int32_t front_face, mat_id, hit; /* hit = 0: miss (all other fields 0) */
} rt_hit;

typedef struct rt_ctx rt_ctx;
typedef struct rt_scene rt_scene;

/* Binds the context to one HIP device (one process per GPU). Replaces nothing in the
 * reference (it has no device notion); it is the price of entry for every other call. */
int rt_init(int32_t device_id, rt_ctx** out);
void rt_shutdown(rt_ctx* ctx);
const char* rt_last_error(void);
int rt_abi_version(void);

/* Uploads a world and builds its device form (flattened instance chains, GPU-built BVHs).
 * Stands in for holding the `world` vector that core.clj:66-88 selects. */
int rt_scene_create(rt_ctx* ctx, const rt_scene_desc* desc, rt_scene** out);
/* The same with a choice of tree builder for the large groups (the reference builds its own
 * BvhNode trees at base.clj:137-164; the closest hit does not depend on the tree's shape, so the
 * library builds its own over the flattened leaves). Records and frames are identical, bit for
 * bit, whichever builder ran; only the speed of the traversal differs. */
enum {
  RT_BVH_DEFAULT = 0,      /* what rt_scene_create uses: RT_BVH_DEVICE_SAH up to 2^20 primitives per
                              group, RT_BVH_DEVICE_LBVH beyond                                  */
  RT_BVH_DEVICE_SAH = 1,   /* built on the GPU, top-down: exact surface-area sweep over centroid
                              orders sorted once per axis, one launch per tree level             */
  RT_BVH_HOST_SAH = 2,     /* the same sweep on the host, the cross-check of RT_BVH_DEVICE_SAH: both
                              give the same tree (up to 32768 primitives per group; larger groups
                              are built as RT_BVH_DEFAULT builds them)                           */
  RT_BVH_DEVICE_LBVH = 3   /* built on the GPU, bottom-up: Morton order, radix tree, box fit, then
                              rotation sweeps; the quickest build and a 3-9 % slower traversal    */
};
int rt_scene_create_with(rt_ctx* ctx, const rt_scene_desc* desc, int32_t bvh_builder, rt_scene** out);
/* A digest (FNV-1a, 64 bit) of the tree records as the builder wrote them, the number of records and
 * the height of the tallest tree: what a test compares between builders (RT_BVH_DEVICE_SAH and
 * RT_BVH_HOST_SAH give equal digests). No counterpart in the reference, whose trees are random
 * (base.clj:143). */
int rt_scene_bvh_digest(rt_scene* scene, uint64_t* digest, int32_t* n_nodes, int32_t* height);
void rt_scene_destroy(rt_scene* scene);

/* The sample loop of core.clj:96-127 over this call's shard of tiles and slice of samples.
 *   rgb_sum: width*height*3 floats, row 0 = y 0 = BOTTOM image row (the reference flips rows
 *            only when writing to the BufferedImage, core.clj:123-126); sums over this call's
 *            samples; pixels of other shards are written as 0. May be NULL.
 *   argb:    width*height packed 0xFFRRGGBB after x(1/spp) -> sqrt -> clamp[0,0.999] ->
 *            (int)(c*255+0.5), NaN -> 0 (core.clj:106-108,118,121-122; java.awt.Color(float,
 *            float,float)); only meaningful when the call covers all samples of its pixels.
 *            May be NULL.
 *   stats:   may be NULL.
 * rt_render takes HOST buffers (copies back over PCIe); rt_render_device takes DEVICE
 * buffers on the context's GPU and enqueues on `hip_stream` (a hipStream_t, NULL = the
 * null stream) without synchronising unless stats is non-NULL. */
int rt_render(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, const rt_params* params,
              float* rgb_sum, uint32_t* argb, rt_stats* stats);
int rt_render_device(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam,
                     const rt_params* params, float* d_rgb_sum, uint32_t* d_argb,
                     rt_stats* stats, void* hip_stream);

/* rt_render_device for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc). It has no counterpart in the reference.
 * The render's scratch (work-queue heads, counters, one 12-byte partial sum per work item) lives in
 * a fixed 32 MiB device array inside the library instead of in buffers allocated on demand, so the
 * call allocates and frees nothing and never synchronises. Results are bit-identical to
 * rt_render_device's. Limits: at most 2 795 861 work items per call (work items = owned 8x8 tiles
 * x 64 x sample chunks, where spp <= 32 gives spp chunks, spp <= 128 gives 32 and larger spp up to
 * 112: e.g. 256x256 at any spp, 1024x1024 up to 2 spp per call); larger requests are refused with
 * RT_ERR_UNSUPPORTED. No statistics (RT_FLAG_COUNTERS is refused). All calls on one GPU share the
 * array: two of them in flight at once on that GPU (any streams, any contexts) race. */
int rt_render_device_noalloc(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam,
                             const rt_params* params, float* d_rgb_sum, uint32_t* d_argb,
                             void* hip_stream);

/* The post-process of core.clj:106-108,118,121-122 alone, device buffers: used on the root
 * rank after per-GPU sums have been combined. `spp_total` is the divisor. */
int rt_tonemap_device(rt_ctx* ctx, const float* d_rgb_sum, int32_t n_pixels, int32_t spp_total,
                      uint32_t* d_argb, void* hip_stream);

/* Parity hook: closest hit for a batch of rays exactly as ray-color's `hit` call would
 * return it (core.clj:23). orig3/dir3: n*3 floats, time: n floats (NULL = 0).
 * medium_xi: n * n_media scripted uniforms for the free-flight draw of each medium
 * (constant_medium.clj:32), or NULL to skip MEDIUM nodes. Host buffers.
 * t_min < 0 is refused (RT_ERR_UNSUPPORTED) for worlds that list the same sphere twice: the
 * duplicate is left out of the scan, which is exact only for t_min >= 0 (ray-color asks 0.001). */
int rt_hit_batch(rt_ctx* ctx, const rt_scene* scene, int32_t n, const float* orig3,
                 const float* dir3, const float* time, float t_min, float t_max,
                 const float* medium_xi, rt_hit* out);

/* The kernels are compiled once per feature set ("instantiation": 0 rects under chains,
 * 1 + constant media, 2 spheres and rects in plain lists with metal / dielectric / checker / image
 * textures, 3 the same with BVHs, 4 spheres and rects with Perlin noise, 5 everything the
 * reference's worlds use, 6 + media with arbitrary boundaries, 7 spheres alone in trees, solid colours and checkers) and a world
 * renders on the smallest one that covers it. rt_scene_get_variants reports which one that is and
 * the bit mask of all that cover the world; rt_hit_batch_variant answers rt_hit_batch's question
 * with a chosen instantiation (RT_VARIANT_RENDER = the one the render uses, RT_VARIANT_GENERAL =
 * instantiation 6, which is what rt_hit_batch runs), so a test can show that every instantiation
 * returns the same records. An instantiation that does not cover the world is refused
 * (RT_ERR_UNSUPPORTED). */
#define RT_N_VARIANTS 8
#define RT_VARIANT_GENERAL (-1)
#define RT_VARIANT_RENDER (-2)
int rt_scene_get_variants(const rt_scene* scene, int32_t* render_variant, uint32_t* covering_mask);
int rt_hit_batch_variant(rt_ctx* ctx, const rt_scene* scene, int32_t variant, int32_t n, const float* orig3,
                         const float* dir3, const float* time, float t_min, float t_max,
                         const float* medium_xi, rt_hit* out);

/* Parity hook for (scatter mat r rec) / (emitted mat u v p), materials.clj:10-96: one ray-color
 * invocation (core.clj:23-25) per ray with t-min 0.001, t-max inf and the keyed random draws of
 * pixel px[i], sample smp[i], bounce bounce[i] (seed = params' seed). Runs on the instantiation
 * the render uses. hit = 0: the ray left the world (other fields 0). ok = 1: `orig dir time` are
 * the scattered ray and `attenuation` its colour; ok = 0: `emitted` is the material's emission. */
typedef struct rt_scatter {
  int32_t hit, ok, mat_id, pad;
  float orig[3], dir[3], time;
  float attenuation[3], emitted[3];
} rt_scatter;
int rt_scatter_batch(rt_ctx* ctx, const rt_scene* scene, uint64_t seed, int32_t n, const float* orig3,
                     const float* dir3, const float* time, const int32_t* px, const int32_t* smp,
                     const int32_t* bounce, rt_scatter* out);

/* Parity hook for the scatter/texture functions: per-sample radiance of single samples,
 * host buffers. For i in [0,n): pixel index px[i], global sample index smp[i] ->
 * out_rgb[3*i..] = ray-color of that one camera path (core.clj:96-105). */
int rt_sample_batch(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam,
                    const rt_params* params, int32_t n, const int32_t* px, const int32_t* smp,
                    float* out_rgb);

/*
 * Scene files: the rt_scene_desc as one little-endian blob, so that a host that cannot
 * pass nested pointers comfortably (the JVM) writes bytes and hands over one buffer or path.
 * The reference has no scene format; this stands in for passing the `world` vector of
 * core.clj:66-88 by reference.
 *
 *   offset 0   char    magic[8] = "RTMISCN1"
 *          8   int32   abi_version
 *         12   int32   n_prims, n_nodes, n_children, root, n_media, n_mats, n_tex, n_perlin,
 *                      n_images
 *         48   int32   reserved[4] = 0
 *         64   the arrays of rt_scene_desc in struct order (prim_type, prim_d, prim_mat,
 *              node_kind, node_i, node_d, node_children, mat_type, mat_d, mat_tex, tex_type,
 *              tex_d, tex_i, perlin_ranvec, perlin_perm, img_w, img_h), then image 0 ..
 *              n_images-1 (h*w*3 bytes, top row first, R G B); every array and image is
 *              zero-padded to a multiple of 8 bytes, so all offsets follow from the counts.
 *
 * rt_scene_file_size: bytes rt_scene_write produces for `desc`, 0 if the description is
 *   invalid (rt_last_error says why). Neither needs a context or a device.
 * rt_scene_write: fills `bytes` (capacity >= that size); equal descriptions give equal bytes.
 * rt_scene_create_from_memory / _from_file: validate the blob (magic, counts, truncation)
 *   and then behave exactly as rt_scene_create on the description it holds; the buffer is
 *   only read during the call.
 */
uint64_t rt_scene_file_size(const rt_scene_desc* desc);
int rt_scene_write(const rt_scene_desc* desc, void* bytes, uint64_t capacity, uint64_t* written);
int rt_scene_create_from_memory(rt_ctx* ctx, const void* bytes, uint64_t n_bytes, rt_scene** out);
int rt_scene_create_from_file(rt_ctx* ctx, const char* path, rt_scene** out);

/*
 * (image-texture url), texture.clj:68-81: the reference reads its texture files with
 * javax.imageio.ImageIO/read and keeps rows of [R G B] bytes (texture.clj:74-80). This call decodes
 * a JPEG file held in memory (resources/earthmap.jpg and the two files beside it are JPEGs, whatever
 * map2.png is called) to that layout: height * width * 3 bytes, top row first, R G B - what
 * rt_scene_desc.img_rgb8 takes. Huffman-coded 8-bit JPEG, sequential or progressive, one or three
 * components, any integral sampling factors, restart intervals; reconstruction as the IJG decoder
 * behind ImageIO does it by default (accurate integer inverse DCT, triangle-filter chroma upsampling,
 * 16-bit fixed-point YCbCr -> RGB), so the texels are that decoder's. Host code: no context and no
 * GPU needed. rgb8 = NULL only reports the size. RT_ERR_ARG: not a JPEG file, a damaged one, or
 * capacity < width * height * 3; RT_ERR_UNSUPPORTED: arithmetic coding, lossless / hierarchical
 * processes, 12-bit samples, four components.
 */
int rt_image_decode_jpeg(const uint8_t* bytes, uint64_t n_bytes, int32_t* width, int32_t* height,
                         uint8_t* rgb8, uint64_t capacity);

/*
 * The same call for a PNG file (texture.clj:72: ImageIO/read takes either): every colour type and bit
 * depth of ISO/IEC 15948, Adam7 interlacing, chunk CRCs and the Adler-32 of the image data checked.
 * An 8-bit RGB file gives exactly the bytes the reference's raster read gives (texture.clj:73-77). The
 * reference's read assumes three bytes per pixel, so it scrambles grey, palette and RGBA rasters and
 * throws on 16-bit ones; here those become what the file shows: grey replicated to R = G = B (1, 2
 * and 4 bits scaled to 0..255), palette entries looked up, 16-bit samples rounded to 8 bits, alpha
 * and tRNS dropped without compositing. Same arguments and errors as rt_image_decode_jpeg;
 * RT_ERR_UNSUPPORTED: more than 2^28 pixels.
 */
int rt_image_decode_png(const uint8_t* bytes, uint64_t n_bytes, int32_t* width, int32_t* height,
                        uint8_t* rgb8, uint64_t capacity);

/*
 * The other two formats ImageIO/read (texture.clj:72) reads out of the box. GIF: the first image of the
 * file at its own size (what ImageIO/read returns), through its local or the global colour table,
 * interlaced or not, transparency ignored. BMP: 1-, 4-, 8-bit palette, 16-, 24-, 32-bit, bottom-up or
 * top-down, uncompressed or bit fields; RT_ERR_UNSUPPORTED for run-length coded files. A 24-bit BMP
 * gives exactly the bytes the reference's raster read gives (texture.clj:73-77); the palette and
 * other depths, which that read scrambles or throws on, become the picture the file holds. Same
 * arguments and errors as rt_image_decode_jpeg; RT_ERR_UNSUPPORTED: more than 2^28 pixels.
 */
int rt_image_decode_gif(const uint8_t* bytes, uint64_t n_bytes, int32_t* width, int32_t* height,
                        uint8_t* rgb8, uint64_t capacity);
int rt_image_decode_bmp(const uint8_t* bytes, uint64_t n_bytes, int32_t* width, int32_t* height,
                        uint8_t* rgb8, uint64_t capacity);

/*
 * ImageIO/read itself (texture.clj:72): picks the decoder by the file's first bytes, not by its name
 * (resources/map2.png is a JPEG and loads as one). RT_ERR_UNSUPPORTED: not a JPEG, PNG, GIF or BMP
 * file (WBMP and TIFF are not read).
 */
int rt_image_decode(const uint8_t* bytes, uint64_t n_bytes, int32_t* width, int32_t* height,
                    uint8_t* rgb8, uint64_t capacity);

/*
 * Several GPUs behind one handle, for a caller that is ONE process (the JVM): what replaces the
 * reference's fan-out of a frame over one thread per image row and its latch (core.clj:44-48,
 * 95,109-128). rt_multi_init binds one context per listed device; rt_multi_scene_create uploads
 * the world to each of them; rt_multi_render runs one host thread per device, each rendering its
 * shard through rt_render_device (RT_SPLIT_TILES: device i of n owns the 8x8 tiles that rt_params'
 * shard_index = i, shard_count = n name - one of every n consecutive tiles, the place moving on by
 * one per tile row - all samples, and the frame is bit-identical to a single-GPU render;
 * RT_SPLIT_SAMPLES: device i renders samples [spp*i/n, spp*(i+1)/n) of every pixel, and the FP32
 * sums differ from a single-GPU render by re-association only), then ONE RCCL reduce(sum) of the
 * width*height*3 FP32 accumulators onto the first device (over xGMI), the pixel post-process there,
 * and the copy to the caller's host buffers (same meaning as rt_render's). params->shard_index /
 * shard_count must be 0. RCCL (librccl.so) is loaded on first need: never for one device, and a
 * list that names ONE device several times (the fan-out rehearsed on a one-GPU machine) is summed
 * by a kernel on that device in list order. (The reduce between distinct devices is exercised by
 * tests/test_gpu_multi.py wherever two or more GPUs are visible; a one-GPU machine can only run
 * the repeated-device form and a one-rank RCCL self-test.) Errors of these calls are reported by
 * rt_multi_last_error().
 */
typedef struct rt_multi rt_multi;
typedef struct rt_multi_scene rt_multi_scene;
enum { RT_SPLIT_TILES = 0, RT_SPLIT_SAMPLES = 1 };
int rt_multi_init(const int32_t* device_ids, int32_t n_devices, rt_multi** out);
void rt_multi_shutdown(rt_multi* m);
int rt_multi_device_count(const rt_multi* m);
/* Loads RCCL and makes the calls rt_multi_render makes (communicator, grouped in-place float
 * sum-reduce to rank 0, teardown) on ONE device, where the reduce must return its input. */
int rt_multi_selftest_rccl(int32_t device_id);
const char* rt_multi_last_error(void);
int rt_multi_scene_create(rt_multi* m, const rt_scene_desc* desc, rt_multi_scene** out);
void rt_multi_scene_destroy(rt_multi_scene* scene);
int rt_multi_render(rt_multi* m, const rt_multi_scene* scene, const rt_camera* cam, const rt_params* params,
                    int32_t split, float* rgb_sum, uint32_t* argb, rt_stats* stats);

/* What the context's device reports about itself (hipGetDeviceProperties): the numbers a caller
 * needs to price a render against the device's peak. */
typedef struct rt_device_info {
  int32_t n_cu;              /* compute units                                             */
  int32_t clock_khz;         /* peak shader clock                                         */
  int32_t lds_bytes_max;     /* LDS one workgroup may use                                 */
  int32_t timer_khz;         /* rate of the constant timer rt_stats.shader_mhz is taken against */
} rt_device_info;
int rt_get_device_info(const rt_ctx* ctx, rt_device_info* out);

/* Introspection for tests and reports. */
typedef struct rt_scene_info {
  int32_t n_prims_flat;   /* leaves reachable from the root, deduplicated          */
  int32_t n_groups;       /* (instance chain, primitive set) groups                */
  int32_t n_chains;
  int32_t n_media;
  int32_t n_bvh_nodes;    /* over all GPU-built BVHs                               */
  int32_t lds_bytes;      /* per-workgroup LDS the path kernel uses for this scene */
} rt_scene_info;
int rt_scene_get_info(const rt_scene* scene, rt_scene_info* out);

#ifdef __cplusplus
}
#endif
#endif /* RTMI_H */
