// rt_launch.h — host-callable launch wrappers around the kernels (internal to librtmi.so).
#ifndef RT_LAUNCH_H
#define RT_LAUNCH_H

#include <hip/hip_runtime.h>

#include "../../include/rtmi.h"
#include "rt_scene.h"

#ifndef RT_BLOCK
#define RT_BLOCK 256  // threads per workgroup of the path kernels (4 waves)
#endif

// Trees of up to this many nodes keep the second half of each packed node in LDS (see bvh_traverse).
#ifndef RT_LDS_NODE_MAX
#define RT_LDS_NODE_MAX 1280
#endif

// Boxes with a sunk lid as single tree leaves (RT_PRIM_LIDBOX); 0 keeps their six rects as six leaves.
#ifndef RT_LIDBOX_LEAVES
#define RT_LIDBOX_LEAVES 1
#endif

namespace rt {

size_t shade_lds_bytes(const DScene& S, int block);
// Lets every kernel that shades ask for `bytes` of dynamic LDS per workgroup (a no-op up to the
// 64 KiB every kernel may use without asking; the device allows 160 KiB).
hipError_t allow_lds(size_t bytes);
hipError_t launch_render(const DScene& S, const DCamera& cam, const DRender& R, float* out, uint32_t* queue,
                         unsigned long long* stats, bool count, bool pool, int grid, hipStream_t stream);
hipError_t render_occupancy(const DScene& S, bool count, bool pool, int* blocks_per_cu);
// k_render_pool: the path loop that re-packs paths between stages (worlds with trees).
bool pool_kernel_covers(const DScene& S);
size_t pool_lds_bytes(const DScene& S, int block);
hipError_t launch_resolve(const DRender& R, const float* partial, float* rgb_sum, uint32_t* argb,
                          hipStream_t stream);
hipError_t launch_tonemap(const float* rgb_sum, int n_pixels, int spp, uint32_t* argb, hipStream_t stream);
// Instantiations of the kernels: 0 V_RECT, 1 V_SMOKE, 2 V_LIST, 3 V_SPH, 4 V_NOISE, 5 V_ALL_DEEP, 6 V_FULL, 7 V_BALLS.
#ifndef RT_N_VARIANTS
#define RT_N_VARIANTS 8
#endif
bool variant_covers(const DScene& S, int variant);
int render_variant(const DScene& S);
size_t variant_lds_bytes(const DScene& S, int block, int variant);
hipError_t launch_hit_batch(const DScene& S, int variant, int n, const float* orig3, const float* dir3,
                            const float* time, float tmin, float tmax, const float* medium_xi, rt_hit* out,
                            hipStream_t stream);
hipError_t launch_scatter_batch(const DScene& S, uint32_t seed_lo, uint32_t seed_hi, int n, const float* orig3,
                                const float* dir3, const float* time, const int* px, const int* smp,
                                const int* bounce, rt_scatter* out, hipStream_t stream);
hipError_t launch_sample_batch(const DScene& S, const DCamera& cam, const DRender& R, int n, const int* px,
                               const int* smp, float* out_rgb, hipStream_t stream);

// Builds a BVH over prims[first, first+n) on the GPU (n >= 2). Writes n-1 nodes to
// nodes[node_base, node_base+n-1); the root is nodes[node_base]. time_lo/time_hi bound the ray
// times moving spheres must be enclosed for. Returns the tree height in *height.
hipError_t build_bvh(const DPrim* d_prims, int first, int n, DBvhNode* d_nodes, int node_base, float time_lo,
                     float time_hi, int* height, hipStream_t stream);

// The same tree contract, built on the GPU top-down by the surface-area heuristic (exact sweep over
// presorted centroid orders, one launch per level). Gives the tree build_bvh_sah gives.
#define RT_SWEEP_MAX (1 << 20)
hipError_t build_bvh_sweep(const DPrim* d_prims, int first, int n, DBvhNode* d_nodes, int node_base, float time_lo,
                           float time_hi, int* height, hipStream_t stream);

// The same tree contract, built on the host by the surface-area heuristic (exact sweep): for
// groups of up to RT_SAH_MAX primitives. h_prims is the host copy of the primitive records.
#define RT_SAH_MAX 32768
hipError_t build_bvh_sah(const DPrim* h_prims, int first, int n, DBvhNode* d_nodes, int node_base, float time_lo,
                         float time_hi, int* height, hipStream_t stream);

// Packs n nodes into the 32-byte form the traversal reads (FP16 boxes rounded outwards).
hipError_t pack_bvh_nodes(const DBvhNode* d_nodes, const DPrim* d_prims, DBvhNode16* d_packed, int n, hipStream_t stream);

}  // namespace rt
#endif
