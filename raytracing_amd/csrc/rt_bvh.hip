// rt_bvh.hip — GPU construction of the bounding-volume hierarchy of one primitive group.
//
// The reference builds its BvhNode trees on the host by random-axis median splits
// (base.clj:137-164) and its RotateY/Translate bounding boxes are unusable (rotate.clj:72-80,
// translate.clj:18-21). Closest-hit results do not depend on the tree shape, so the device
// builds its own: Morton codes of primitive centroids, a bitonic sort, Karras' parallel radix
// tree ("Maximizing parallelism in the construction of BVHs, octrees, and k-d trees", HPG 2012)
// and a bottom-up box fit done as one kernel launch per tree level (kernel boundaries give the
// cross-workgroup visibility a single-launch atomic scheme would have to fence for).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

#include "rt_launch.h"

#ifndef RT_BVH_SWEEPS
#define RT_BVH_SWEEPS 4
#endif

namespace rt {
namespace {

struct Box {
  float mn[3], mx[3];
};

__host__ __device__ __forceinline__ void prim_box(const DPrim& P, float time_lo, float time_hi, Box& b) {
  const int type = P.type & 15;
  if (type == RT_PRIM_SPHERE || type == RT_PRIM_MOVING_SPHERE) {
    double r = fabs(type == RT_PRIM_SPHERE ? P.d[3] : P.d[8]);
    double c0[3], c1[3];
    if (type == RT_PRIM_SPHERE) {
      for (int a = 0; a < 3; ++a) c0[a] = c1[a] = P.d[a];
    } else {  // the centre is linear in time (base.clj:298-302): bound its two extreme positions
      double k0 = ((double)time_lo - P.d[6]) / (P.d[7] - P.d[6]);
      double k1 = ((double)time_hi - P.d[6]) / (P.d[7] - P.d[6]);
      for (int a = 0; a < 3; ++a) {
        c0[a] = P.d[a] + (P.d[3 + a] - P.d[a]) * k0;
        c1[a] = P.d[a] + (P.d[3 + a] - P.d[a]) * k1;
      }
    }
    for (int a = 0; a < 3; ++a) {
      b.mn[a] = (float)(fmin(c0[a], c1[a]) - r);
      b.mx[a] = (float)(fmax(c0[a], c1[a]) + r);
    }
  } else if (type == RT_PRIM_LIDBOX) {
    for (int a = 0; a < 3; ++a) { b.mn[a] = P.f[2 * a]; b.mx[a] = P.f[2 * a + 1]; }
  } else {
    int ia, ib, ic;
    if (type == RT_PRIM_XY_RECT) { ia = 0; ib = 1; ic = 2; }
    else if (type == RT_PRIM_XZ_RECT) { ia = 0; ib = 2; ic = 1; }
    else { ia = 1; ib = 2; ic = 0; }
    b.mn[ia] = P.f[0]; b.mx[ia] = P.f[1];
    b.mn[ib] = P.f[2]; b.mx[ib] = P.f[3];
    b.mn[ic] = P.f[4]; b.mx[ic] = P.f[4];
  }
}

__global__ void k_prim_boxes(const DPrim* __restrict__ prims, int first, int n, float time_lo, float time_hi,
                             Box* __restrict__ boxes) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Box b;
  prim_box(prims[first + i], time_lo, time_hi, b);
  boxes[i] = b;
}

// One block: bounds of all centroids and the largest absolute coordinate, then pad every box.
__global__ void k_scene_bounds(Box* __restrict__ boxes, int n, float* __restrict__ out7) {
  __shared__ float smin[3][256], smax[3][256], sext[256];
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f}, ext = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    for (int a = 0; a < 3; ++a) {
      float c = 0.5f * (boxes[i].mn[a] + boxes[i].mx[a]);
      mn[a] = fminf(mn[a], c);
      mx[a] = fmaxf(mx[a], c);
      ext = fmaxf(ext, fmaxf(fabsf(boxes[i].mn[a]), fabsf(boxes[i].mx[a])));
    }
  }
  for (int a = 0; a < 3; ++a) { smin[a][threadIdx.x] = mn[a]; smax[a][threadIdx.x] = mx[a]; }
  sext[threadIdx.x] = ext;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      for (int a = 0; a < 3; ++a) {
        smin[a][threadIdx.x] = fminf(smin[a][threadIdx.x], smin[a][threadIdx.x + s]);
        smax[a][threadIdx.x] = fmaxf(smax[a][threadIdx.x], smax[a][threadIdx.x + s]);
      }
      sext[threadIdx.x] = fmaxf(sext[threadIdx.x], sext[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    for (int a = 0; a < 3; ++a) { out7[a] = smin[a][0]; out7[3 + a] = smax[a][0]; }
    out7[6] = sext[0];
  }
}

// Boxes are padded by 2^-18 of the group's largest coordinate: the FP32 slab test and the FP32 /
// FP64 primitive tests round differently, and a primitive must never be culled by its own box.
__global__ void k_pad_boxes(Box* __restrict__ boxes, int n, const float* __restrict__ bounds7) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float pad = bounds7[6] * 3.814697265625e-6f + 1e-30f;
  for (int a = 0; a < 3; ++a) { boxes[i].mn[a] -= pad; boxes[i].mx[a] += pad; }
}

__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {  // 10 bits -> every third bit
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__global__ void k_morton(const Box* __restrict__ boxes, int n, int n_pad, const float* __restrict__ bounds7,
                         unsigned long long* __restrict__ keys) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  if (i >= n) { keys[i] = ~0ull; return; }
  uint32_t q[3];
  for (int a = 0; a < 3; ++a) {
    float lo = bounds7[a], hi = bounds7[3 + a];
    float c = 0.5f * (boxes[i].mn[a] + boxes[i].mx[a]);
    float u = (hi > lo) ? (c - lo) / (hi - lo) : 0.f;
    q[a] = (uint32_t)fminf(fmaxf(u * 1024.f, 0.f), 1023.f);
  }
  uint32_t code = (expand_bits(q[0]) << 2) | (expand_bits(q[1]) << 1) | expand_bits(q[2]);
  keys[i] = ((unsigned long long)code << 32) | (unsigned long long)(uint32_t)i;
}

__global__ void k_bitonic(unsigned long long* __restrict__ keys, int n_pad, int j, int k) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  int ixj = i ^ j;
  if (ixj > i) {
    unsigned long long a = keys[i], b = keys[ixj];
    bool up = (i & k) == 0;
    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
  }
}

__device__ __forceinline__ int delta(const unsigned long long* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  return __clzll((long long)(keys[i] ^ keys[j]));  // keys are unique (the index is in the low word)
}

// Karras 2012, Algorithm 1 body: children and parent links of internal node i.
__global__ void k_radix_tree(const unsigned long long* __restrict__ keys, int n, int first, int node_base,
                             DBvhNode* __restrict__ nodes) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  int j = i + l * d;
  int dnode = delta(keys, n, i, j);
  int s = 0;
  int t = l;
  do {
    t = (t + 1) >> 1;
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  int gamma = i + s * d + min(d, 0);
  int lo = min(i, j), hi = max(i, j);
  DBvhNode* N = &nodes[node_base + i];
  // a child range of one key is a leaf: the primitive whose group-local index is the key's low word
  N->left = (lo == gamma) ? ~(first + (int)(uint32_t)keys[gamma]) : node_base + gamma;
  N->right = (hi == gamma + 1) ? ~(first + (int)(uint32_t)keys[gamma + 1]) : node_base + gamma + 1;
  N->pad[0] = N->pad[1] = 0;
}

// One level of the bottom-up fit. ready[i] = number of the pass that completed node i (0 = not
// yet); a child counts as ready only if an EARLIER launch completed it, so every box a node
// reads was written before the kernel boundary. height[i] = levels below node i.
__device__ __forceinline__ float half_area(const Box& b) {
  const float x = b.mx[0] - b.mn[0], y = b.mx[1] - b.mn[1], z = b.mx[2] - b.mn[2];
  return x * y + y * z + z * x;
}

__global__ void k_fit_pass(const Box* __restrict__ boxes, int n, int first, int node_base,
                           DBvhNode* __restrict__ nodes, int* __restrict__ ready, int* __restrict__ height,
                           int* __restrict__ count, int pass, int rotate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1 || ready[i] != 0) return;
  DBvhNode* N = &nodes[node_base + i];
  const int child[2] = {N->left, N->right};
  Box cb[2];
  int ch[2];
  for (int c = 0; c < 2; ++c) {
    if (child[c] < 0) {
      cb[c] = boxes[(~child[c]) - first];
      ch[c] = 0;
    } else {
      int ci = child[c] - node_base;
      int rp = ready[ci];
      if (rp == 0 || rp >= pass) return;
      const DBvhNode* C = &nodes[child[c]];
      for (int a = 0; a < 3; ++a) {
        cb[c].mn[a] = fminf(C->lmin[a], C->rmin[a]);
        cb[c].mx[a] = fmaxf(C->lmax[a], C->rmax[a]);
      }
      ch[c] = height[ci];
    }
  }
  int kid[2] = {child[0], child[1]};
  if (rotate) {
    // Tree rotations (Kensler 2008) on the way up: a Morton tree ignores box sizes, so a child is
    // often better paired with one of its NIECES. Swapping child c of this node with grandchild g
    // of the other child changes only that other child's box; the swap that shrinks it most is
    // taken. The other child is complete and referenced by this node alone, so rewriting it here
    // races with nothing; the pass structure (a node reads only what earlier launches wrote) holds
    // because a rewritten node is not read again in this launch.
    float best_gain = 0.f;
    int best_c = -1, best_g = -1;
    Box best_box = cb[0];
    for (int c = 0; c < 2; ++c) {
      const int o = 1 - c;            // the child that would be rebuilt
      if (kid[o] < 0) continue;       // a leaf has no children to trade
      const DBvhNode* O = &nodes[kid[o]];
      const float area_o = half_area(cb[o]);
      for (int g = 0; g < 2; ++g) {   // trade kid[c] for O's child g: O keeps child 1-g and gains kid[c]
        Box keep;
        for (int a = 0; a < 3; ++a) {
          keep.mn[a] = (1 - g) == 0 ? O->lmin[a] : O->rmin[a];
          keep.mx[a] = (1 - g) == 0 ? O->lmax[a] : O->rmax[a];
        }
        Box nb;
        for (int a = 0; a < 3; ++a) { nb.mn[a] = fminf(keep.mn[a], cb[c].mn[a]); nb.mx[a] = fmaxf(keep.mx[a], cb[c].mx[a]); }
        const float gain = area_o - half_area(nb);
        if (gain > best_gain) { best_gain = gain; best_c = c; best_g = g; best_box = nb; }
      }
    }
    if (best_c >= 0) {
      const int c = best_c, o = 1 - c, g = best_g;
      DBvhNode* O = &nodes[kid[o]];
      const int up = g == 0 ? O->left : O->right;     // the grandchild that comes up to this node
      Box upb;
      for (int a = 0; a < 3; ++a) { upb.mn[a] = g == 0 ? O->lmin[a] : O->rmin[a]; upb.mx[a] = g == 0 ? O->lmax[a] : O->rmax[a]; }
      // O's slot g takes this node's child c
      if (g == 0) { O->left = kid[c]; for (int a = 0; a < 3; ++a) { O->lmin[a] = cb[c].mn[a]; O->lmax[a] = cb[c].mx[a]; } }
      else { O->right = kid[c]; for (int a = 0; a < 3; ++a) { O->rmin[a] = cb[c].mn[a]; O->rmax[a] = cb[c].mx[a]; } }
      const int h_keep = (g == 0 ? O->right : O->left) < 0 ? 0 : height[(g == 0 ? O->right : O->left) - node_base];
      const int h_up = up < 0 ? 0 : height[up - node_base];
      height[kid[o] - node_base] = 1 + max(h_keep, ch[c]);
      {
        const int keep_ref = g == 0 ? O->right : O->left;
        count[kid[o] - node_base] = 1 + (keep_ref < 0 ? 0 : count[keep_ref - node_base]) + (kid[c] < 0 ? 0 : count[kid[c] - node_base]);
      }
      ch[o] = height[kid[o] - node_base];
      ch[c] = h_up;
      kid[c] = up;
      cb[c] = upb;
      cb[o] = best_box;
    }
  }
  N->left = kid[0]; N->right = kid[1];
  for (int a = 0; a < 3; ++a) {
    N->lmin[a] = cb[0].mn[a]; N->lmax[a] = cb[0].mx[a];
    N->rmin[a] = cb[1].mn[a]; N->rmax[a] = cb[1].mx[a];
  }
  height[i] = 1 + max(ch[0], ch[1]);
  // internal nodes in this subtree (a rotated child's count was refreshed above)
  count[i] = 1 + (kid[0] < 0 ? 0 : count[kid[0] - node_base]) + (kid[1] < 0 ? 0 : count[kid[1] - node_base]);
  ready[i] = pass;
}

// Depth-first numbering, one tree level per launch: a numbered node numbers its children (left child
// next to its parent, right child behind the left subtree), so a walk that goes down to the left
// reads consecutive records.
__global__ void k_number_pass(const DBvhNode* __restrict__ nodes, int n, int node_base, const int* __restrict__ count,
                              int* __restrict__ number, int* __restrict__ numbered, int pass) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int mark = numbered[i];
  if (mark == 0 || mark >= pass) return;     // not numbered before this launch
  if (mark < pass - 1) return;               // its children were numbered by an earlier launch
  const DBvhNode* N = &nodes[node_base + i];
  const int k = number[i];
  int next = k + 1;
  if (N->left >= 0) { const int c = N->left - node_base; number[c] = next; numbered[c] = pass; next += count[c]; }
  if (N->right >= 0) { const int c = N->right - node_base; number[c] = next; numbered[c] = pass; }
}
__global__ void k_renumber(const DBvhNode* __restrict__ nodes, DBvhNode* __restrict__ out, int n, int node_base,
                           const int* __restrict__ number) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  DBvhNode N = nodes[node_base + i];
  if (N.left >= 0) N.left = node_base + number[N.left - node_base];
  if (N.right >= 0) N.right = node_base + number[N.right - node_base];
  out[number[i]] = N;
}

__global__ void k_pack_nodes(const DBvhNode* __restrict__ nodes, const DPrim* __restrict__ prims, DBvhNode16* __restrict__ packed, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const DBvhNode N = nodes[i];
  DBvhNode16 P;
  // centre to the nearest half, then the half-extent that still covers both bounds FROM THAT CENTRE,
  // rounded up (the differences are exact in FP64): the packed box can only be larger
  auto pack = [](float mn, float mx, uint16_t& c16, uint16_t& h16) {
    const __half c = __float2half_rn(0.5f * mn + 0.5f * mx);
    const double cd = (double)__half2float(c);
    const double h = fmax((double)mx - cd, cd - (double)mn);
    c16 = __half_as_ushort(c);
    h16 = __half_as_ushort(__float2half_ru((float)(h * (1.0 + 1e-6))));
  };
  for (int a = 0; a < 3; ++a) {
    pack(N.lmin[a], N.lmax[a], P.lc[a], P.lh[a]);
    pack(N.rmin[a], N.rmax[a], P.rc[a], P.rh[a]);
  }
  // leaves: the primitive's type beside its index
  auto tag = [&](int child) { return child >= 0 ? child : ~(~child | ((prims[~child].type & 15) << RT_LEAF_TYPE_SHIFT)); };
  P.left = tag(N.left); P.right = tag(N.right);
  packed[i] = P;
}

}  // namespace

hipError_t pack_bvh_nodes(const DBvhNode* d_nodes, const DPrim* d_prims, DBvhNode16* d_packed, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_nodes, dim3((n + 255) / 256), dim3(256), 0, stream, d_nodes, d_prims, d_packed, n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(stream);
}

namespace {

}  // namespace

#define BVH_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return e_; } } while (0)

hipError_t build_bvh(const DPrim* d_prims, int first, int n, DBvhNode* d_nodes, int node_base, float time_lo,
                     float time_hi, int* height_out, hipStream_t stream) {
  if (n < 2) return hipErrorInvalidValue;
  int n_pad = 1;
  while (n_pad < n) n_pad <<= 1;
  Box* boxes = nullptr;
  float* bounds = nullptr;
  unsigned long long* keys = nullptr;
  int *ready = nullptr, *height = nullptr, *count = nullptr, *number = nullptr;
  DBvhNode* ordered = nullptr;
  auto cleanup = [&]() {
    hipFree(boxes); hipFree(bounds); hipFree(keys); hipFree(ready); hipFree(height); hipFree(count); hipFree(number); hipFree(ordered);
  };
  BVH_CHECK(hipMalloc(&boxes, sizeof(Box) * n));
  BVH_CHECK(hipMalloc(&bounds, sizeof(float) * 8));
  BVH_CHECK(hipMalloc(&keys, sizeof(unsigned long long) * n_pad));
  BVH_CHECK(hipMalloc(&ready, sizeof(int) * n));
  BVH_CHECK(hipMalloc(&height, sizeof(int) * n));
  BVH_CHECK(hipMalloc(&count, sizeof(int) * n));
  BVH_CHECK(hipMalloc(&number, sizeof(int) * n));
  BVH_CHECK(hipMalloc(&ordered, sizeof(DBvhNode) * n));
  BVH_CHECK(hipMemsetAsync(count, 0, sizeof(int) * n, stream));
  BVH_CHECK(hipMemsetAsync(ready, 0, sizeof(int) * n, stream));
  BVH_CHECK(hipMemsetAsync(height, 0, sizeof(int) * n, stream));
  const int B = 256;
  hipLaunchKernelGGL(k_prim_boxes, dim3((n + B - 1) / B), dim3(B), 0, stream, d_prims, first, n, time_lo, time_hi, boxes);
  hipLaunchKernelGGL(k_scene_bounds, dim3(1), dim3(256), 0, stream, boxes, n, bounds);
  hipLaunchKernelGGL(k_morton, dim3((n_pad + B - 1) / B), dim3(B), 0, stream, boxes, n, n_pad, bounds, keys);
  hipLaunchKernelGGL(k_pad_boxes, dim3((n + B - 1) / B), dim3(B), 0, stream, boxes, n, bounds);
  for (int k = 2; k <= n_pad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1)
      hipLaunchKernelGGL(k_bitonic, dim3((n_pad + B - 1) / B), dim3(B), 0, stream, keys, n_pad, j, k);
  hipLaunchKernelGGL(k_radix_tree, dim3((n - 1 + B - 1) / B), dim3(B), 0, stream, keys, n, first, node_base, d_nodes);
  BVH_CHECK(hipGetLastError());
  // bottom-up sweeps: the first fits the boxes of the radix tree, the following ones fit again while
  // trading children for nieces wherever that shrinks a box (RT_BVH_SWEEPS - 1 rotation sweeps)
  for (int sweep = 0; sweep < RT_BVH_SWEEPS; ++sweep) {
    if (sweep > 0) BVH_CHECK(hipMemsetAsync(ready, 0, sizeof(int) * n, stream));
    int root_ready = 0;
    int pass = 1;
    const int max_pass = n + 2;  // a binary tree over n leaves is at most n-1 levels deep
    while (!root_ready && pass <= max_pass) {
      for (int rep = 0; rep < 8 && pass <= max_pass; ++rep, ++pass)
        hipLaunchKernelGGL(k_fit_pass, dim3((n - 1 + B - 1) / B), dim3(B), 0, stream, boxes, n, first, node_base,
                           d_nodes, ready, height, count, pass, sweep > 0 ? 1 : 0);
      BVH_CHECK(hipMemcpyAsync(&root_ready, ready, sizeof(int), hipMemcpyDeviceToHost, stream));
      BVH_CHECK(hipStreamSynchronize(stream));
    }
    if (!root_ready) { cleanup(); return hipErrorUnknown; }
  }
  BVH_CHECK(hipMemcpyAsync(height_out, height, sizeof(int), hipMemcpyDeviceToHost, stream));
  BVH_CHECK(hipStreamSynchronize(stream));
  {  // depth-first order (the root keeps record 0): `ready` is reused as the numbering pass of each node
    const int one = 1, zero = 0;
    BVH_CHECK(hipMemsetAsync(ready, 0, sizeof(int) * n, stream));
    BVH_CHECK(hipMemcpyAsync(ready, &one, sizeof(int), hipMemcpyHostToDevice, stream));
    BVH_CHECK(hipMemcpyAsync(number, &zero, sizeof(int), hipMemcpyHostToDevice, stream));
    for (int pass = 2; pass <= *height_out + 2; ++pass)
      hipLaunchKernelGGL(k_number_pass, dim3((n - 1 + B - 1) / B), dim3(B), 0, stream, d_nodes, n, node_base, count, number, ready, pass);
    hipLaunchKernelGGL(k_renumber, dim3((n - 1 + B - 1) / B), dim3(B), 0, stream, d_nodes, ordered, n, node_base, number);
    BVH_CHECK(hipGetLastError());
    BVH_CHECK(hipMemcpyAsync(d_nodes + node_base, ordered, sizeof(DBvhNode) * (size_t)(n - 1), hipMemcpyDeviceToDevice, stream));
    BVH_CHECK(hipStreamSynchronize(stream));
  }
  cleanup();
  return hipSuccess;
}

// ---------------------------------------------------------------------------------------------
// Top-down construction by the surface-area heuristic ON THE DEVICE, exact sweep. Every candidate
// plane between two centroids, on every axis, is priced as area(left) * count(left) + area(right) *
// count(right); one primitive per leaf. The primitives are sorted by centroid once per axis; a node
// owns the same segment [lo, hi) of all three orders, and splitting it is a stable partition of the
// three segments, which keeps each sorted (Wald 2007). One launch per tree level, one workgroup per
// open node: suffix scan of boxes for the right-hand areas, prefix scan for the left-hand ones and
// the cost, a lexicographic minimum over (cost, axis, plane), then the partition. Box unions are
// min/max and therefore exactly associative, and areas and costs use the same rounded operations
// in the same order as build_bvh_sah below, so both give the same tree, record for record. A node's
// record number is its place in depth-first order, computed on the way down (the left child follows
// its parent, the right child follows the left subtree of mid - lo - 1 records).
namespace {

constexpr int SAH_B = 256;
struct SahTask { int lo, hi, me, depth; };

__device__ __forceinline__ Box empty_box() { return Box{{3.4e38f, 3.4e38f, 3.4e38f}, {-3.4e38f, -3.4e38f, -3.4e38f}}; }
__device__ __forceinline__ void grow(Box& b, const Box& o) {
  for (int a = 0; a < 3; ++a) { b.mn[a] = fminf(b.mn[a], o.mn[a]); b.mx[a] = fmaxf(b.mx[a], o.mx[a]); }
}
// Bits of a float in an order that sorts as the floats do (-0 counts as +0).
__device__ __forceinline__ uint32_t float_order(float f) {
  const uint32_t u = __float_as_uint(f + 0.0f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ void k_centroid_keys(const Box* __restrict__ boxes, int n, int n_pad, int axis, unsigned long long* __restrict__ keys) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  if (i >= n) { keys[i] = ~0ull; return; }
  const float c = 0.5f * (boxes[i].mn[axis] + boxes[i].mx[axis]);
  keys[i] = ((unsigned long long)float_order(c) << 32) | (unsigned long long)(uint32_t)i;
}
__global__ void k_keys_to_order(const unsigned long long* __restrict__ keys, int n, int* __restrict__ order) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) order[i] = (int)(uint32_t)keys[i];
}

// Inclusive scans over the workgroup (Hillis-Steele in LDS); sh[SAH_B - 1] holds the total afterwards.
__device__ __forceinline__ Box scan_boxes(Box v, Box* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int off = 1; off < SAH_B; off <<= 1) {
    Box o = v;
    if (t >= off) o = sh[t - off];
    __syncthreads();
    if (t >= off) { grow(v, o); sh[t] = v; }
    __syncthreads();
  }
  return v;
}
__device__ __forceinline__ int scan_ints(int v, int* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int off = 1; off < SAH_B; off <<= 1) {
    int o = 0;
    if (t >= off) o = sh[t - off];
    __syncthreads();
    if (t >= off) { v += o; sh[t] = v; }
    __syncthreads();
  }
  return v;
}

struct SahPick { float cost; int axis, split; };
__device__ __forceinline__ bool better(const SahPick& a, const SahPick& b) {  // a before b?
  if (a.axis < 0) return false;
  if (b.axis < 0) return true;
  if (a.cost != b.cost) return a.cost < b.cost;
  if (a.axis != b.axis) return a.axis < b.axis;
  return a.split < b.split;
}

__global__ void __launch_bounds__(SAH_B)
k_sah_level(const Box* __restrict__ boxes, int n, int first, int node_base, const SahTask* __restrict__ tasks,
            const int* order_in, int* order_out, float* right_area, unsigned char* side,
            DBvhNode* __restrict__ nodes, SahTask* __restrict__ next_tasks, int* __restrict__ counters) {
  __shared__ Box sh_box[SAH_B];
  __shared__ int sh_int[SAH_B];
  __shared__ SahPick sh_pick[SAH_B];
  const SahTask T = tasks[blockIdx.x];
  const int lo = T.lo, hi = T.hi, m = hi - lo, t = threadIdx.x;
  SahPick mine{3.4e38f, -1, lo + m / 2};
  for (int a = 0; a < 3; ++a) {
    const int* ord = order_in + (size_t)a * n;
    // areas of the right-hand sides: union of positions [e, hi), scanned from the end
    Box carry = empty_box();
    for (int c0 = 0; c0 < m; c0 += SAH_B) {
      const int e = hi - 1 - (c0 + t);
      Box v = e >= lo ? boxes[ord[e]] : empty_box();
      v = scan_boxes(v, sh_box);
      grow(v, carry);
      if (e > lo) right_area[e] = half_area(v);
      grow(carry, sh_box[SAH_B - 1]);
      __syncthreads();
    }
    // left-hand sides: union of positions [lo, e], the plane after e is split e + 1
    carry = empty_box();
    for (int c0 = 0; c0 < m; c0 += SAH_B) {
      const int e = lo + c0 + t;
      Box v = e < hi ? boxes[ord[e]] : empty_box();
      v = scan_boxes(v, sh_box);
      grow(v, carry);
      if (e + 1 < hi) {
        const int i = e + 1;
        const float cost = half_area(v) * (float)(i - lo) + right_area[i] * (float)(hi - i);
        if (cost < mine.cost) { mine.cost = cost; mine.axis = a; mine.split = i; }
      }
      grow(carry, sh_box[SAH_B - 1]);
      __syncthreads();
    }
  }
  sh_pick[t] = mine;
  __syncthreads();
  for (int s = SAH_B / 2; s > 0; s >>= 1) {
    if (t < s && better(sh_pick[t + s], sh_pick[t])) sh_pick[t] = sh_pick[t + s];
    __syncthreads();
  }
  const SahPick best = sh_pick[0];
  const int axis = best.axis < 0 ? 0 : best.axis;
  // past 28 levels the rest is split in halves: the traversal's per-lane stack is sized by the height
  const int mid = (T.depth >= 28 || best.axis < 0) ? lo + m / 2 : best.split;
  const int* win = order_in + (size_t)axis * n;
  Box lb = empty_box(), rb = empty_box();
  for (int e = lo + t; e < hi; e += SAH_B) {
    const int p = win[e];
    side[p] = e >= mid;
    if (e < mid) grow(lb, boxes[p]); else grow(rb, boxes[p]);
  }
  lb = scan_boxes(lb, sh_box);
  lb = sh_box[SAH_B - 1];
  __syncthreads();
  rb = scan_boxes(rb, sh_box);
  rb = sh_box[SAH_B - 1];
  __syncthreads();  // also: every side[] flag of this segment is written
  for (int a = 0; a < 3; ++a) {
    const int* ord = order_in + (size_t)a * n;
    int* out = order_out + (size_t)a * n;
    int lefts = 0;  // left-hand primitives before this chunk
    for (int c0 = 0; c0 < m; c0 += SAH_B) {
      const int e = lo + c0 + t;
      const int p = e < hi ? ord[e] : 0;
      const int is_left = e < hi ? !side[p] : 0;
      const int incl = scan_ints(is_left, sh_int);
      const int before = lefts + incl - is_left;
      if (e < hi) out[is_left ? lo + before : mid + (e - lo) - before] = p;
      lefts += sh_int[SAH_B - 1];
      __syncthreads();
    }
  }
  if (t == 0) {
    DBvhNode N;
    for (int a = 0; a < 3; ++a) { N.lmin[a] = lb.mn[a]; N.lmax[a] = lb.mx[a]; N.rmin[a] = rb.mn[a]; N.rmax[a] = rb.mx[a]; }
    const int left_me = T.me + 1, right_me = T.me + (mid - lo);
    N.left = (mid - lo == 1) ? ~(first + win[lo]) : node_base + left_me;
    N.right = (hi - mid == 1) ? ~(first + win[mid]) : node_base + right_me;
    N.pad[0] = N.pad[1] = 0;
    nodes[node_base + T.me] = N;
    if (mid - lo >= 2) next_tasks[atomicAdd(&counters[0], 1)] = SahTask{lo, mid, left_me, T.depth + 1};
    if (hi - mid >= 2) next_tasks[atomicAdd(&counters[0], 1)] = SahTask{mid, hi, right_me, T.depth + 1};
    atomicMax(&counters[1], T.depth + 1);
  }
}

}  // namespace

hipError_t build_bvh_sweep(const DPrim* d_prims, int first, int n, DBvhNode* d_nodes, int node_base, float time_lo,
                           float time_hi, int* height_out, hipStream_t stream) {
  if (n < 2) return hipErrorInvalidValue;
  int n_pad = 1;
  while (n_pad < n) n_pad <<= 1;
  Box* boxes = nullptr;
  float *bounds = nullptr, *right_area = nullptr;
  unsigned long long* keys = nullptr;
  int *order[2] = {nullptr, nullptr}, *counters = nullptr;
  unsigned char* side = nullptr;
  SahTask* tasks[2] = {nullptr, nullptr};
  auto cleanup = [&]() {
    hipFree(boxes); hipFree(bounds); hipFree(right_area); hipFree(keys); hipFree(order[0]); hipFree(order[1]);
    hipFree(counters); hipFree(side); hipFree(tasks[0]); hipFree(tasks[1]);
  };
  BVH_CHECK(hipMalloc(&boxes, sizeof(Box) * n));
  BVH_CHECK(hipMalloc(&bounds, sizeof(float) * 8));
  BVH_CHECK(hipMalloc(&right_area, sizeof(float) * n));
  BVH_CHECK(hipMalloc(&keys, sizeof(unsigned long long) * n_pad));
  BVH_CHECK(hipMalloc(&order[0], sizeof(int) * 3 * (size_t)n));
  BVH_CHECK(hipMalloc(&order[1], sizeof(int) * 3 * (size_t)n));
  BVH_CHECK(hipMalloc(&counters, sizeof(int) * 2));
  BVH_CHECK(hipMalloc(&side, (size_t)n));
  BVH_CHECK(hipMalloc(&tasks[0], sizeof(SahTask) * n));
  BVH_CHECK(hipMalloc(&tasks[1], sizeof(SahTask) * n));
  const int B = 256;
  hipLaunchKernelGGL(k_prim_boxes, dim3((n + B - 1) / B), dim3(B), 0, stream, d_prims, first, n, time_lo, time_hi, boxes);
  hipLaunchKernelGGL(k_scene_bounds, dim3(1), dim3(256), 0, stream, boxes, n, bounds);
  hipLaunchKernelGGL(k_pad_boxes, dim3((n + B - 1) / B), dim3(B), 0, stream, boxes, n, bounds);
  for (int a = 0; a < 3; ++a) {
    hipLaunchKernelGGL(k_centroid_keys, dim3((n_pad + B - 1) / B), dim3(B), 0, stream, boxes, n, n_pad, a, keys);
    for (int k = 2; k <= n_pad; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1)
        hipLaunchKernelGGL(k_bitonic, dim3((n_pad + B - 1) / B), dim3(B), 0, stream, keys, n_pad, j, k);
    hipLaunchKernelGGL(k_keys_to_order, dim3((n + B - 1) / B), dim3(B), 0, stream, keys, n, order[0] + (size_t)a * n);
  }
  BVH_CHECK(hipGetLastError());
  const SahTask root{0, n, 0, 0};
  BVH_CHECK(hipMemcpyAsync(tasks[0], &root, sizeof(root), hipMemcpyHostToDevice, stream));
  int open = 1, cur = 0, height = 0;
  for (int level = 0; open > 0; ++level) {
    if (level > 28 + 40) { cleanup(); return hipErrorUnknown; }  // halving from level 28 on ends within 32 more
    BVH_CHECK(hipMemsetAsync(counters, 0, sizeof(int) * (level == 0 ? 2 : 1), stream));  // [0] open nodes, [1] height
    hipLaunchKernelGGL(k_sah_level, dim3(open), dim3(SAH_B), 0, stream, boxes, n, first, node_base, tasks[cur], order[cur],
                       order[cur ^ 1], right_area, side, d_nodes, tasks[cur ^ 1], counters);
    BVH_CHECK(hipGetLastError());
    int back[2] = {0, 0};
    BVH_CHECK(hipMemcpyAsync(back, counters, sizeof(back), hipMemcpyDeviceToHost, stream));
    BVH_CHECK(hipStreamSynchronize(stream));
    open = back[0];
    height = back[1];
    if (open < 0 || open > n) { cleanup(); return hipErrorUnknown; }
    cur ^= 1;
  }
  *height_out = height;
  cleanup();
  return hipSuccess;
}

// ---------------------------------------------------------------------------------------------
// Host construction by the surface-area heuristic, for groups small enough that a top-down build
// is instant. A ray's walk is as long as the boxes it pierces, and the Morton tree above ignores
// box sizes altogether; every candidate plane between two centroids, on every axis, is priced as
// area(left) * count(left) + area(right) * count(right) (exact sweep, one primitive per leaf).
// Nodes are laid out in depth-first order: a node's left child is the next record.
namespace {

struct HBox {
  float mn[3], mx[3];
  void grow(const HBox& o) {
    for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], o.mn[a]); mx[a] = std::max(mx[a], o.mx[a]); }
  }
  float half_area() const {
    const float x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2];
    return x * y + y * z + z * x;
  }
  static HBox empty() { return HBox{{3.4e38f, 3.4e38f, 3.4e38f}, {-3.4e38f, -3.4e38f, -3.4e38f}}; }
};

struct SahBuilder {
  const std::vector<HBox>& boxes;
  int first, node_base;
  std::vector<DBvhNode>& nodes;  // n - 1 records, appended in depth-first order
  std::vector<int> idx, tmp;
  std::vector<float> right_area;
  int max_height = 0;

  HBox bounds(int lo, int hi) const {
    HBox b = HBox::empty();
    for (int i = lo; i < hi; ++i) b.grow(boxes[idx[i]]);
    return b;
  }
  float centroid(int i, int a) const { return 0.5f * (boxes[i].mn[a] + boxes[i].mx[a]); }

  // Builds the subtree over idx[lo, hi) (hi - lo >= 2), returns its node index; depth = levels above.
  int build(int lo, int hi, int depth) {
    const int n = hi - lo;
    const int me = (int)nodes.size();
    nodes.push_back(DBvhNode{});
    max_height = std::max(max_height, depth + 1);
    int best_axis = -1, best_split = lo + n / 2;
    float best_cost = 3.4e38f;
    for (int a = 0; a < 3; ++a) {
      std::sort(idx.begin() + lo, idx.begin() + hi, [&](int x, int y) {
        const float cx = centroid(x, a), cy = centroid(y, a);
        return cx < cy || (cx == cy && x < y);
      });
      HBox r = HBox::empty();
      for (int i = hi - 1; i > lo; --i) { r.grow(boxes[idx[i]]); right_area[i] = r.half_area(); }
      HBox l = HBox::empty();
      for (int i = lo + 1; i < hi; ++i) {  // left = [lo, i), right = [i, hi)
        l.grow(boxes[idx[i - 1]]);
        const float cost = l.half_area() * (float)(i - lo) + right_area[i] * (float)(hi - i);
        if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = i; }
      }
    }
    if (best_axis != 2) {  // idx is sorted along axis 2 now: redo the winner's order
      const int a = best_axis < 0 ? 0 : best_axis;
      std::sort(idx.begin() + lo, idx.begin() + hi, [&](int x, int y) {
        const float cx = centroid(x, a), cy = centroid(y, a);
        return cx < cy || (cx == cy && x < y);
      });
    }
    // Past 28 levels the rest is split in halves: the per-lane stack (LDS) is sized by the height.
    const int mid = depth >= 28 ? lo + n / 2 : best_split;
    const HBox lb = bounds(lo, mid), rb = bounds(mid, hi);
    const int left = (mid - lo == 1) ? ~(first + idx[lo]) : build(lo, mid, depth + 1);
    const int right = (hi - mid == 1) ? ~(first + idx[mid]) : build(mid, hi, depth + 1);
    DBvhNode& N = nodes[me];
    for (int a = 0; a < 3; ++a) { N.lmin[a] = lb.mn[a]; N.lmax[a] = lb.mx[a]; N.rmin[a] = rb.mn[a]; N.rmax[a] = rb.mx[a]; }
    N.left = left < 0 ? left : node_base + left;
    N.right = right < 0 ? right : node_base + right;
    N.pad[0] = N.pad[1] = 0;
    return me;
  }
};

}  // namespace

hipError_t build_bvh_sah(const DPrim* h_prims, int first, int n, DBvhNode* d_nodes, int node_base, float time_lo,
                         float time_hi, int* height_out, hipStream_t stream) {
  if (n < 2) return hipErrorInvalidValue;
  std::vector<HBox> boxes((size_t)n);
  float ext = 0.f;
  for (int i = 0; i < n; ++i) {
    Box b;
    prim_box(h_prims[first + i], time_lo, time_hi, b);
    for (int a = 0; a < 3; ++a) {
      boxes[i].mn[a] = b.mn[a]; boxes[i].mx[a] = b.mx[a];
      ext = std::max(ext, std::max(std::fabs(b.mn[a]), std::fabs(b.mx[a])));
    }
  }
  const float pad = ext * 3.814697265625e-6f + 1e-30f;  // as k_pad_boxes
  for (auto& b : boxes)
    for (int a = 0; a < 3; ++a) { b.mn[a] -= pad; b.mx[a] += pad; }
  std::vector<DBvhNode> nodes;
  nodes.reserve((size_t)n - 1);
  SahBuilder B{boxes, first, node_base, nodes, {}, {}, {}, 0};
  B.idx.resize((size_t)n);
  std::iota(B.idx.begin(), B.idx.end(), 0);
  B.right_area.resize((size_t)n + 1);
  B.build(0, n, 0);
  if ((int)nodes.size() != n - 1) return hipErrorUnknown;
  *height_out = B.max_height;
  hipError_t e = hipMemcpyAsync(d_nodes + node_base, nodes.data(), sizeof(DBvhNode) * nodes.size(), hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(stream);
}

}  // namespace rt
