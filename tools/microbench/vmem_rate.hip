// What the vector memory pipe charges per wave instruction for the ways a tree-walk step could fetch its 32-byte node:
// per-lane random records of a small table (L1-resident, as the walk's nodes are), 8 waves per SIMD, a handful of VALU
// instructions per round so that the pipe is what is measured. Typed buffer loads convert FP16 fields to FP32
// registers in the texture unit; the question is what they cost against the plain 16-byte load.
// Build: hipcc -O3 --offload-arch=gfx950 vmem_rate.hip -o vmem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

#define KERNEL(NAME, LOADS)                                                                          \
  __global__ void NAME(const uint32_t* table, uint32_t mask, uint32_t* out, int iters) {                    \
    u4 rsrc;                                                                                                \
    rsrc.x = (uint32_t)(uintptr_t)table;                                                                    \
    rsrc.y = (uint32_t)((uintptr_t)table >> 32);                                                            \
    rsrc.z = (mask + 1u) * 32u;                                                                             \
    rsrc.w = 0x00027FACu; /* dst_sel xyzw, 32-bit float: typed loads name their own format */               \
    rsrc.x = __builtin_amdgcn_readfirstlane(rsrc.x); rsrc.y = __builtin_amdgcn_readfirstlane(rsrc.y);       \
    rsrc.z = __builtin_amdgcn_readfirstlane(rsrc.z); rsrc.w = __builtin_amdgcn_readfirstlane(rsrc.w);       \
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u, acc = 0;                             \
    const uint64_t base = (uint64_t)(uintptr_t)table;                                                       \
    for (int i = 0; i < iters; ++i) {                                                                       \
      x = (x ^ (x >> 7)) + 0x9E3779B9u;                                                                     \
      const uint32_t off = (x & mask) << 5;                                                                 \
      const uint64_t addr = base + off;                                                                     \
      uint32_t first, last;                                                                                 \
      asm volatile(LOADS                                                                                    \
                   : "=&v"(first), "=&v"(last)                                                              \
                   : "v"(off), "v"(addr), "s"(rsrc)                                                         \
                   : "memory", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73"); \
      acc ^= first ^ last;                                                                                  \
      x += acc & 1u;                                                                                        \
    }                                                                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                                       \
  }

// %2 = byte offset (VGPR), %3 = 64-bit address (VGPR pair), %4 = buffer resource (SGPR quad); results into consecutive registers
// is not required here: each load names its own destination registers through the operand list
KERNEL(k_global_x4, "global_load_dwordx4 v[60:63], %3, off\n s_waitcnt vmcnt(0)\n v_mov_b32 %0, v60\n v_mov_b32 %1, v63\n")
KERNEL(k_global_x4_x4, "global_load_dwordx4 v[60:63], %3, off\n global_load_dwordx4 v[64:67], %3, off offset:16\n s_waitcnt vmcnt(0)\n v_mov_b32 %0, v60\n v_mov_b32 %1, v67\n")
KERNEL(k_buffer_x4, "buffer_load_dwordx4 v[60:63], %2, %4, 0 offen\n s_waitcnt vmcnt(0)\n v_mov_b32 %0, v60\n v_mov_b32 %1, v63\n")
KERNEL(k_buffer_x4_x4, "buffer_load_dwordx4 v[60:63], %2, %4, 0 offen\n buffer_load_dwordx4 v[64:67], %2, %4, 0 offen offset:16\n s_waitcnt vmcnt(0)\n v_mov_b32 %0, v60\n v_mov_b32 %1, v67\n")
KERNEL(k_buffer_x2_4, "buffer_load_dwordx2 v[60:61], %2, %4, 0 offen\n buffer_load_dwordx2 v[62:63], %2, %4, 0 offen offset:8\n buffer_load_dwordx2 v[64:65], %2, %4, 0 offen offset:16\n buffer_load_dwordx2 v[66:67], %2, %4, 0 offen offset:24\n s_waitcnt vmcnt(0)\n v_mov_b32 %0, v60\n v_mov_b32 %1, v67\n")
KERNEL(k_typed_1, "tbuffer_load_format_xyzw v[60:63], %2, %4, 0 format:[BUF_DATA_FORMAT_16_16_16_16,BUF_NUM_FORMAT_FLOAT] offen\n s_waitcnt vmcnt(0)\n v_mov_b32 %0, v60\n v_mov_b32 %1, v63\n")
KERNEL(k_typed_3_x2, "tbuffer_load_format_xyzw v[60:63], %2, %4, 0 format:[BUF_DATA_FORMAT_16_16_16_16,BUF_NUM_FORMAT_FLOAT] offen\n tbuffer_load_format_xyzw v[64:67], %2, %4, 0 format:[BUF_DATA_FORMAT_16_16_16_16,BUF_NUM_FORMAT_FLOAT] offen offset:8\n tbuffer_load_format_xyzw v[68:71], %2, %4, 0 format:[BUF_DATA_FORMAT_16_16_16_16,BUF_NUM_FORMAT_FLOAT] offen offset:16\n buffer_load_dwordx2 v[72:73], %2, %4, 0 offen offset:24\n s_waitcnt vmcnt(0)\n v_mov_b32 %0, v60\n v_mov_b32 %1, v73\n")
KERNEL(k_typed_xy_6_x2, "tbuffer_load_format_xy v[60:61], %2, %4, 0 format:[BUF_DATA_FORMAT_16_16,BUF_NUM_FORMAT_FLOAT] offen\n tbuffer_load_format_xy v[62:63], %2, %4, 0 format:[BUF_DATA_FORMAT_16_16,BUF_NUM_FORMAT_FLOAT] offen offset:4\n tbuffer_load_format_xy v[64:65], %2, %4, 0 format:[BUF_DATA_FORMAT_16_16,BUF_NUM_FORMAT_FLOAT] offen offset:8\n tbuffer_load_format_xy v[66:67], %2, %4, 0 format:[BUF_DATA_FORMAT_16_16,BUF_NUM_FORMAT_FLOAT] offen offset:12\n tbuffer_load_format_xy v[68:69], %2, %4, 0 format:[BUF_DATA_FORMAT_16_16,BUF_NUM_FORMAT_FLOAT] offen offset:16\n tbuffer_load_format_xy v[70:71], %2, %4, 0 format:[BUF_DATA_FORMAT_16_16,BUF_NUM_FORMAT_FLOAT] offen offset:20\n buffer_load_dwordx2 v[72:73], %2, %4, 0 offen offset:24\n s_waitcnt vmcnt(0)\n v_mov_b32 %0, v60\n v_mov_b32 %1, v73\n")

typedef void (*kern_t)(const uint32_t*, uint32_t, uint32_t*, int);

static void run(const char* name, kern_t kern, int loads, int bytes, const uint32_t* table, uint32_t mask, uint32_t* out, int cus, double khz) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, waves_per_simd = 8;
  const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves: one per SIMD
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, table, mask, out, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, table, mask, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double cycles = ms * 1e-3 * khz * 1e3;
  // wave-rounds a CU serves: 4 SIMDs x waves_per_simd waves x iters
  const double per_round = cycles / ((double)iters * 4 * waves_per_simd);
  printf("%-18s %d loads, %2d bytes per lane and round: %7.3f ms  %6.1f CU cycles per wave-round  (%5.1f per load)\n", name, loads, bytes, ms,
         per_round, per_round / loads);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const uint32_t n_nodes = 512;  // 16 KiB of 32-byte records
  std::vector<uint32_t> h(n_nodes * 8);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3C003C00u + (uint32_t)i;  // halves near 1.0
  uint32_t *table, *out;
  hipMalloc(&table, h.size() * 4);
  hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&out, sizeof(uint32_t) * p.multiProcessorCount * 8 * 256);
#define R(K, L, B) run(#K, K, L, B, table, n_nodes - 1, out, p.multiProcessorCount, p.clockRate);
  R(k_global_x4, 1, 16) R(k_global_x4_x4, 2, 32) R(k_buffer_x4, 1, 16) R(k_buffer_x4_x4, 2, 32) R(k_buffer_x2_4, 4, 32)
  R(k_typed_1, 1, 8) R(k_typed_3_x2, 4, 32) R(k_typed_xy_6_x2, 7, 32)
  hipFree(table); hipFree(out);
  return 0;
}
