// Issue cost of individual gfx950 VALU instructions, 8 waves per SIMD, independent operands.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate2.hip -o valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X

#define KERNEL(NAME, BODY)                                                        \
  __global__ void NAME(float* out, int iters, float seed) {                       \
    float a = seed + threadIdx.x, b = a + 1.5f, c = a + 2.5f, d = a + 3.5f;       \
    float e = a * 0.5f, f = b * 0.5f, g = c * 0.5f, h = d * 0.5f;                 \
    double da = a, db = b, dc = c, dd = d;                                        \
    unsigned ua = threadIdx.x * 2654435761u + 1u, ub = ua ^ 0x9E3779B9u, uc = ua + 77u, ud = ub + 99u; \
    unsigned long long la = ua, lb = ub;                                          \
    for (int i = 0; i < iters; ++i) {                                             \
      asm volatile(BODY                                                           \
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(da), "+v"(db), \
                     "+v"(dc), "+v"(dd), "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud), "+v"(la), "+v"(lb)              \
                   :                                                              \
                   : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");                     \
    }                                                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] =                                  \
        a + b + c + d + e + f + g + h + (float)(da + db + dc + dd) + (float)(ua ^ ub ^ uc ^ ud) + (float)(la ^ lb); \
  }

// operands: %0-%7 floats a-h, %8-%11 doubles, %12-%15 uints, %16-%17 u64
KERNEL(k_fma, REP8("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %6, %6, %7, %1\n v_fma_f32 %2, %2, %4, %7\n"))
KERNEL(k_max, REP8("v_max_f32 %0, %0, %1\n v_max_f32 %3, %3, %4\n v_min_f32 %6, %6, %7\n v_min_f32 %2, %2, %5\n"))
KERNEL(k_max3, REP8("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %3, %3, %4, %5\n v_min3_f32 %6, %6, %7, %1\n v_min3_f32 %2, %2, %5, %4\n"))
KERNEL(k_cmp_vcc, REP8("v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %3, %4\n v_cmp_gt_f32 vcc, %6, %7\n v_cmp_gt_f32 vcc, %2, %5\n"))
KERNEL(k_cmp_sgpr, REP8("v_cmp_gt_f32 s[20:21], %0, %1\n v_cmp_gt_f32 s[22:23], %3, %4\n v_cmp_gt_f32 s[24:25], %6, %7\n v_cmp_gt_f32 s[26:27], %2, %5\n"))
KERNEL(k_cndmask_fixed, REP8("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %2, %2, %5, vcc\n"))
KERNEL(k_cmp_cnd, REP8("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %3, %3, %4, vcc\n v_cmp_gt_f32 vcc, %6, %7\n v_cndmask_b32 %2, %2, %5, vcc\n"))
KERNEL(k_cmp_cnd_far, REP8("v_cmp_gt_f32 s[20:21], %0, %1\n v_cmp_gt_f32 s[22:23], %6, %7\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %6, %6, %7, %7\n v_cndmask_b32 %3, %3, %4, s[20:21]\n v_cndmask_b32 %2, %2, %5, s[22:23]\n v_fma_f32 %4, %4, %1, %1\n v_fma_f32 %5, %5, %7, %7\n"))
KERNEL(k_cmp_sand, REP8("v_cmp_gt_f32 s[20:21], %0, %1\n v_cmp_gt_f32 s[22:23], %6, %7\n s_and_b64 s[24:25], s[20:21], s[22:23]\n v_cndmask_b32 %3, %3, %4, s[24:25]\n"))
KERNEL(k_mul_lo, REP8("v_mul_lo_u32 %12, %12, %13\n v_mul_lo_u32 %14, %14, %15\n v_mul_lo_u32 %13, %13, %15\n v_mul_lo_u32 %15, %15, %12\n"))
KERNEL(k_mul_hi, REP8("v_mul_hi_u32 %12, %12, %13\n v_mul_hi_u32 %14, %14, %15\n v_mul_hi_u32 %13, %13, %15\n v_mul_hi_u32 %15, %15, %12\n"))
KERNEL(k_mad_u64, REP8("v_mad_u64_u32 %16, vcc, %12, %13, %16\n v_mad_u64_u32 %17, vcc, %14, %15, %17\n v_mad_u64_u32 %16, vcc, %13, %15, %16\n v_mad_u64_u32 %17, vcc, %12, %14, %17\n"))
KERNEL(k_mul_u24, REP8("v_mul_u32_u24 %12, %12, %13\n v_mul_u32_u24 %14, %14, %15\n v_mul_hi_u32_u24 %13, %13, %15\n v_mul_hi_u32_u24 %15, %15, %12\n"))
KERNEL(k_mad_u32_u24, REP8("v_mad_u32_u24 %12, %12, %13, %14\n v_mad_u32_u24 %14, %14, %15, %12\n v_mad_u32_u24 %13, %13, %15, %12\n v_mad_u32_u24 %15, %15, %12, %13\n"))
KERNEL(k_xor, REP8("v_xor_b32 %12, %12, %13\n v_xor_b32 %14, %14, %15\n v_add_u32 %13, %13, %15\n v_lshrrev_b32 %15, 3, %12\n"))
KERNEL(k_xor3, REP8("v_xad_u32 %12, %12, %13, %14\n v_lshl_add_u32 %14, %14, 3, %12\n v_add3_u32 %13, %13, %15, %12\n v_and_or_b32 %15, %15, %12, %13\n"))
KERNEL(k_fma64, REP8("v_fma_f64 %8, %8, %9, %10\n v_fma_f64 %11, %11, %9, %10\n v_fma_f64 %9, %9, %8, %10\n v_fma_f64 %10, %10, %8, %11\n"))
KERNEL(k_add64, REP8("v_add_f64 %8, %8, %9\n v_add_f64 %11, %11, %9\n v_mul_f64 %9, %9, %8\n v_mul_f64 %10, %10, %8\n"))
KERNEL(k_rcp, REP8("v_rcp_f32 %0, %0\n v_rcp_f32 %3, %3\n v_rsq_f32 %6, %6\n v_sqrt_f32 %2, %2\n"))
KERNEL(k_rcp64, REP8("v_rcp_f64 %8, %8\n v_rsq_f64 %11, %11\n v_rcp_f64 %9, %9\n v_sqrt_f64 %10, %10\n"))
KERNEL(k_cvt, REP8("v_cvt_f64_f32 %8, %0\n v_cvt_f32_f64 %3, %9\n v_cvt_f32_u32 %6, %12\n v_cvt_u32_f32 %13, %2\n"))
KERNEL(k_pk_fma, REP8("v_pk_fma_f32 %8, %8, %9, %10\n v_pk_fma_f32 %11, %11, %9, %10\n v_pk_mul_f32 %9, %9, %8\n v_pk_add_f32 %10, %10, %8\n"))
KERNEL(k_cmpx, REP8("v_cmpx_gt_f32 %0, %1\n v_cmpx_gt_f32 %3, %4\n v_cmpx_gt_f32 %6, %7\n v_cmpx_gt_f32 %2, %5\n") "s_mov_b64 exec, -1\n")
KERNEL(k_readlane, REP8("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n"))
KERNEL(k_salu, REP8("s_add_u32 s20, s20, s21\n s_and_b64 s[22:23], s[22:23], s[24:25]\n s_xor_b32 s21, s21, s20\n s_or_b64 s[24:25], s[24:25], s[26:27]\n"))
KERNEL(k_salu_valu, REP8("s_add_u32 s20, s20, s21\n v_fma_f32 %0, %0, %1, %2\n s_xor_b32 s21, s21, s20\n v_fma_f32 %3, %3, %4, %5\n"))
KERNEL(k_fma_sgpr, REP8("v_fma_f32 %0, %0, s20, %2\n v_fma_f32 %3, %3, s21, %5\n v_fma_f32 %6, %6, s22, %1\n v_fma_f32 %2, %2, s23, %7\n"))
KERNEL(k_dpp, REP8("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %5 row_bcast:15 row_mask:0xf bank_mask:0xf\n"))

KERNEL(k_cndmask_indep, REP8("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n v_cndmask_b32 %6, %7, %1, vcc\n v_cndmask_b32 %0, %4, %7, vcc\n"))
KERNEL(k_cndmask_sgpr, REP8("v_cndmask_b32 %0, %1, %2, s[20:21]\n v_cndmask_b32 %3, %4, %5, s[22:23]\n v_cndmask_b32 %6, %7, %1, s[24:25]\n v_cndmask_b32 %0, %4, %7, s[26:27]\n"))
KERNEL(k_add, REP8("v_add_f32 %0, %0, %1\n v_mul_f32 %3, %3, %4\n v_sub_f32 %6, %6, %7\n v_mul_f32 %2, %2, %5\n"))
KERNEL(k_add_e64, REP8("v_add_f32_e64 %0, %0, -%1\n v_mul_f32_e64 %3, |%3|, %4\n v_sub_f32_e64 %6, %6, |%7|\n v_mul_f32_e64 %2, %2, -%5\n"))
KERNEL(k_mov, REP8("v_mov_b32 %0, %1\n v_mov_b32 %3, %4\n v_mov_b32 %6, %7\n v_mov_b32 %2, %5\n"))
KERNEL(k_int, REP8("v_add_u32 %12, %12, %13\n v_and_b32 %14, %14, %15\n v_or_b32 %13, %13, %15\n v_sub_u32 %15, %15, %12\n"))
KERNEL(k_shift, REP8("v_lshlrev_b32 %12, 3, %13\n v_lshrrev_b32 %14, 5, %15\n v_bfe_u32 %13, %15, 3, 7\n v_lshl_or_b32 %15, %12, 4, %13\n"))
KERNEL(k_med3, REP8("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %3, %3, %4, %5\n v_med3_f32 %6, %6, %7, %1\n v_med3_f32 %2, %2, %5, %4\n"))
KERNEL(k_mad_legacy, REP8("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %3, %4, %5\n v_fmac_f32 %6, %7, %1\n v_fmac_f32 %2, %5, %4\n"))
KERNEL(k_ldexp, REP8("v_ldexp_f32 %0, %0, %12\n v_floor_f32 %3, %4\n v_fract_f32 %6, %7\n v_trunc_f32 %2, %5\n"))
KERNEL(k_cmp_class, REP8("v_cmp_lt_u32 vcc, %12, %13\n v_cmp_eq_u32 s[20:21], %14, %15\n v_cmp_class_f32 s[22:23], %0, %12\n v_cmp_ne_u32 s[24:25], %12, %15\n"))
KERNEL(k_f64_mul_add, REP8("v_add_f64 %8, %8, %9\n v_add_f64 %11, %11, %9\n v_add_f64 %9, %9, %8\n v_add_f64 %10, %10, %8\n"))
KERNEL(k_bperm, REP8("ds_bpermute_b32 %0, %12, %1\n ds_bpermute_b32 %3, %13, %4\n ds_swizzle_b32 %6, %7 offset:swizzle(QUAD_PERM,1,0,3,2)\n ds_bpermute_b32 %2, %14, %5\n") "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_ballot, REP8("v_cmp_gt_f32 vcc, %0, %1\n s_bcnt1_i32_b64 s20, vcc\n v_mbcnt_lo_u32_b32 %12, vcc_lo, 0\n v_mbcnt_hi_u32_b32 %12, vcc_hi, %12\n"))
KERNEL(k_salu_dep, REP8("s_add_u32 s20, s20, s21\n s_add_u32 s20, s20, s21\n s_add_u32 s20, s20, s21\n s_add_u32 s20, s20, s21\n"))
KERNEL(k_vsv, REP8("v_cmp_gt_f32 s[20:21], %0, %1\n s_and_b64 s[22:23], s[20:21], exec\n v_cndmask_b32 %3, %3, %4, s[22:23]\n v_fma_f32 %0, %0, %1, %3\n"))

typedef void (*kern_t)(float*, int, float);

static void run(const char* name, kern_t kern, int insts_per_iter, int waves_per_simd, int cus, double khz, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 5000;
  int blocks = cus * waves_per_simd;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double cycles = ms * 1e-3 * khz * 1e3;
  printf("%-16s waves/SIMD %d: %7.3f ms  %.2f cycles per wave-instruction per SIMD\n", name, waves_per_simd, ms,
         cycles / ((double)iters * insts_per_iter * waves_per_simd));
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  float* out;
  hipMalloc(&out, sizeof(float) * p.multiProcessorCount * 8 * 256);
#define R(K, N) for (int w : {1, 8}) run(#K, K, N, w, p.multiProcessorCount, p.clockRate, out);
  R(k_fma, 32) R(k_max, 32) R(k_max3, 32) R(k_cmp_vcc, 32) R(k_cmp_sgpr, 32) R(k_cndmask_fixed, 32) R(k_cmp_cnd, 32)
  R(k_cmp_cnd_far, 64) R(k_cmp_sand, 32) R(k_mul_lo, 32) R(k_mul_hi, 32) R(k_mad_u64, 32) R(k_mul_u24, 32)
  R(k_mad_u32_u24, 32) R(k_xor, 32) R(k_xor3, 32) R(k_fma64, 32) R(k_add64, 32) R(k_rcp, 32) R(k_rcp64, 32)
  R(k_cvt, 32) R(k_pk_fma, 32) R(k_cmpx, 33) R(k_readlane, 32) R(k_salu, 32) R(k_salu_valu, 32) R(k_fma_sgpr, 32)
  R(k_dpp, 32)
  R(k_cndmask_indep, 32) R(k_cndmask_sgpr, 32) R(k_add, 32) R(k_add_e64, 32) R(k_mov, 32) R(k_int, 32) R(k_shift, 32) R(k_med3, 32)
  R(k_mad_legacy, 32) R(k_ldexp, 32) R(k_cmp_class, 32) R(k_f64_mul_add, 32) R(k_bperm, 33) R(k_ballot, 32) R(k_salu_dep, 32) R(k_vsv, 32)
  hipFree(out);
  return 0;
}
