// Issue cost of the instructions of the tree walk's step (and of what a compressed node would need instead),
// 8 waves per SIMD, independent operands.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate4.hip -o valu_rate4
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
// the forms of slab16: an FP16 half (low or high) times an FP32 register plus an FP32 register, with |x| and -x modifiers
KERNEL(k_fma_mix, REP8("v_fma_mix_f32 %0, %12, %1, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %13, %4, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, -%14, |%7|, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %15, |%4|, %7 op_sel_hi:[1,0,0]\n"))
// the root's step: the halves arrive in scalar registers
KERNEL(k_fma_mix_sgpr, REP8("v_fma_mix_f32 %0, s20, %1, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, s21, %4, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, -s22, |%7|, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, s23, |%4|, %7 op_sel_hi:[1,0,0]\n"))
KERNEL(k_cvt_f16, REP8("v_cvt_f32_f16 %0, %12\n v_cvt_f32_f16 %3, %13\n v_cvt_f32_f16 %6, %14\n v_cvt_f32_f16 %2, %15\n"))
KERNEL(k_cvt_ubyte, REP8("v_cvt_f32_ubyte0 %0, %12\n v_cvt_f32_ubyte1 %3, %13\n v_cvt_f32_ubyte2 %6, %14\n v_cvt_f32_ubyte3 %2, %15\n"))
KERNEL(k_lshl_add_u64, REP8("v_lshl_add_u64 %16, %16, 4, %17\n v_lshl_add_u64 %17, %17, 4, %16\n v_lshl_add_u64 %16, %16, 3, %17\n v_lshl_add_u64 %17, %17, 2, %16\n"))
KERNEL(k_max3, REP8("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %3, %3, %4, %5\n v_min3_f32 %6, %6, %7, %1\n v_min3_f32 %2, %2, %5, %4\n"))
KERNEL(k_pk_fma, REP8("v_pk_fma_f32 %8, %8, %9, %10\n v_pk_fma_f32 %11, %11, %9, %10\n v_pk_mul_f32 %9, %9, %8\n v_pk_add_f32 %10, %10, %8\n"))
KERNEL(k_pk_fma_f16, REP8("v_pk_fma_f16 %12, %12, %13, %14\n v_pk_fma_f16 %14, %14, %15, %12\n v_pk_mul_f16 %13, %13, %15\n v_pk_add_f16 %15, %15, %12\n"))

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
  R(k_fma, 32) R(k_fma_mix, 32) R(k_fma_mix_sgpr, 32) R(k_cvt_f16, 32) R(k_cvt_ubyte, 32) R(k_lshl_add_u64, 32) R(k_max3, 32) R(k_pk_fma, 32) R(k_pk_fma_f16, 32)
  hipFree(out);
  return 0;
}
