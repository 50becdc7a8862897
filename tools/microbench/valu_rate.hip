// Measures the issue rate of plain FP32 VALU instructions (v_fma_f32, v_cndmask, v_cmp, v_mad_u64_u32)
// per SIMD at several occupancies. Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned long long u0 = threadIdx.x + 1, u1 = u0 + 7;
  const float m = 1.0000001f, c = 1e-9f;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // 8 independent v_fma_f32
      a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
      a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
    } else if (KIND == 1) {  // v_mad_u64_u32 chains
      u0 = (unsigned long long)(unsigned)u0 * 0xD2511F53u + (u0 >> 32);
      u1 = (unsigned long long)(unsigned)u1 * 0xCD9E8D57u + (u1 >> 32);
      u0 = (unsigned long long)(unsigned)u0 * 0xD2511F53u + (u0 >> 32);
      u1 = (unsigned long long)(unsigned)u1 * 0xCD9E8D57u + (u1 >> 32);
      u0 = (unsigned long long)(unsigned)u0 * 0xD2511F53u + (u0 >> 32);
      u1 = (unsigned long long)(unsigned)u1 * 0xCD9E8D57u + (u1 >> 32);
      u0 = (unsigned long long)(unsigned)u0 * 0xD2511F53u + (u0 >> 32);
      u1 = (unsigned long long)(unsigned)u1 * 0xCD9E8D57u + (u1 >> 32);
    } else {  // compare + select pairs
      a0 = (a1 > a2) ? a3 : a0; a1 = (a2 > a3) ? a4 : a1; a2 = (a3 > a4) ? a5 : a2; a3 = (a4 > a5) ? a6 : a3;
      a4 = (a5 > a6) ? a7 : a4; a5 = (a6 > a7) ? a0 : a5; a6 = (a7 > a0) ? a1 : a6; a7 = (a0 > a1) ? a2 : a7;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1);
}

template <int KIND>
void run(const char* name, int insts_per_iter) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount;
  float* out;
  hipMalloc(&out, sizeof(float) * cus * 32 * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 200000;
  for (int waves_per_simd : {1, 2, 4, 8}) {
    int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1000, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double wave_insts_per_simd = (double)iters * insts_per_iter * waves_per_simd;
    double cycles = ms * 1e-3 * p.clockRate * 1e3;  // clockRate in kHz
    printf("%-14s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD (clock %.0f MHz)\n", name,
           waves_per_simd, ms, cycles / wave_insts_per_simd, p.clockRate / 1e3);
  }
  hipFree(out);
}

int main() {
  run<0>("v_fma_f32", 8);
  run<1>("v_mad_u64_u32", 8);
  run<2>("v_cmp+cndmask", 16);
  return 0;
}
