// Prints the LDS limits HIP reports and checks that a kernel can be launched with a large dynamic LDS size.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out, int n) {
  extern __shared__ int lds[];
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = i;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[n - 1];
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu sharedMemPerBlockOptin %zu regsPerBlock %d\n", p.sharedMemPerBlock,
         p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlockOptin, p.regsPerBlock);
  int* out;
  hipMalloc(&out, 4096);
  for (size_t kb : {48, 64, 96, 128, 160}) {
    size_t bytes = kb * 1024;
    hipError_t a = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipLaunchKernelGGL(k, dim3(4), dim3(256), bytes, 0, out, (int)(bytes / 4));
    hipError_t e = hipGetLastError();
    hipError_t s = hipDeviceSynchronize();
    int h = -1;
    hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, bytes);
    printf("%zu KB: setattr %s launch %s sync %s out %d blocks/CU %d\n", kb, hipGetErrorName(a), hipGetErrorName(e), hipGetErrorName(s), h, nb);
  }
  return 0;
}
