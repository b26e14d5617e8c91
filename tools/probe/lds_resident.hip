// Probe: how many 64-thread workgroups with X bytes of dynamic LDS are RESIDENT on a CU at once (what the hardware does, not
// what hipOccupancyMaxActiveBlocksPerMultiprocessor computes).  Every workgroup waits ~200 us; a launch of n_cu * k
// workgroups takes one wait if k fit on a CU together, two if they do not.
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ char raw[];
__global__ void __launch_bounds__(64) k(long long ticks, int* out) {
  raw[threadIdx.x] = 1;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0 && raw[5] == 77) out[0] = 1;
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  int* out;
  hipMalloc(&out, 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const long long ticks = 20000;  // 100 MHz wall clock: 200 us
  printf("CUs %d\n", n_cu);
  const int sizes[] = {10240, 11520, 11521, 12800, 12801, 13312, 13648, 13653, 14080, 14081, 15360};
  for (int per_cu = 10; per_cu <= 16; per_cu++) {
    printf("%2d per CU:", per_cu);
    for (int lds : sizes) {
      hipLaunchKernelGGL(k, dim3(n_cu * per_cu), dim3(64), lds, 0, 100, out);  // warm
      hipDeviceSynchronize();
      hipEventRecord(a);
      hipLaunchKernelGGL(k, dim3(n_cu * per_cu), dim3(64), lds, 0, ticks, out);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      printf("  %d:%s", lds, ms < 0.3f ? "fit" : (ms < 0.5f ? "2x" : "3x+"));
    }
    printf("\n");
  }
  return 0;
}
