// Probe: what hipOccupancyMaxActiveBlocksPerMultiprocessor ANSWERS for 64-thread workgroups as a function of the dynamic LDS
// request.  On MI355X it counts 16-byte pieces (13 648 B -> 12 per CU); the hardware allocates 1 280-byte pieces (11 per CU):
// lds_resident.hip measures what is really resident.
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ char raw[];
__global__ void __launch_bounds__(64) k(int* out) { raw[threadIdx.x] = 1; __syncthreads(); out[threadIdx.x] = raw[63 - threadIdx.x]; }
int main() {
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  int prev = -1;
  for (int lds = 1024; lds <= 64 * 1024; lds += 16) {
    int n = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k, 64, lds);
    if (n != prev) printf("lds %6d -> %d workgroups/CU\n", lds, n);
    prev = n;
  }
  return 0;
}
