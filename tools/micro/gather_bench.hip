// Random-access request rate of MI355X HBM: how many scattered 8..64-byte accesses per second the memory system sustains
// (the bound the step kernels run into at full batch).   hipcc --offload-arch=gfx950 -O3 gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
// every thread: R independent random reads of `bytes` (8 or 16) at 64-byte-aligned offsets; mode 1 also writes back
template <int R>
__global__ void k_gather(u64* buf, u64 n_lines, int mode, u64 seed, u64* sink) {
  const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 idx[R], v[R];
#pragma unroll
  for (int r = 0; r < R; r++) idx[r] = (mix(tid * R + r + seed) % n_lines) * 8;
#pragma unroll
  for (int r = 0; r < R; r++) v[r] = buf[idx[r]];
  u64 acc = 0;
#pragma unroll
  for (int r = 0; r < R; r++) acc ^= v[r];
  if (mode == 1) {
#pragma unroll
    for (int r = 0; r < R; r++) buf[idx[r]] = v[r] + 1;
  }
  if (acc == 0x1234567ull) *sink = acc;
}
// streaming read / copy over the same buffer (16-byte loads, grid-stride): what "HBM peak" means in practice on this GPU
__global__ void k_stream(const ulonglong2* __restrict__ src, ulonglong2* __restrict__ dst, u64 n, u64* sink) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 acc = 0;
  for (u64 j = i; j < n; j += (u64)gridDim.x * blockDim.x) {
    ulonglong2 v = src[j];
    if (dst) dst[j] = v; else acc ^= v.x ^ v.y;
  }
  if (!dst && acc == 0x1234567ull) *sink = acc;
}
int main() {
  const u64 bytes = 2ull << 30, n_lines = bytes / 64;
  u64 *buf, *sink;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 8);
  hipMemset(buf, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; mode++)
    for (int threads_log = 18; threads_log <= 22; threads_log += 2) {
      const u64 nthreads = 1ull << threads_log;
      constexpr int R = 8;
      hipLaunchKernelGGL(k_gather<R>, dim3(nthreads / 256), dim3(256), 0, 0, buf, n_lines, mode, 1ull, sink);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int reps = 10;
      for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_gather<R>, dim3(nthreads / 256), dim3(256), 0, 0, buf, n_lines, mode, 77ull + i, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double acc = (double)nthreads * R * reps * (mode ? 2 : 1);
      printf("%s threads 2^%d x %d accesses: %.2f us per launch, %.1f G accesses/s (64-B lines over a 2 GiB buffer)\n",
             mode ? "read+write" : "read      ", threads_log, R, ms * 1e3 / reps, acc / (ms * 1e-3) / 1e9);
    }
  for (int copy = 0; copy < 2; copy++) {
    const u64 half = bytes / 2, n = (copy ? half : bytes) / 16;
    ulonglong2* src = (ulonglong2*)buf;
    ulonglong2* dst = copy ? (ulonglong2*)((char*)buf + half) : nullptr;
    hipLaunchKernelGGL(k_stream, dim3(256 * 32), dim3(256), 0, 0, src, dst, n, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_stream, dim3(256 * 32), dim3(256), 0, 0, src, dst, n, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double moved = (double)n * 16 * reps * (copy ? 2 : 1);
    printf("%s: %.2f TB/s (%s)\n", copy ? "stream copy 1 GiB -> 1 GiB" : "stream read 2 GiB", moved / (ms * 1e-3) / 1e12,
           copy ? "read + write bytes" : "read bytes");
  }
  return 0;
}
