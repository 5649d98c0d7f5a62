// Diagnostic (not product): does a large straight-line kernel pay for instruction fetch on
// every launch?  A: 64-instruction loop body x N; B: the same work fully unrolled.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int UNROLL, int ITERS>
__global__ void kern(unsigned* out, unsigned seed) {
  unsigned x = seed + threadIdx.x, y = blockIdx.x;
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      x = x * 1664525u + 1013904223u + u;   // distinct constants keep the unrolled body distinct
      y ^= x >> 7;
    }
  }
  if (y == 0x12345678u) out[0] = x;
}

template <int UNROLL, int ITERS>
float run(unsigned* d, int blocks, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((kern<UNROLL, ITERS>), dim3(blocks), dim3(256), 0, 0, d, i);
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((kern<UNROLL, ITERS>), dim3(blocks), dim3(256), 0, 0, d, i);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / reps;
}

int main() {
  unsigned* d;
  CHK(hipMalloc(&d, 4));
  const int blocks = 1024;  // 4096 waves
  printf("work = 2048 (mul-add, xor-shift) pairs per thread, %d blocks x 256\n", blocks);
  printf("loop   unroll   16 x128 : %.2f us/launch\n", run<16, 128>(d, blocks, 300));
  printf("loop   unroll  128 x 16 : %.2f us/launch\n", run<128, 16>(d, blocks, 300));
  printf("loop   unroll  512 x  4 : %.2f us/launch\n", run<512, 4>(d, blocks, 300));
  printf("unrolled      2048 x  1 : %.2f us/launch\n", run<2048, 1>(d, blocks, 300));
  printf("small work: 256 pairs\n");
  printf("loop   unroll   16 x 16 : %.2f us/launch\n", run<16, 16>(d, blocks, 300));
  printf("unrolled       256 x  1 : %.2f us/launch\n", run<256, 1>(d, blocks, 300));
  printf("empty-ish       1 x  1 : %.2f us/launch\n", run<1, 1>(d, blocks, 300));
  return 0;
}
