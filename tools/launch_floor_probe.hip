// Diagnostic (not product): what a launch of k_slab's SHAPE costs before it does any game work -- 512 blocks x 512 threads,
// 46 KB of LDS per block (two blocks per CU: 4096 resident waves), back-to-back on one stream:
//   empty      every wave returns at once
//   spin N     every wave idles for N cycles (s_memtime): the time of a launch whose waves all take exactly N cycles
//   dirty      every wave stores what a step_slab iteration stores at 65,536 tables (176-B state rows, ~6 list rows, outputs)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_floor_probe tools/launch_floor_probe.hip && /tmp/launch_floor_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(512, 4) void k(int mode, long long spin, uint4* state, uint4* rows, int* counts, long long T) {
  __shared__ uint4 lds[46 * 1024 / 16];
  if (threadIdx.x == 0) lds[blockIdx.x % 64] = make_uint4(1, 2, 3, 4);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 8 + (threadIdx.x >> 6);
  if (mode == 1) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < spin) __builtin_amdgcn_s_sleep(4);
  } else if (mode == 2) {
    for (int i = 0; i < 16; ++i) {
      const long long t = wave * 16 + i;
      if (t >= T) break;
      if (lane < 11) state[t * 11 + lane] = lds[lane];
      if (lane < 6) rows[t * 512 + lane] = lds[lane + 11];
      if (lane == 0) counts[t] = 6;
    }
  }
}

int main() {
  const long long T = 65536;
  uint4 *state, *rows; int* counts;
  if (hipMalloc(&state, T * 176) != hipSuccess || hipMalloc(&rows, T * 512 * 16) != hipSuccess || hipMalloc(&counts, T * 4) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  struct { const char* name; int mode; long long spin; } cases[] = {
      {"empty", 0, 0}, {"spin 10k cycles", 1, 10000}, {"spin 46k cycles (k_slab's mean wave)", 1, 46000},
      {"spin 82k cycles (k_slab's slowest wave)", 1, 82000}, {"dirty (the stores of one step_slab iteration)", 2, 0}};
  for (auto& c : cases) {
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, c.mode, c.spin, state, rows, counts, T);
    (void)hipDeviceSynchronize();
    const int n = 500;
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, c.mode, c.spin, state, rows, counts, T);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-48s %7.2f us per launch (back to back, one stream)\n", c.name, ms * 1e3 / n);
  }
  return 0;
}
