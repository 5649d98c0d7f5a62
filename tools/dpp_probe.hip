// Diagnostic (not product): wave64 inclusive scan / max-reduction with DPP row operations against the ds_bpermute
// (__shfl) forms -- correctness on random data and cycles per call at 1 wave per SIMD (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp_probe tools/dpp_probe.hip && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../doudizhu-rl_amd/csrc/ddz_device.h"
using namespace ddz;

__device__ __forceinline__ int shfl_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(v, d); if (lane >= d) v += y; }
  return v;
}
__device__ __forceinline__ int shfl_max(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(v, d); v = o > v ? o : v; }
  return v;
}
__device__ __forceinline__ double shfl_maxd(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(v, d); v = o > v ? o : v; }
  return v;
}

__global__ void k(const int* in, const double* ind, int* out, double* outd, unsigned long long* cyc, int reps) {
  const int lane = threadIdx.x & 63;
  int v = in[threadIdx.x];
  double dv = ind[threadIdx.x];
  out[0 * 64 + lane] = shfl_scan(v, lane);
  out[1 * 64 + lane] = wave_scan_add(v);
  out[2 * 64 + lane] = shfl_max(v);
  out[3 * 64 + lane] = wave_max_i32(v);
  out[4 * 64 + lane] = wave_sum_i32(v);
  outd[0 * 64 + lane] = shfl_maxd(dv);
  outd[1 * 64 + lane] = wave_max_f64(dv);
  int acc = v;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) acc = shfl_scan(acc & 3, lane);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) acc = wave_scan_add(acc & 3);
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) acc = shfl_max(acc ^ i);
  unsigned long long t3 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) acc = wave_max_i32(acc ^ i);
  unsigned long long t4 = __builtin_amdgcn_s_memtime();
  double dacc = dv;
  for (int i = 0; i < reps; ++i) dacc = shfl_maxd(dacc + lane * 1e-9);
  unsigned long long t5 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) dacc = wave_max_f64(dacc + lane * 1e-9);
  unsigned long long t6 = __builtin_amdgcn_s_memtime();
  out[5 * 64 + lane] = acc + (int)dacc;
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; cyc[5] = t6 - t5; }
}

int main() {
  int h[64]; double hd[64];
  int *din, *dout; double *dind, *doutd; unsigned long long* dc;
  hipMalloc(&din, 256); hipMalloc(&dout, 6 * 256); hipMalloc(&dind, 512); hipMalloc(&doutd, 2 * 512); hipMalloc(&dc, 64);
  int bad = 0;
  for (int trial = 0; trial < 200; ++trial) {
    for (int i = 0; i < 64; ++i) { h[i] = (rand() % 2001) - 1000; hd[i] = (rand() % 1000) / 7.0 - 50.0; if (trial % 7 == 0) hd[i] = -1.0 / 0.0; }
    hipMemcpy(din, h, 256, hipMemcpyHostToDevice); hipMemcpy(dind, hd, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dind, dout, doutd, dc, 1);
    int o[6 * 64]; double od[2 * 64];
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost); hipMemcpy(od, doutd, sizeof(od), hipMemcpyDeviceToHost);
    int sum = 0;
    for (int i = 0; i < 64; ++i) {
      sum += h[i];
      if (o[i] != o[64 + i] || o[128 + i] != o[192 + i]) ++bad;
      if (od[i] != od[64 + i] && !(od[i] != od[i])) ++bad;
    }
    for (int i = 0; i < 64; ++i) if (o[256 + i] != sum) ++bad;
  }
  const int reps = 2000;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dind, dout, doutd, dc, reps);
  unsigned long long c[6];
  hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
  printf("mismatches: %d\n", bad);
  const char* nm[6] = {"scan shfl", "scan dpp", "max shfl", "max dpp", "max f64 shfl", "max f64 dpp"};
  for (int i = 0; i < 6; ++i) printf("%-14s %.1f s_memtime ticks per call\n", nm[i], (double)c[i] / reps);
  return bad != 0;
}
