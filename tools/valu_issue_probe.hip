// Diagnostic (not product): what does one wave64 vector-ALU instruction cost on a gfx950 SIMD, for the
// instruction kinds the nibble-SWAR engine actually emits, at 1 / 2 / 4 / 8 resident waves per SIMD?
// Settles the peak used by bench.py's roofline.issue: MI355X_MICROARCH.md says 2 cycles per
// wave64 VALU instruction with more than one wave resident (SIMD-32), 4 for a lone wave.
//
// Method: every wave runs REPS x 64 independent instructions of one kind (8 independent register chains, inline
// asm so that the compiler can neither merge nor reorder them), stamped with s_memtime (shader clock ticks);
// the launch is also timed with HIP events.  Reported per (kind, waves/SIMD):
//   cyc/instr/wave  = wave-local s_memtime delta / instructions      (what ONE wave sees)
//   cyc/instr/SIMD  = that / waves per SIMD                           (issue slots the SIMD spends per instruction)
//   G wave-instr/s  = all instructions / event time                   (chip-wide rate, for the roofline peak)
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue_probe tools/valu_issue_probe.hip && /tmp/valu_issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

enum { K_ADD_U32, K_AND_OR, K_LSHL_B64, K_ADD_U64, K_SUB_U64, K_CMP_CNDMASK, K_MUL_LO, K_MBCNT, K_READLANE, K_FMA_F32, K_NKINDS };
static const char* KNAME[K_NKINDS] = {"v_add_u32", "v_and_b32/v_or_b32", "v_lshlrev_b64", "v_add_co_u32+v_addc_co_u32 (64-bit add)",
                                      "v_sub_co_u32+v_subb_co_u32 (64-bit sub)", "v_cmp_gt_u32+v_cndmask_b32", "v_mul_lo_u32",
                                      "v_mbcnt_lo+v_mbcnt_hi", "v_readlane_b32 (to SGPR)", "v_fma_f32"};

// 8 instructions on 8 independent chains; 8 of these per loop body = 64 instructions (pairs count as 2)
#define I8(s) s(0) s(1) s(2) s(3) s(4) s(5) s(6) s(7)
#define BODY(s) I8(s) I8(s) I8(s) I8(s) I8(s) I8(s) I8(s) I8(s)

template <int KIND>
__global__ void probe(unsigned long long* ticks, unsigned* sink, int reps) {
  unsigned a[8], b[8];
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i; b[i] = blockIdx.x + 17u * i; f[i] = (float)a[i]; }
  unsigned s_acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    if (KIND == K_ADD_U32) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      BODY(S)
#undef S
    } else if (KIND == K_AND_OR) {
#define S(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      I8(S) I8(S) I8(S) I8(S)
#undef S
#define S(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      I8(S) I8(S) I8(S) I8(S)
#undef S
    } else if (KIND == K_LSHL_B64) {
#define S(i) { unsigned long long v_ = ((unsigned long long)a[i] << 32) | b[i]; asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(v_)); a[i] = (unsigned)(v_ >> 32); b[i] = (unsigned)v_; }
      BODY(S)
#undef S
    } else if (KIND == K_ADD_U64) {  // 32 pairs = 64 instructions
#define S(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(a[(i + 1) & 7]) : "vcc");
      I8(S) I8(S) I8(S) I8(S)
#undef S
    } else if (KIND == K_SUB_U64) {
#define S(i) asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n\tv_subb_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(a[(i + 1) & 7]) : "vcc");
      I8(S) I8(S) I8(S) I8(S)
#undef S
    } else if (KIND == K_CMP_CNDMASK) {
#define S(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
      I8(S) I8(S) I8(S) I8(S)
#undef S
    } else if (KIND == K_MUL_LO) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      BODY(S)
#undef S
    } else if (KIND == K_MBCNT) {
#define S(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, 0\n\tv_mbcnt_hi_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
      I8(S) I8(S) I8(S) I8(S)
#undef S
    } else if (KIND == K_READLANE) {
#define S(i) { unsigned s_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(a[i])); s_acc ^= s_; }
      BODY(S)
#undef S
    } else {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
      BODY(S)
#undef S
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned x = s_acc;
#pragma unroll
  for (int i = 0; i < 8; ++i) x ^= a[i] ^ b[i] ^ __float_as_uint(f[i]);
  if (x == 0x12345679u) sink[0] = x;
  if ((threadIdx.x & 63) == 0) ticks[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(unsigned long long*, unsigned*, int);
static kern_t KERNS[K_NKINDS] = {probe<K_ADD_U32>, probe<K_AND_OR>, probe<K_LSHL_B64>, probe<K_ADD_U64>, probe<K_SUB_U64>,
                                 probe<K_CMP_CNDMASK>, probe<K_MUL_LO>, probe<K_MBCNT>, probe<K_READLANE>, probe<K_FMA_F32>};

int main() {
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  int wall_khz = 0;
  CHK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
  printf("# %s, %d CUs, clockRate %d kHz, s_memtime/wall clock %d kHz\n", p.gcnArchName, cus, p.clockRate, wall_khz);
  unsigned long long* ticks;
  unsigned* sink;
  const int max_waves = cus * 32;
  CHK(hipMalloc(&ticks, sizeof(unsigned long long) * max_waves));
  CHK(hipMalloc(&sink, 4));
  unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * max_waves);
  const int reps = 4000;  // x 64 instructions per wave
  printf("# kind | waves/SIMD | s_memtime ticks/instr/wave (median) | event us | G wave-instr/s chip-wide | cycles/instr/SIMD at event time and %d CUs x 4 SIMDs (clock from GRBM not available here: uses 2.4 GHz)\n", cus);
  const int wps[4] = {1, 2, 4, 8};
  for (int k = 0; k < K_NKINDS; ++k)
    for (int wi = 0; wi < 4; ++wi) {
      const int w = wps[wi];
      // w waves per SIMD: blocks of 256 threads (4 waves = one per SIMD), w blocks per CU
      const int blocks = cus * w, threads = 256;
      const int waves = blocks * 4;
      hipEvent_t e0, e1;
      CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
      hipLaunchKernelGGL(KERNS[k], dim3(blocks), dim3(threads), 0, 0, ticks, sink, 200);  // warm
      CHK(hipDeviceSynchronize());
      CHK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(KERNS[k], dim3(blocks), dim3(threads), 0, 0, ticks, sink, reps);
      CHK(hipEventRecord(e1, 0));
      CHK(hipEventSynchronize(e1));
      float ms = 0;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      CHK(hipMemcpy(h, ticks, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
      // median of the per-wave tick counts
      for (int i = 1; i < waves; ++i) { unsigned long long v = h[i]; int j = i; while (j > 0 && h[j - 1] > v) { h[j] = h[j - 1]; --j; } h[j] = v; }
      const double instr = 64.0 * reps;
      const double med = (double)h[waves / 2] / instr;
      const double rate = instr * waves / (ms * 1e-3);
      // s_memtime counts at the constant 100 MHz-class reference on some parts: report raw ticks, and cycles from the event time
      const double cyc_simd = (ms * 1e-3) * 2.4e9 / (instr * w);
      printf("%-44s | %d | %8.3f | %9.1f | %8.1f | %6.2f\n", KNAME[k], w, med, ms * 1e3, rate / 1e9, cyc_simd);
      fflush(stdout);
      CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    }
  return 0;
}
