// micro-benchmark: v_mfma_f32_32x32x2_f32 issue rate alone and with v_max3_f32 consumers (the sweep's inner loop)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a = threadIdx.x * 1e-3f + 0.5f, b = 1.0f - threadIdx.x * 1e-3f;
  f32x16 zero; for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  f32x16 m[NACC]; for (int q = 0; q < NACC; ++q) m[q] = zero;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
      if (KIND == 0) {           // accumulate chain per accumulator (pure MFMA)
        m[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, m[q], 0, 0, 0);
      } else if (KIND == 1) {    // C = 0, result consumed by 16 v_max (one per output)
        f32x16 r0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b + q, zero, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) m[q][r] = fmaxf(m[q][r], fabsf(r0[r]));
      } else {                   // C = 0, two MFMAs consumed by 16 v_max3
        f32x16 r0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b + q, zero, 0, 0, 0);
        f32x16 r1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a + 1, b + q, zero, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) m[q][r] = fmaxf(m[q][r], fmaxf(fabsf(r0[r]), fabsf(r1[r])));
      }
    }
    a += 1e-6f;
  }
  float s = 0; for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) s += m[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND, int NACC> void run(const char* name, float* d) {
  for (int wps : {1, 2, 3, 4}) {
    int blocks = 256 * wps, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, d, 50);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)blocks * 4 * iters * NACC * (KIND == 2 ? 2 : 1);   // MFMAs issued (per wave)
    double per_simd = mf / 1024.0;                                         // per SIMD
    printf("%-28s waves/SIMD %d: %.3f ms, %.1f ns per MFMA per SIMD (=%.0f cycles @2.1GHz)\n", name, wps, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.1);
  }
}
int main() {
  float* d; hipMalloc(&d, 256 * 4 * 256 * 4 * 2);
  run<0, 4>("mfma chain x4 acc", d);
  run<1, 3>("mfma + 16 v_max", d);
  run<2, 3>("2 mfma + 16 v_max3", d);
  return 0;
}
