// micro-benchmark: at what rate do the radix-16 butterflies of prot_fft.h (packed fp32) issue on gfx950, alone?
// KIND 0: pkt_fft16 back to back on 16 complex registers; 1: the same plus 15 twiddle multiplies from registers;
// 2: a stream of INDEPENDENT v_pk_fma_f32 on 16 register pairs (the pipe's own rate, same register footprint);
// 3: v_pk_add_f32 with the op_sel / neg modifiers of add_mj on 16 register pairs.
// WPS workgroups of 256 threads per CU (1..3 waves per SIMD).  Prints ns and shader cycles per packed instruction per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -w -I phaserotate.lv2_amd/csrc tools/ubench/fft16_rate.hip -o tools/ubench/fft16_rate
#include "prot_fft.h"
#include <cstdio>
#include <cstdlib>
using namespace prot;
template <int KIND, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, int iters, unsigned long long* cyc) {
  f2 v[16], w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = mkf2(threadIdx.x * 1e-3f + i, 1.f - i * 0.01f); w[i] = mkf2(0.999f, 0.01f * i); }
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) { pkt_fft16<pk32, 1>(v); }
    else if (KIND == 1) {
      pkt_fft16<pk32, 1>(v);
      v[1] = pk32::cmul<1>(v[1], w[1]);
#pragma unroll
      for (int i = 2; i < 16; i += 2) pk32::cmul2<1>(v[i], w[i], v[i + 1], w[i + 1]);
    } else if (KIND == 2) {
#pragma unroll
      for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[1]), "v"(w[2]));
    } else {
#pragma unroll
      for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(v[i]) : "v"(w[(i + 1) & 15]));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(v[i]));
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = c1 - c0;
}
template <int KIND, int WPS>
static void run(const char* name, int pk_per_iter) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 3 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 4000, grid = 256 * WPS;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, WPS>), dim3(grid), dim3(256), 0, 0, out, 10, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, WPS>), dim3(grid), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // per SIMD: WPS waves, each iters * pk_per_iter packed instructions
  const double n = (double)iters * pk_per_iter * WPS;
  printf("%-34s %d waves/SIMD: %.3f ms, %.2f ns / %.2f shader cycles per packed instruction per SIMD (clock %.0f MHz)\n", name, WPS, ms, ms * 1e6 / n,
         (double)c / ((double)iters * pk_per_iter) / WPS, (double)c / (ms * 1e3));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<2, 1>("independent v_pk_fma_f32", 80); run<2, 2>("independent v_pk_fma_f32", 80); run<2, 3>("independent v_pk_fma_f32", 80);
  run<3, 1>("v_pk_add_f32 op_sel/neg", 80); run<3, 3>("v_pk_add_f32 op_sel/neg", 80);
  run<0, 1>("pkt_fft16 (80 packed)", 80); run<0, 2>("pkt_fft16 (80 packed)", 80); run<0, 3>("pkt_fft16 (80 packed)", 80);
  run<1, 1>("pkt_fft16 + 15 cmul (110 packed)", 110); run<1, 3>("pkt_fft16 + 15 cmul (110 packed)", 110);
  return 0;
}
