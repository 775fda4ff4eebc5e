// micro-benchmark: dependent-issue cost of v_pk_fma_f32 / v_pk_add_f32 / v_fma_f32 on gfx950.
// NCH independent chains per wave (1, 2, 4, 8), 1..3 waves per SIMD.  Prints cycles-equivalent ns per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND, int NCH>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  v2f p[8];
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { p[i] = v2f{threadIdx.x * 1e-3f + i, 1.f + i}; a[i] = threadIdx.x * 1e-3f + i; }
  const v2f pc = {1.0001f, 0.9999f}, pd = {0.5f, 0.25f};
  const float c = 1.0001f, d = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 64 / NCH; ++j) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[ch]) : "v"(pc), "v"(pd));
        else if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[ch]) : "v"(pd));
        else if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[ch]) : "v"(c), "v"(d));
        else asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(p[ch]) : "v"(pc), "v"(pd));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y + a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND, int NCH> void run(const char* name, float* d) {
  for (int wps : {1, 2, 3, 4}) {
    int blocks = 256 * wps, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NCH>), dim3(blocks), dim3(256), 0, 0, d, 50);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NCH>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)wps * iters * 64.0;
    printf("%-22s chains %d waves/SIMD %d: %.3f ms, %.2f ns per instr per SIMD\n", name, NCH, wps, ms, ms * 1e6 / instr_per_simd);
  }
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
  run<0, 1>("v_pk_fma_f32", d); run<0, 2>("v_pk_fma_f32", d); run<0, 4>("v_pk_fma_f32", d); run<0, 8>("v_pk_fma_f32", d);
  run<3, 1>("v_pk_fma_f32 swz", d); run<3, 4>("v_pk_fma_f32 swz", d);
  run<1, 1>("v_pk_add_f32", d); run<1, 2>("v_pk_add_f32", d); run<1, 4>("v_pk_add_f32", d);
  run<2, 1>("v_fma_f32", d); run<2, 2>("v_fma_f32", d); run<2, 4>("v_fma_f32", d);
  return 0;
}
