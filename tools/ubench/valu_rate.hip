// micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_pk_add_f32 on gfx950
// at 1..8 waves per SIMD (256-thread blocks, blocks/CU controlled by grid size).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const float c = 1.0001f, d = 0.5f;
  const v2f pc = {c, c}, pd = {d, d};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (KIND == 0) { // 8 scalar fma
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(c), "v"(d));
      } else if (KIND == 1) { // 4 packed fma = 8 fma
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pc), "v"(pd));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pc), "v"(pd));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(pc), "v"(pd));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pc), "v"(pd));
      } else if (KIND == 2) { // 8 scalar add
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(d));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(d));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(d));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a3) : "v"(d));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a4) : "v"(d));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a5) : "v"(d));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a6) : "v"(d));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a7) : "v"(d));
      } else { // 4 packed add = 8 add
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(pd));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p1) : "v"(pd));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2) : "v"(pd));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p3) : "v"(pd));
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0+a1+a2+a3+a4+a5+a6+a7 + p0.x+p0.y+p1.x+p1.y+p2.x+p2.y+p3.x+p3.y;
}
template <int KIND> void run(const char* name, float* d) {
  for (int wps : {1, 2, 4, 8}) {
    int blocks = 256 * wps;  // 256 CUs x wps blocks of 4 waves -> wps waves per SIMD
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double lane_ops = (double)blocks * 256 * iters * 64.0;  // 64 scalar-equivalent ops per iter per lane
    printf("%-12s waves/SIMD %d: %.3f ms, %.2f T lane-ops/s (scalar-equivalent)\n", name, wps, ms, lane_ops / ms / 1e9);
  }
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
  run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_add_f32", d); run<3>("v_pk_add_f32", d);
  return 0;
}
