#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, float sa) {
  f2 x = {1.0f + threadIdx.x, 2.0f}, y;
  asm volatile(
    "s_mov_b32 s43, %1\n"
    "s_mov_b32 s44, 0x40000000\n"   // 2.0
    "s_mov_b32 s45, 0x40400000\n"   // 3.0
    "v_mov_b32 v104, s43\n"
    "v_mov_b32 v105, s44\n"
    "v_mov_b32 v106, s45\n"
    "v_mul_f32 v105, v105, v104\n"
    "v_mul_f32 v106, v106, v104\n"
    "v_readfirstlane_b32 s44, v105\n"
    "v_readfirstlane_b32 s45, v106\n"
    "s_nop 4\n"
    "v_pk_mul_f32 %0, %2, s[44:45]\n"
    : "=v"(y) : "s"(sa), "v"(x) : "s43", "s44", "s45", "v104", "v105", "v106");
  out[2 * threadIdx.x] = y.x; out[2 * threadIdx.x + 1] = y.y;
}
int main() {
  float* d; hipMalloc(&d, 1024);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 0.5f);
  float h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("expect 1.0 3.0 | 2.0 3.0 -> got %g %g | %g %g\n", h[0], h[1], h[2], h[3]);
  return 0;
}
