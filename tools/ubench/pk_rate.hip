// micro-benchmark: cycles per wave-instruction of the packed-fp32 operations the OLA kernels are made of, by waves per SIMD
// (shader cycles by s_memtime).  build: hipcc --offload-arch=gfx950 -O3 -w pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters) {
  extern __shared__ char dyn[];
  f2 a[8], b = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
  for (int i = 0; i < 8; ++i) a[i] = f2{1.f + threadIdx.x * 1e-3f + i, 2.f};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) { REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
    if (OP == 1) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
    if (OP == 2) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
    if (OP == 3) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a[0].x), "+v"(a[1].x), "+v"(a[2].x), "+v"(a[3].x), "+v"(a[4].x), "+v"(a[5].x), "+v"(a[6].x), "+v"(a[7].x) : "v"(c.x));) }
    if (OP == 4) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a[0].x), "+v"(a[1].x), "+v"(a[2].x), "+v"(a[3].x), "+v"(a[4].x), "+v"(a[5].x), "+v"(a[6].x), "+v"(a[7].x) : "v"(b.x), "v"(c.x));) }
    if (OP == 5) { REP16(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(a[0].x), "+v"(a[1].x), "+v"(a[2].x), "+v"(a[3].x), "+v"(a[4].x), "+v"(a[5].x), "+v"(a[6].x), "+v"(a[7].x) : "v"(c.x));) }
    if (OP == 6) { REP16(asm volatile("v_pk_add_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %6, %6, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %7, %7, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  f2 s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
  if (s.x == 123.456f) out[1] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  const char* names[7] = {"v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_add_f32", "v_fma_f32", "v_mov_b32", "v_pk_add_f32 (swizzled)"};
  typedef void (*kt)(unsigned long long*, int);
  kt ks[7] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>};
  for (int i = 0; i < 7; ++i) hipFuncSetAttribute((const void*)ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int iters = 2000;
  for (int wps = 1; wps <= 8; wps *= 2) {
    const unsigned lds = wps >= 8 ? 0 : ((160 * 1024 / wps) & ~1023u);   // wps workgroups of 4 waves per CU = wps waves per SIMD
    for (int op = 0; op < 7; ++op) {
      hipLaunchKernelGGL(ks[op], dim3(256 * wps), dim3(256), lds, 0, d, iters);
      hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(ks[op], dim3(256 * wps), dim3(256), lds, 0, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      // workgroup 0's wave 0 shares its SIMD with wps - 1 other waves running the same loop: cycles per instruction of ONE wave / wps = per-SIMD issue cost
      const double per = (double)h[0] / (iters * 128.0);
      printf("waves/SIMD %d  %-26s %.2f cycles per instruction of one wave; launch %.3f ms = %.2f ns per instruction per SIMD\n", wps, names[op], per, ms,
             ms * 1e6 / (iters * 128.0 * wps));
    }
  }
  return 0;
}
