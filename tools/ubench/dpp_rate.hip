// micro-benchmark for the 8192-point block (round 4): per-SIMD issue cost of the candidates for its cross-lane radix-2 layer --
// v_fmac_f32_dpp (quad_perm partner), v_permlane32_swap, packed ops under a half-empty EXEC mask -- at 4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -w dpp_rate.hip -o dpp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters) {
  f2 a[8], b = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
  for (int i = 0; i < 8; ++i) a[i] = f2{1.f + threadIdx.x * 1e-3f + i, 2.f};
  float sg = (threadIdx.x & 1) ? -1.f : 1.f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));) }
    if (OP == 1) { // the same with the lower 32 lanes masked off
      asm volatile("s_mov_b32 exec_lo, 0\n s_mov_b32 exec_hi, -1" ::: "exec");
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));)
      asm volatile("s_mov_b64 exec, -1" ::: "exec");
    }
    if (OP == 2) { REP16(asm volatile("v_fmac_f32_dpp %0, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %2, %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %4, %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %6, %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[0].x), "+v"(a[1].x), "+v"(a[2].x), "+v"(a[3].x), "+v"(a[4].x), "+v"(a[5].x), "+v"(a[6].x), "+v"(a[7].x) : "v"(sg));) }
    if (OP == 3) { REP16(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %3, %4\n v_permlane32_swap_b32 %5, %6\n v_permlane32_swap_b32 %7, %0" : "+v"(a[0].x), "+v"(a[1].x), "+v"(a[2].x), "+v"(a[3].x), "+v"(a[4].x), "+v"(a[5].x), "+v"(a[6].x), "+v"(a[7].x));) }
    if (OP == 4) { REP16(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %6 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[0].x), "+v"(a[1].x), "+v"(a[2].x), "+v"(a[3].x), "+v"(a[4].x), "+v"(a[5].x), "+v"(a[6].x), "+v"(a[7].x));) }
    if (OP == 5) { // unpacked fma, lower half masked
      asm volatile("s_mov_b32 exec_lo, 0\n s_mov_b32 exec_hi, -1" ::: "exec");
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a[0].x), "+v"(a[1].x), "+v"(a[2].x), "+v"(a[3].x), "+v"(a[4].x), "+v"(a[5].x), "+v"(a[6].x), "+v"(a[7].x) : "v"(b.x), "v"(c.x));)
      asm volatile("s_mov_b64 exec, -1" ::: "exec");
    }
    if (OP == 6) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a[0].x), "+v"(a[1].x), "+v"(a[2].x), "+v"(a[3].x), "+v"(a[4].x), "+v"(a[5].x), "+v"(a[6].x), "+v"(a[7].x) : "v"(b.x), "v"(c.x));) }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  f2 s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
  if (s.x == 123.456f) out[1] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
// workgroup barrier cost: B barriers per iteration between short bursts of arithmetic, WG of NW waves
template <int NT>
__global__ __launch_bounds__(NT) void kb(unsigned long long* out, int iters, int with_barrier) {
  f2 a[8], b = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
  for (int i = 0; i < 8; ++i) a[i] = f2{1.f + threadIdx.x * 1e-3f + i, 2.f};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));)
    if (with_barrier) asm volatile("s_barrier" ::: "memory");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  f2 s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
  if (s.x == 123.456f) out[1] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  const char* names[7] = {"v_pk_fma_f32", "v_pk_fma_f32 (exec = upper 32 lanes)", "v_fmac_f32_dpp quad_perm", "v_permlane32_swap_b32", "v_mov_b32_dpp wave_shr:1", "v_fma_f32 (exec = upper 32 lanes)", "v_fma_f32"};
  typedef void (*kt)(unsigned long long*, int);
  kt ks[7] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>};
  const int iters = 2000, wps = 4;
  for (int op = 0; op < 7; ++op) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(ks[op], dim3(256 * wps), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      if (rep) printf("%-40s %.2f cycles per instruction of one wave (4 waves/SIMD) -> %.2f per SIMD; launch %.3f ms\n", names[op], (double)h[0] / (iters * 128.0), (double)h[0] / (iters * 128.0) / wps, ms);
    }
  }
  // barriers: 128 packed instructions (512 issue cycles per wave) between barriers
  for (int nt = 256; nt <= 512; nt *= 2) for (int wb = 0; wb < 2; ++wb) for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int wgs = 256 * (nt == 256 ? 4 : 2);   // 16 waves per CU either way
    if (nt == 256) hipLaunchKernelGGL(kb<256>, dim3(wgs), dim3(256), 0, 0, d, iters, wb); else hipLaunchKernelGGL(kb<512>, dim3(wgs), dim3(512), 0, 0, d, iters, wb);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    if (rep) printf("workgroup of %d threads, 16 waves per CU, 128 v_pk_fma per iteration, barrier %d: %.1f cycles per iteration of one wave; launch %.3f ms\n", nt, wb, (double)h[0] / iters, ms);
  }
  return 0;
}
