// micro-benchmark: do VALU (v_pk_fma_f32) work and LDS traffic (ds_write_b64 + ds_read_b64) of DIFFERENT
// waves on the same SIMD/CU overlap on gfx950?  256-thread workgroups, 2 per CU (70 KiB of LDS each);
// role by blockIdx parity: 0 = VALU only, 1 = LDS only.  Times: all-VALU, all-LDS, half/half.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef volatile __attribute__((address_space(3))) v2f lds_v2f;
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, int mode) {
  __shared__ v2f buf[8704];            // 68 KiB: two workgroups per CU
  const int t = threadIdx.x;
  // mode 0: all VALU; 1: all LDS; 2: even blocks VALU, odd LDS; 3: every wave alternates both (same wave)
  const int role = mode == 2 ? ((blockIdx.x >> 8) & 1) : mode;
  v2f p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = v2f{t * 1e-3f + i, 1.f + i};
  const v2f pc = {1.0001f, 0.9999f}, pd = {0.5f, 0.25f};
  lds_v2f* w = (lds_v2f*)&buf[t];
  for (int it = 0; it < iters; ++it) {
    if (role == 0 || role == 3) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(pc), "v"(pd));
    }
    if (role == 1 || role == 3) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int c = 0; c < 8; ++c) w[c * 272 + r * 2176] = p[c];
#pragma unroll
        for (int c = 0; c < 8; ++c) p[c] = w[c * 272 + r * 2176];
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * 256 + t] = s;
}
int main() {
  float* d; hipMalloc(&d, 512 * 256 * 4);
  const char* names[] = {"all VALU (128 pk_fma / iter)", "all LDS (16 wr + 16 rd b64 / iter)", "half the workgroups each", "every wave does both"};
  for (int mode = 0; mode < 4; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d, 100, mode);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d, 20000, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.3f ms\n", names[mode], ms);
  }
  return 0;
}
