// debugging aid (tools/attic/sv8_core_test.py): ONE workgroup runs the generated 8192-point CORE block on a staged block read from
// sv8_test_in.bin, once per tap; the 16 register pairs of the tap go to gpurun_out/sv8_stage_<k>.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "sv8_test_asm.inc"
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kLds = 16 * 560 * 8 + 8192;
template <int ST>
__global__ __launch_bounds__(512, 4) void k(const float* stage, const float* gc, const float* tab, float* out, float ca, float sa) {
  extern __shared__ f2 smem[];
  const int t0 = threadIdx.x;
  volatile __attribute__((address_space(3))) f2* st = (volatile __attribute__((address_space(3))) f2*)smem;
  volatile __attribute__((address_space(3))) float* g = (volatile __attribute__((address_space(3))) float*)(smem + 8192);
  for (int r = 0; r < 16; ++r) st[512 * r + t0] = f2{stage[2 * (512 * r + t0)], stage[2 * (512 * r + t0) + 1]};
  g[t0] = gc[t0];
  __syncthreads();
  const unsigned lds_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) f2*)smem;
  unsigned long long a_sv = (unsigned long long)tab, a_out = (unsigned long long)out;
  a_sv = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(a_sv >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)a_sv);
  a_out = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(a_out >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)a_out);
  const int one = 1;
  ca = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ca)));
  sa = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(sa)));
#define CALL(N) if (ST == N) asm volatile(SV8T_ASM_##N : : "s"(a_sv), "s"(a_out), "s"(one), "s"(lds_off), "s"(a_sv), "s"(ca), "s"(sa), "v"(t0) : SV8T_CLOBBERS);
  CALL(0) CALL(1) CALL(2) CALL(3) CALL(4) CALL(5) CALL(6) CALL(7) CALL(8) CALL(9) CALL(10)
}
int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  FILE* f = fopen("tools/ubench/sv8_test_in.bin", "rb");
  if (!f) { printf("no input\n"); return 1; }
  std::vector<float> st(16 * 512 * 2), gc(512), tab(5 * 512 * 2), sc(2);
  fread(st.data(), 4, st.size(), f); fread(gc.data(), 4, gc.size(), f); fread(tab.data(), 4, tab.size(), f); fread(sc.data(), 4, 2, f); fclose(f);
  float *dst, *dgc, *dtab, *dout;
  hipMalloc(&dst, st.size() * 4); hipMalloc(&dgc, gc.size() * 4); hipMalloc(&dtab, tab.size() * 4); hipMalloc(&dout, 16 * 512 * 2 * 4);
  hipMemcpy(dst, st.data(), st.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dgc, gc.data(), gc.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
  typedef void (*kt)(const float*, const float*, const float*, float*, float, float);
  kt ks[11] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>};
  system("mkdir -p gpurun_out");
  for (int s = 0; s < SV8T_NSTAGE; ++s) {
    if (only >= 0 && s != only) continue;
    hipFuncSetAttribute((const void*)ks[s], hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    hipMemset(dout, 0, 16 * 512 * 2 * 4);
    hipLaunchKernelGGL(ks[s], dim3(1), dim3(512), kLds, 0, dst, dgc, dtab, dout, sc[0], sc[1]);
    hipError_t e = hipDeviceSynchronize();
    std::vector<float> o(16 * 512 * 2);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    char fn[128]; snprintf(fn, sizeof fn, "gpurun_out/sv8_stage_%d.bin", s);
    FILE* g = fopen(fn, "wb"); fwrite(o.data(), 4, o.size(), g); fclose(g);
    printf("stage %d: %s\n", s, hipGetErrorString(e));
  }
  return 0;
}
