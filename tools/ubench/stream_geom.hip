// micro-benchmark: what HBM rate does the OLA kernels' stream geometry reach, by access width and residency?
// Every 256-thread workgroup streams its own contiguous run in blocks of 5120 frames (20 KiB in, 20 KiB out), the next
// block's loads in flight while the current one is stored -- the real kernel's pattern without the transform.
//   mode 0: 8 B per lane loads (thread t <- frames pair t + 256 k), 8 B stores          (the round-1 kernel)
//   mode 1: 16 B per lane loads and stores (even lanes row 2m, odd lanes row 2m+1)
//   mode 2: LDS-DMA 16 B per lane into a per-wave LDS slot, ds_read_b64, 8 B stores
//   mode 3: LDS-DMA in, 16 B stores
// `filler` v_pk_fma_f32 per thread and block stand in for the transform (0 = pure stream).
// Workgroups per CU are set with dynamic LDS.   build: hipcc --offload-arch=gfx950 -O3 -w stream_geom.hip -o stream_geom
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int HS = 10, BLK = 512 * HS;

__device__ __forceinline__ void glds16(const void* g, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_byte) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ in, float* __restrict__ out, long run, int filler) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const float* src = in + (long)blockIdx.x * run;
  float* dst = out + (long)blockIdx.x * run;
  const int nb = (int)(run / BLK);
  f2 acc = {1.f + t, 2.f};
  const f2 pc = {1.0001f, 0.9999f}, pd = {0.5f, 0.25f};
  if (MODE == 0) {
    f2 nx[HS], cur[HS];
    for (int k2 = 0; k2 < HS; ++k2) nx[k2] = ((const f2*)src)[t + 256 * k2];
    for (int b = 0; b < nb; ++b) {
#pragma unroll
      for (int k2 = 0; k2 < HS; ++k2) cur[k2] = nx[k2];
      for (int i = 0; i < filler / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(pc), "v"(pd));
      if (b + 1 < nb) {
#pragma unroll
        for (int k2 = 0; k2 < HS; ++k2) nx[k2] = ((const f2*)(src + (long)(b + 1) * BLK))[t + 256 * k2];
      }
      for (int i = 0; i < filler / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(pc), "v"(pd));
#pragma unroll
      for (int k2 = 0; k2 < HS; ++k2) ((f2*)(dst + (long)b * BLK))[t + 256 * k2] = cur[k2] * acc.y;
    }
  } else if (MODE == 1) {
    // lane l: row 2m + (l & 1), samples (64 w + (l & ~1)) .. +1   -> float4 index within the block
    const int r = l & 1, s2 = (64 * w + (l & ~1)) / 2;
    f4 nx[HS / 2], cur[HS / 2];
    for (int m = 0; m < HS / 2; ++m) nx[m] = ((const f4*)src)[128 * (2 * m + r) + s2];
    for (int b = 0; b < nb; ++b) {
#pragma unroll
      for (int m = 0; m < HS / 2; ++m) cur[m] = nx[m];
      for (int i = 0; i < filler / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(pc), "v"(pd));
      if (b + 1 < nb) {
#pragma unroll
        for (int m = 0; m < HS / 2; ++m) nx[m] = ((const f4*)(src + (long)(b + 1) * BLK))[128 * (2 * m + r) + s2];
      }
      for (int i = 0; i < filler / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(pc), "v"(pd));
#pragma unroll
      for (int m = 0; m < HS / 2; ++m) ((f4*)(dst + (long)b * BLK))[128 * (2 * m + r) + s2] = cur[m] * acc.y;
    }
  } else {
    // per-wave slot: [HS rows][64 samples] f2 = 5 KiB; one DMA instruction = rows (2m, 2m+1): lanes 0..31 row 2m
    const unsigned slot = (unsigned)(size_t)(dyn) + 0; // LDS byte address of dyn (address space 3 pointer value)
    const unsigned wbase = (unsigned)__builtin_amdgcn_readfirstlane((int)(w * HS * 512));
    const int hr = l >> 5, ls = 2 * (l & 31);
    auto issue = [&](const float* blk) {
#pragma unroll
      for (int m = 0; m < HS / 2; ++m) {
        const float* g = blk + 2 * (256 * (2 * m + hr) + 64 * w + ls);
        glds16(g, __builtin_amdgcn_readfirstlane((int)(slot + wbase + m * 1024)));
      }
    };
    issue(src);
    const f2* lr = (const f2*)(dyn + w * HS * 512) + l;
    for (int b = 0; b < nb; ++b) {
      f2 cur[HS];
      if (MODE == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(b == 0 ? 0 : 0) : "memory"); // first block: nothing younger
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k2 = 0; k2 < HS; ++k2) cur[k2] = ((volatile const f2*)lr)[64 * k2];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (b + 1 < nb) issue(src + (long)(b + 1) * BLK);
      for (int i = 0; i < filler; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(pc), "v"(pd));
      if (MODE == 2) {
#pragma unroll
        for (int k2 = 0; k2 < HS; ++k2) ((f2*)(dst + (long)b * BLK))[t + 256 * k2] = cur[k2] * acc.y;
      } else {
        // pairs of rows to 16-B stores (values not exchanged: bandwidth probe only)
        const int r = l & 1, s2 = (64 * w + (l & ~1)) / 2;
#pragma unroll
        for (int m = 0; m < HS / 2; ++m) {
          f4 v = {cur[2 * m].x, cur[2 * m].y, cur[2 * m + 1].x, cur[2 * m + 1].y};
          ((f4*)(dst + (long)b * BLK))[128 * (2 * m + r) + s2] = v * acc.y;
        }
      }
    }
  }
}

int main(int argc, char** argv) {
  const long n = 1L << 27;
  float *in, *out;
  hipMalloc(&in, (n + (1 << 22)) * 4); hipMalloc(&out, (n + (1 << 22)) * 4);
  hipMemset(in, 0x3c, n * 4);
  typedef void (*kern_t)(const float*, float*, long, int);
  kern_t ks[4] = {k<0>, k<1>, k<2>, k<3>};
  for (int i = 0; i < 4; ++i) hipFuncSetAttribute((const void*)ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const bool scan = argc > 1;  // any argument: scan the run length (pure stream, 3 workgroups per CU) instead of the mode table
  if (scan) {
    const int wgs = 3, grid = 256 * wgs;
    const unsigned lds = (160 * 1024 / wgs) & ~1023u;
    for (int nb = 18; nb <= 34; ++nb) { if ((long)nb * BLK * grid > n) break;   // blocks of 20 KiB per run: 24..40 (the whole buffer is not covered: rate only)
      const long run = (long)nb * BLK;
      for (int mode = 0; mode < 4; mode += 3) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
          hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
          hipEventRecord(e0);
          hipLaunchKernelGGL(ks[mode], dim3(grid), dim3(256), lds, 0, in, out, run, 0);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep > 0 && ms < best) best = ms;
        }
        printf("run %2d blocks = %8ld B (mod 1 MiB %7ld) mode %d: %.4f ms  %.2f TB/s\n", nb, run * 4, (run * 4) % (1 << 20), mode, best, 8.0 * run * grid / best * 1e-9);
      }
    }
    printf("status %s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
  }
  const int fillers[3] = {0, 330, 660};
  for (int wgs = 1; wgs <= 4; ++wgs) {
    const unsigned lds = (160 * 1024 / wgs) & ~1023u;          // forces `wgs` workgroups per CU
    const int grid = 256 * wgs;
    long run = n / grid / BLK * BLK;
    for (int f = 0; f < 3; ++f) {
      for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
          hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
          hipEventRecord(e0);
          hipLaunchKernelGGL(ks[mode], dim3(grid), dim3(256), lds - (mode < 2 ? 0 : 0), 0, in, out, run, fillers[f]);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep > 0 && ms < best) best = ms;
        }
        const double bytes = 8.0 * (double)run * grid;
        printf("wg/CU %d filler %3d mode %d: %.4f ms  %.2f TB/s\n", wgs, fillers[f], mode, best, bytes / best * 1e-9);
      }
    }
  }
  hipError_t e = hipDeviceSynchronize();
  printf("status %s\n", hipGetErrorString(e));
  return 0;
}
