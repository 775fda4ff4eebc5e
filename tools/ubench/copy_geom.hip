// micro-benchmark: why does the OLA kernels' stream geometry copy at 5.0-5.3 TB/s when the guide's
// float4 copy reaches 6.29 TB/s?  One 2^27-frame fp32 stream in, one out (1.07 GB per launch), varied one thing at a time:
//   A  classic copy: one float4 per thread, one-shot grid (the guide's shape)
//   B  persistent grid, block-cyclic: workgroup g copies blocks g, g + G, g + 2G ... (every workgroup in the same 15 MiB window)
//   C  persistent grid, contiguous runs (the OLA kernels): workgroup g copies blocks [g R, (g + 1) R)
//   D  C with runs of K blocks dealt block-cyclically (K consecutive blocks per visit)
// each with W workgroups per CU, a block of ROWS x 2 KiB, the next block's loads issued `ahead` blocks in advance
// (register prefetch, as the kernels do), 8 B or 16 B per lane; also the read-only and write-only halves of C.
//   build: hipcc --offload-arch=gfx950 -O3 -w copy_geom.hip -o copy_geom
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>
#include <type_traits>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_classic(const f4* __restrict__ in, f4* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  out[i] = in[i];
}
__global__ __launch_bounds__(256) void copy_classic4(const f4* __restrict__ in, f4* __restrict__ out) {
  const long i = (long)blockIdx.x * 1024 + threadIdx.x;   // four float4 per thread, 256 apart
  f4 a = in[i], b = in[i + 256], c = in[i + 512], d = in[i + 768];
  out[i] = a; out[i + 256] = b; out[i + 512] = c; out[i + 768] = d;
}

// ROWS rows of 256 x 8 B (or 128 x 16 B lanes-wise: W16) per block; block index sequence given by (first, step, count)
// WHAT: 0 copy, 1 read only (sum kept), 2 write only
template <int ROWS, bool W16, int WHAT, int AHEAD>
__global__ __launch_bounds__(256) void copy_blocks(const float* __restrict__ in, float* __restrict__ out, int mode, long nblk, int K) {
  extern __shared__ char dyn[];
  constexpr int BLK = 512 * ROWS;  // frames per block
  constexpr int NR = W16 ? ROWS / 2 : ROWS;
  const int t = threadIdx.x, g = blockIdx.x, G = gridDim.x;
  // block sequence
  long per = (nblk + G - 1) / G;
  auto blk_of = [&](long i) -> long {   // i-th block of this workgroup, or -1
    if (mode == 1) { long b = (long)g + i * G; return b < nblk ? b : -1; }                         // block-cyclic
    if (mode == 2) { long b = (long)g * per + i; return (i < per && b < nblk) ? b : -1; }           // contiguous runs
    long visit = i / K, j = i % K; long b = (visit * G + g) * K + j; return b < nblk ? b : -1;    // runs of K, dealt cyclically
  };
  typedef typename std::conditional<W16, f4, f2>::type V;
  V nx[AHEAD][NR];
  f2 acc = {0.f, 0.f};
  auto load = [&](long b, V (&r)[NR]) {
    const V* s = (const V*)(in + b * BLK);
#pragma unroll
    for (int k = 0; k < NR; ++k) r[k] = (WHAT == 2) ? V{} : s[t + 256 * k];
  };
#pragma unroll
  for (int a = 0; a < AHEAD; ++a) { long b = blk_of(a); if (b >= 0) load(b, nx[a]); }
  for (long i = 0;; ++i) {
    long b = blk_of(i);
    if (b < 0) break;
    V cur[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) cur[k] = nx[0][k];
#pragma unroll
    for (int a = 0; a + 1 < AHEAD; ++a)
#pragma unroll
      for (int k = 0; k < NR; ++k) nx[a][k] = nx[a + 1][k];
    long bn = blk_of(i + AHEAD);
    if (bn >= 0) load(bn, nx[AHEAD - 1]);
    if (WHAT == 1) {
#pragma unroll
      for (int k = 0; k < NR; ++k) { acc.x += cur[k][0]; acc.y += cur[k][1]; }
    } else {
      V* d = (V*)(out + b * BLK);
#pragma unroll
      for (int k = 0; k < NR; ++k) d[t + 256 * k] = cur[k];
    }
  }
  if (WHAT == 1 && acc.x == 12345.678f) out[t] = acc.y;
}

// E: contiguous runs, row by row: the load of row r + D goes out, then the store of row r (a steady 1:1 mix of loads and
// stores instead of bursts of ROWS loads and ROWS stores); D rows of 256 x 8 B in flight per workgroup
template <int D>
__global__ __launch_bounds__(256) void copy_rowpipe(const float* __restrict__ in, float* __restrict__ out, long rows_per_wg) {
  extern __shared__ char dyn[];
  const int t = threadIdx.x;
  const f2* s = (const f2*)in + (long)blockIdx.x * rows_per_wg * 256;
  f2* d = (f2*)out + (long)blockIdx.x * rows_per_wg * 256;
  f2 ring[D];
#pragma unroll
  for (int i = 0; i < D; ++i) ring[i] = s[t + 256 * i];
  long r = 0;
  for (; r + D <= rows_per_wg - D; r += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      f2 cur = ring[i];
      ring[i] = s[t + 256 * (r + D + i)];
      d[t + 256 * (r + i)] = cur;
    }
  }
  for (; r < rows_per_wg; ++r) {
    f2 cur = s[t + 256 * r];
    d[t + 256 * r] = cur;
  }
}

// F: does it matter WHICH XCD touches an address?  The one-shot copy deals 4 KiB chunks round-robin over the 8 XCDs (workgroup
// i -> XCD i % 8), i.e. XCD k only ever touches chunks = k (mod 8).  Persistent workgroups, workgroup g on XCD g % 8 (observed
// round-robin placement), each copying chunks of CS bytes whose index is = (g + shift) (mod 8): shift 0 reproduces the one-shot
// copy's address -> XCD map, shift 4 its opposite; "mix": chunk index = anything (plain block-cyclic).  float4 per lane,
// DEPTH chunks in flight.
template <int CSV, int DEPTH>   // CSV float4 per thread per chunk: chunk = CSV * 4 KiB
__global__ __launch_bounds__(256) void copy_xcd(const f4* __restrict__ in, f4* __restrict__ out, long nchunks, int shift, int mix) {
  extern __shared__ char dyn[];
  const int t = threadIdx.x, g = blockIdx.x, G = gridDim.x;
  auto chunk_of = [&](long i) -> long {
    long c = mix ? (i * G + g) : ((i * (G / 8) + g / 8) * 8 + ((g + shift) & 7));
    return c < nchunks ? c : -1;
  };
  f4 ring[DEPTH][CSV];
#pragma unroll
  for (int a = 0; a < DEPTH; ++a) { long c = chunk_of(a); if (c >= 0) {
#pragma unroll
    for (int k = 0; k < CSV; ++k) ring[a][k] = in[(c * CSV + k) * 256 + t]; } }
  for (long i = 0;; i += DEPTH) {
#pragma unroll
    for (int a = 0; a < DEPTH; ++a) {
      long c = chunk_of(i + a);
      if (c < 0) return;
      f4 cur[CSV];
#pragma unroll
      for (int k = 0; k < CSV; ++k) cur[k] = ring[a][k];
      long cn = chunk_of(i + a + DEPTH);
      if (cn >= 0) {
#pragma unroll
        for (int k = 0; k < CSV; ++k) ring[a][k] = in[(cn * CSV + k) * 256 + t];
      }
#pragma unroll
      for (int k = 0; k < CSV; ++k) out[(c * CSV + k) * 256 + t] = cur[k];
    }
  }
}

static float time_it(std::function<void()> f) {
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  return best;
}

template <int ROWS, bool W16, int WHAT, int AHEAD>
static void run_cfg(const float* in, float* out, long n, int wgs, int mode, int K, const char* tag) {
  auto kern = copy_blocks<ROWS, W16, WHAT, AHEAD>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const unsigned lds = wgs >= 8 ? 0 : ((160 * 1024 / wgs) & ~1023u);
  const long nblk = n / (512 * ROWS);
  const int grid = 256 * wgs;
  float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, in, out, mode, nblk, K); });
  const double bytes = (WHAT == 0 ? 8.0 : 4.0) * (double)nblk * 512 * ROWS;
  printf("%-10s rows %2d %2dB what %d ahead %d wg/CU %d mode %d K %2d : %.4f ms  %.2f TB/s\n", tag, ROWS, W16 ? 16 : 8, WHAT, AHEAD, wgs, mode, K, ms,
         bytes / ms * 1e-9);
  fflush(stdout);
}

int main() {
  const long n = 1L << 27;
  float *in, *out;
  hipMalloc(&in, n * 4 + (1 << 24)); hipMalloc(&out, n * 4 + (1 << 24));
  hipMemset(in, 0x3c, n * 4); hipMemset(out, 0, n * 4);
  // warm the clocks
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(copy_classic, dim3(n / 4 / 256), dim3(256), 0, 0, (const f4*)in, (f4*)out);
  hipDeviceSynchronize();
  float ms = time_it([&] { hipLaunchKernelGGL(copy_classic, dim3(n / 4 / 256), dim3(256), 0, 0, (const f4*)in, (f4*)out); });
  printf("A classic float4, 1 per thread: %.4f ms %.2f TB/s\n", ms, 8.0 * n / ms * 1e-9);
  ms = time_it([&] { hipLaunchKernelGGL(copy_classic4, dim3(n / 4 / 1024), dim3(256), 0, 0, (const f4*)in, (f4*)out); });
  printf("A classic float4, 4 per thread: %.4f ms %.2f TB/s\n", ms, 8.0 * n / ms * 1e-9);
  ms = time_it([&] { hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, 0); });
  printf("A hipMemcpy D2D: %.4f ms %.2f TB/s\n", ms, 8.0 * n / ms * 1e-9);
  for (int wgs = 2; wgs <= 8; wgs += (wgs < 4 ? 1 : 2)) {
    run_cfg<10, false, 0, 1>(in, out, n, wgs, 1, 1, "B cyclic");
    run_cfg<10, false, 0, 1>(in, out, n, wgs, 2, 1, "C runs");
    run_cfg<10, false, 0, 1>(in, out, n, wgs, 3, 4, "D runs4");
    run_cfg<10, false, 0, 1>(in, out, n, wgs, 3, 16, "D runs16");
  }
  for (int wgs = 3; wgs <= 4; ++wgs) {
    run_cfg<10, true, 0, 1>(in, out, n, wgs, 1, 1, "B cyclic");
    run_cfg<10, true, 0, 1>(in, out, n, wgs, 2, 1, "C runs");
    run_cfg<10, false, 0, 2>(in, out, n, wgs, 1, 1, "B cyclic");
    run_cfg<10, false, 0, 2>(in, out, n, wgs, 2, 1, "C runs");
    run_cfg<16, false, 0, 1>(in, out, n, wgs, 1, 1, "B cyclic");
    run_cfg<16, false, 0, 1>(in, out, n, wgs, 2, 1, "C runs");
    run_cfg<4, false, 0, 1>(in, out, n, wgs, 1, 1, "B cyclic");
    run_cfg<4, false, 0, 1>(in, out, n, wgs, 2, 1, "C runs");
    run_cfg<4, false, 0, 2>(in, out, n, wgs, 2, 1, "C runs");
    run_cfg<10, false, 1, 1>(in, out, n, wgs, 1, 1, "B read");
    run_cfg<10, false, 1, 1>(in, out, n, wgs, 2, 1, "C read");
    run_cfg<10, false, 2, 1>(in, out, n, wgs, 1, 1, "B write");
    run_cfg<10, false, 2, 1>(in, out, n, wgs, 2, 1, "C write");
  }
  for (int wgs = 2; wgs <= 8; wgs *= 2) {
    const unsigned lds = wgs >= 8 ? 0 : ((160 * 1024 / wgs) & ~1023u);
    const int grid = 256 * wgs;
    const long rows = (n / 512) / grid;
#define ROWPIPE(D_) { auto kern = copy_rowpipe<D_>; hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, in, out, rows); }); \
      printf("E rowpipe D %2d wg/CU %d: %.4f ms  %.2f TB/s\n", D_, wgs, ms, 8.0 * rows * 512 * grid / ms * 1e-9); fflush(stdout); }
    ROWPIPE(4) ROWPIPE(8) ROWPIPE(10) ROWPIPE(16) ROWPIPE(24)
  }
#define XCDRUN(CSV_, DEPTH_, wgs, shift, mix) { auto kern = copy_xcd<CSV_, DEPTH_>; hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      const unsigned lds = wgs >= 8 ? 0 : ((160 * 1024 / wgs) & ~1023u); const long nch = n * 4 / (CSV_ * 4096); \
      float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3(256 * wgs), dim3(256), lds, 0, (const f4*)in, (f4*)out, nch, shift, mix); }); \
      printf("F xcd chunk %2d KiB depth %d wg/CU %d shift %d mix %d: %.4f ms  %.2f TB/s\n", 4 * CSV_, DEPTH_, wgs, shift, mix, ms, 8.0 * n / ms * 1e-9); fflush(stdout); }
  for (int wgs = 4; wgs <= 8; wgs += 4) {
    XCDRUN(1, 4, wgs, 0, 0) XCDRUN(1, 4, wgs, 4, 0) XCDRUN(1, 4, wgs, 1, 0) XCDRUN(1, 4, wgs, 0, 1)
    XCDRUN(1, 8, wgs, 0, 0) XCDRUN(1, 8, wgs, 4, 0)
    XCDRUN(2, 4, wgs, 0, 0) XCDRUN(2, 4, wgs, 4, 0) XCDRUN(2, 4, wgs, 0, 1)
    XCDRUN(4, 2, wgs, 0, 0) XCDRUN(4, 2, wgs, 4, 0)
  }
  printf("status %s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
