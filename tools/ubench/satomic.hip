// does gfx950 execute scalar atomics (s_atomic_add ... glc: a wave-uniform fetch-and-add straight into an SGPR, through the
// scalar data cache, counted by lgkmcnt -- independent of the vector memory queue)?  Every wave of a full-chip launch claims
// tickets until the counter passes N and marks them; the host checks that each ticket was claimed exactly once and prints
// the cycles per claim against a lane-0 global_atomic_add (vmcnt).   hipcc --offload-arch=gfx950 -O3 tools/ubench/satomic.hip -o tools/ubench/satomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SCALAR>
__global__ __launch_bounds__ (512) void claim (unsigned* ctr, unsigned* mark, unsigned n, unsigned long long* cyc)
{
	unsigned long long t0 = __builtin_readcyclecounter (), claims = 0;
	for (;;) {
		unsigned t;
		if (SCALAR) {
			asm volatile ("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(ctr), "0"(1u) : "memory");
		} else {
			unsigned v = 0;
			if ((threadIdx.x & 63) == 0) {
				v = atomicAdd (ctr, 1u);
			}
			t = __builtin_amdgcn_readfirstlane (v);
		}
		++claims;
		if (t >= n) {
			break;
		}
		if ((threadIdx.x & 63) == 0) {
			atomicAdd (&mark[t], 1u);
		}
		__builtin_amdgcn_s_sleep (8);
	}
	if ((threadIdx.x & 63) == 0) {
		atomicAdd (&cyc[0], __builtin_readcyclecounter () - t0);
		atomicAdd (&cyc[1], claims);
	}
}

int main ()
{
	const unsigned n = 1u << 20;
	unsigned *ctr, *mark;
	unsigned long long* cyc;
	hipMalloc (&ctr, 4);
	hipMalloc (&mark, 4 * n);
	hipMalloc (&cyc, 16);
	for (int scalar = 0; scalar < 2; ++scalar) {
		hipMemset (ctr, 0, 4);
		hipMemset (mark, 0, 4 * n);
		hipMemset (cyc, 0, 16);
		hipEvent_t e0, e1;
		hipEventCreate (&e0);
		hipEventCreate (&e1);
		hipEventRecord (e0);
		if (scalar) {
			hipLaunchKernelGGL (claim<1>, dim3 (512), dim3 (512), 0, 0, ctr, mark, n, cyc);
		} else {
			hipLaunchKernelGGL (claim<0>, dim3 (512), dim3 (512), 0, 0, ctr, mark, n, cyc);
		}
		hipEventRecord (e1);
		hipError_t err = hipDeviceSynchronize ();
		float ms = 0;
		hipEventElapsedTime (&ms, e0, e1);
		std::vector<unsigned> h (n);
		hipMemcpy (h.data (), mark, 4 * n, hipMemcpyDeviceToHost);
		unsigned long long hc[2];
		hipMemcpy (hc, cyc, 16, hipMemcpyDeviceToHost);
		unsigned bad = 0;
		for (unsigned i = 0; i < n; ++i) {
			bad += h[i] != 1;
		}
		printf ("%s: %s, %u tickets, %u not claimed exactly once, %.3f ms, %.0f cycles per claim and wave (with the sleep and the mark)\n",
		        scalar ? "s_atomic_add" : "lane-0 global atomic", hipGetErrorString (err), n, bad, ms, (double)hc[0] / (double)hc[1]);
	}
	return 0;
}
