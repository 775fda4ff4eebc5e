// What an hipEventRecord behind every launch costs on back-to-back short launches (round 5: the engine marked the caller's stream
// with an event at the end of every call: +3 us per launch on BASELINE config 3's 11 us launches).
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/event_cost tools/ubench/event_cost.hip && tools/ubench/event_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin (float* p, int n)
{
	float a = p[threadIdx.x];
	for (int i = 0; i < n; ++i) {
		a = a * 1.0001f + 0.5f;
	}
	p[threadIdx.x] = a;
}

static double run (hipStream_t st, float* d, int n_it, int mode, hipEvent_t ev)
{
	const int N = 4000;
	for (int i = 0; i < 200; ++i) {
		spin<<<256, 256, 0, st>>> (d, n_it);
	}
	(void)hipStreamSynchronize (st);
	auto t0 = std::chrono::steady_clock::now ();
	for (int i = 0; i < N; ++i) {
		spin<<<256, 256, 0, st>>> (d, n_it);
		if (mode) {
			(void)hipEventRecord (ev, st);
		}
	}
	(void)hipStreamSynchronize (st);
	return std::chrono::duration<double, std::micro> (std::chrono::steady_clock::now () - t0).count () / N;
}

int main ()
{
	float* d;
	(void)hipMalloc (&d, 4096);
	(void)hipMemset (d, 0, 4096);
	hipStream_t st;
	(void)hipStreamCreateWithFlags (&st, hipStreamNonBlocking);
	hipEvent_t e0, e1, e2;
	(void)hipEventCreateWithFlags (&e0, hipEventDisableTiming);
	(void)hipEventCreateWithFlags (&e1, hipEventDisableTiming | hipEventReleaseToDevice);
	(void)hipEventCreateWithFlags (&e2, hipEventDefault);
	for (int n_it : {200, 2000, 6000}) {
		printf ("spin %5d: none %.2f us  disable-timing %.2f  +release-to-device %.2f  default %.2f\n", n_it, run (st, d, n_it, 0, e0),
		        run (st, d, n_it, 1, e0), run (st, d, n_it, 1, e1), run (st, d, n_it, 1, e2));
	}
	return 0;
}
