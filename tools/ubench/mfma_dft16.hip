// micro-benchmark: what a radix-16 pass would cost on the matrix core.
//   A radix-16 DFT of 16 complex values per column = a real 32 x 32 matrix (the DFT, re/im interleaved) times a 32 x N panel
//   of columns.  For one wave (64 columns) that is 2 (N halves) x 2 (K halves) = 4 v_mfma_f32_32x32x16_f16 per pass with half
//   inputs (the fp16 sample path), or 3 x 4 = 12 with fp32 data split into hi + lo halves (hi.hi, hi.lo, lo.hi).
//   Measured here: (a) the issue rate of that MFMA alone; (b) a pass as the fp16 path would run it -- 4 MFMAs, the results
//   converted back to half (8 v_cvt_pk) and multiplied by per-element twiddles (16 complex x 3 v_pk_*_f16) -- against
//   (c) the same pass on the VALU (the ~100 packed instructions per pass of prot_fft.h, issued as independent v_pk_fma_f16);
//   (d) the split-precision fp32 pass: 12 MFMAs + 32 x (cvt hi, back-convert, subtract, cvt lo) + fp32 twiddles (16 x 3 v_pk_*_f32).
// hipcc --offload-arch=gfx950 -O3 -o mfma_dft16 mfma_dft16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__ ((ext_vector_type (16)));
typedef _Float16 h8 __attribute__ ((ext_vector_type (8)));
typedef _Float16 h2 __attribute__ ((ext_vector_type (2)));
typedef float f2 __attribute__ ((ext_vector_type (2)));

template <int KIND>
__global__ __launch_bounds__ (256) void k (float* out, int iters)
{
	const float s0 = threadIdx.x * 1e-3f;
	h8 a0, a1, b[4];
	for (int i = 0; i < 8; ++i) {
		a0[i] = (_Float16)(0.5f + s0 + i);
		a1[i] = (_Float16)(0.25f - s0 + i);
		for (int q = 0; q < 4; ++q) {
			b[q][i] = (_Float16)(s0 * (q + 1) + 0.125f * i);
		}
	}
	f32x16 acc0, acc1;
	for (int r = 0; r < 16; ++r) {
		acc0[r] = 0.f;
		acc1[r] = 0.f;
	}
	h2 tw[16];
	f2 twf[16];
	for (int i = 0; i < 16; ++i) {
		tw[i]  = h2{ (_Float16)(0.9f + 0.001f * i), (_Float16)(0.1f * i) };
		twf[i] = f2{ 0.9f + 0.001f * i, 0.1f * i };
	}
	float sink = 0.f;
	for (int it = 0; it < iters; ++it) {
		if (KIND == 0) { // MFMA alone: 4 per pass, two accumulator chains
			acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a0, b[0], acc0, 0, 0, 0);
			acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a0, b[1], acc1, 0, 0, 0);
			acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a1, b[2], acc0, 0, 0, 0);
			acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a1, b[3], acc1, 0, 0, 0);
		} else if (KIND == 1) { // the fp16 path's pass on the matrix core
			f32x16 z;
			for (int r = 0; r < 16; ++r) {
				z[r] = 0.f;
			}
			f32x16 r0 = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a0, b[0], z, 0, 0, 0);
			r0        = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a1, b[1], r0, 0, 0, 0);
			f32x16 r1 = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a0, b[2], z, 0, 0, 0);
			r1        = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a1, b[3], r1, 0, 0, 0);
			// 32 reals of this lane = 16 complex: back to half, times twiddle (3 packed-half instructions each), into the
			// next pass's B operands
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				h2 v0 = __builtin_bit_cast (h2, __builtin_amdgcn_cvt_pkrtz (r0[2 * i], r0[2 * i + 1]));
				h2 v1 = __builtin_bit_cast (h2, __builtin_amdgcn_cvt_pkrtz (r1[2 * i], r1[2 * i + 1]));
				h2 w0 = h2{ tw[i].x, tw[i].x } * v0 + h2{ -tw[i].y, tw[i].y } * h2{ v0.y, v0.x };
				h2 w1 = h2{ tw[8 + i].x, tw[8 + i].x } * v1 + h2{ -tw[8 + i].y, tw[8 + i].y } * h2{ v1.y, v1.x };
				b[i & 3][2 * (i >> 2)]     = w0.x;
				b[i & 3][2 * (i >> 2) + 1] = w0.y;
				b[i & 3][4 + 2 * (i >> 2)]     = w1.x;
				b[i & 3][4 + 2 * (i >> 2) + 1] = w1.y;
			}
		} else if (KIND == 2) { // the same pass on the VALU: ~100 packed-half instructions
			h2 v[16];
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				v[i] = h2{ b[i & 3][(i >> 2) * 2], b[i & 3][(i >> 2) * 2 + 1] };
			}
#pragma unroll
			for (int r = 0; r < 6; ++r) {
#pragma unroll
				for (int i = 0; i < 16; ++i) {
					v[i] = v[i] * tw[(i + r) & 15] + v[(i + 1 + r) & 15];
				}
			}
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				b[i & 3][(i >> 2) * 2]     = v[i].x;
				b[i & 3][(i >> 2) * 2 + 1] = v[i].y;
			}
		} else { // split-precision fp32 pass: 12 MFMAs, hi/lo split of 32 reals, fp32 twiddles
			f32x16 z;
			for (int r = 0; r < 16; ++r) {
				z[r] = 0.f;
			}
			f32x16 r0 = z, r1 = z;
#pragma unroll
			for (int t = 0; t < 3; ++t) { // hi.hi, hi.lo, lo.hi (operands stand in for each other: the rate is what is measured)
				r0 = __builtin_amdgcn_mfma_f32_32x32x16_f16 (t == 2 ? a1 : a0, b[(0 + t) & 3], r0, 0, 0, 0);
				r0 = __builtin_amdgcn_mfma_f32_32x32x16_f16 (t == 2 ? a0 : a1, b[(1 + t) & 3], r0, 0, 0, 0);
				r1 = __builtin_amdgcn_mfma_f32_32x32x16_f16 (t == 2 ? a1 : a0, b[(2 + t) & 3], r1, 0, 0, 0);
				r1 = __builtin_amdgcn_mfma_f32_32x32x16_f16 (t == 2 ? a0 : a1, b[(3 + t) & 3], r1, 0, 0, 0);
			}
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				f2 v0 = f2{ r0[2 * i], r0[2 * i + 1] }, v1 = f2{ r1[2 * i], r1[2 * i + 1] };
				f2 w0 = f2{ twf[i].x, twf[i].x } * v0 + f2{ -twf[i].y, twf[i].y } * f2{ v0.y, v0.x };
				f2 w1 = f2{ twf[8 + i].x, twf[8 + i].x } * v1 + f2{ -twf[8 + i].y, twf[8 + i].y } * f2{ v1.y, v1.x };
				// hi = half(w), lo = half(w - float(hi)): 2 cvt_pk + 1 packed subtract per complex, plus the back-conversion
				h2 hi0 = __builtin_bit_cast (h2, __builtin_amdgcn_cvt_pkrtz (w0.x, w0.y));
				h2 hi1 = __builtin_bit_cast (h2, __builtin_amdgcn_cvt_pkrtz (w1.x, w1.y));
				f2 d0  = w0 - f2{ (float)hi0.x, (float)hi0.y };
				f2 d1  = w1 - f2{ (float)hi1.x, (float)hi1.y };
				h2 lo0 = __builtin_bit_cast (h2, __builtin_amdgcn_cvt_pkrtz (d0.x, d0.y));
				h2 lo1 = __builtin_bit_cast (h2, __builtin_amdgcn_cvt_pkrtz (d1.x, d1.y));
				b[i & 3][2 * (i >> 2)]         = hi0.x;
				b[i & 3][2 * (i >> 2) + 1]     = hi0.y;
				b[i & 3][4 + 2 * (i >> 2)]     = hi1.x;
				b[i & 3][4 + 2 * (i >> 2) + 1] = hi1.y;
				a0[i] = lo0.x + lo1.x;
				a1[i] = lo0.y + lo1.y;
			}
		}
	}
	for (int r = 0; r < 16; ++r) {
		sink += acc0[r] + acc1[r];
	}
	for (int q = 0; q < 4; ++q) {
		for (int i = 0; i < 8; ++i) {
			sink += (float)b[q][i];
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = sink + (float)a0[0] + (float)a1[0];
}

template <int KIND>
void run (const char* name, float* d)
{
	for (int wps : { 1, 2, 4 }) {
		const int  blocks = 256 * wps, iters = 4000;
		hipEvent_t e0, e1;
		hipEventCreate (&e0);
		hipEventCreate (&e1);
		hipLaunchKernelGGL ((k<KIND>), dim3 (blocks), dim3 (256), 0, 0, d, 200);
		hipEventRecord (e0);
		hipLaunchKernelGGL ((k<KIND>), dim3 (blocks), dim3 (256), 0, 0, d, iters);
		hipEventRecord (e1);
		hipEventSynchronize (e1);
		float ms;
		hipEventElapsedTime (&ms, e0, e1);
		const double passes_per_simd = (double)blocks * 4 * iters / 1024.0;
		printf ("%-44s waves/SIMD %d: %.3f ms = %.1f ns per wave-pass per SIMD\n", name, wps, ms, ms * 1e6 / passes_per_simd);
	}
}

int main ()
{
	float* d;
	hipMalloc (&d, 256 * 4 * 256 * 4);
	run<0> ("4 x v_mfma_f32_32x32x16_f16 alone", d);
	run<1> ("fp16 pass on the matrix core (4 MFMA + cvt + tw)", d);
	run<2> ("fp16 pass on the VALU (96 v_pk_fma_f16)", d);
	run<3> ("fp32 split pass (12 MFMA + split + tw)", d);
	return 0;
}
