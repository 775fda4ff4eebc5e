// prot_fft.h -- register-resident radix-16 butterflies for the gfx950 OLA
// kernels.  A 4096-point complex FFT is three radix-16 passes; each of the
// 256 threads of a workgroup keeps 16 complex points (32 VGPRs) and the two
// digit transpositions between passes go through LDS (prot_kernels.hip).
//
// What it replaces: FFTW's r2c/c2r codelets behind fftwf_execute_dft_r2c /
// fftwf_execute_dft_c2r (call sites src/phaserotate.c:648,657 and
// cli/phase-rotate.cc:193,205).  Two real blocks ride in one complex
// transform (re = block A, im = block B); the FIR is real, so filtering acts
// on the two parts independently and no even/odd split pass is needed.
#pragma once
#include <hip/hip_runtime.h>

namespace prot {

#define PROT_DEV static __device__ __forceinline__

// a * w for DIR > 0, a * conj(w) for DIR < 0   (w = forward twiddle e^{-i phi})
template <int DIR>
PROT_DEV float2 cmulw (float2 a, float2 w)
{
	if (DIR > 0) {
		return make_float2 (a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
	} else {
		return make_float2 (a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y);
	}
}

// 4-point DFT, in place.  DIR > 0: e^{-2 pi i nk/4}; DIR < 0: conjugate kernel.
template <int DIR>
PROT_DEV void radix4 (float2& a, float2& b, float2& c, float2& d)
{
	const float2 t0 = make_float2 (a.x + c.x, a.y + c.y);
	const float2 t1 = make_float2 (a.x - c.x, a.y - c.y);
	const float2 t2 = make_float2 (b.x + d.x, b.y + d.y);
	const float2 t3 = make_float2 (b.x - d.x, b.y - d.y);
	a               = make_float2 (t0.x + t2.x, t0.y + t2.y);
	c               = make_float2 (t0.x - t2.x, t0.y - t2.y);
	if (DIR > 0) { // X1 = t1 - j t3, X3 = t1 + j t3
		b = make_float2 (t1.x + t3.y, t1.y - t3.x);
		d = make_float2 (t1.x - t3.y, t1.y + t3.x);
	} else {
		b = make_float2 (t1.x - t3.y, t1.y + t3.x);
		d = make_float2 (t1.x + t3.y, t1.y - t3.x);
	}
}

// multiply by W16^m (forward) or its conjugate, m a compile-time constant
template <int DIR, int M>
PROT_DEV float2 mul_w16 (float2 a)
{
	constexpr float C1 = 0.92387953251128675613f; // cos(pi/8)
	constexpr float S1 = 0.38268343236508977173f; // sin(pi/8)
	constexpr float R  = 0.70710678118654752440f; // sqrt(1/2)
	constexpr int   m  = M & 15;
	if (m == 0) {
		return a;
	} else if (m == 4) { // -j (fwd) / +j (inv)
		return DIR > 0 ? make_float2 (a.y, -a.x) : make_float2 (-a.y, a.x);
	} else if (m == 8) {
		return make_float2 (-a.x, -a.y);
	} else if (m == 12) {
		return DIR > 0 ? make_float2 (-a.y, a.x) : make_float2 (a.y, -a.x);
	} else if (m == 2) { // R(1 - j)
		return DIR > 0 ? make_float2 (R * (a.x + a.y), R * (a.y - a.x))
		               : make_float2 (R * (a.x - a.y), R * (a.y + a.x));
	} else if (m == 6) { // R(-1 - j)
		return DIR > 0 ? make_float2 (R * (a.y - a.x), -R * (a.x + a.y))
		               : make_float2 (-R * (a.x + a.y), R * (a.x - a.y));
	} else if (m == 10) { // -W^2
		return DIR > 0 ? make_float2 (-R * (a.x + a.y), R * (a.x - a.y))
		               : make_float2 (R * (a.y - a.x), -R * (a.y + a.x));
	} else if (m == 14) { // -W^6 = R(1 + j)
		return DIR > 0 ? make_float2 (R * (a.x - a.y), R * (a.x + a.y))
		               : make_float2 (R * (a.x + a.y), R * (a.y - a.x));
	} else {
		// generic: W16^m = (cos, -sin)
		constexpr float cw = (m == 1 || m == 15) ? C1 : (m == 3 || m == 13) ? S1
		                   : (m == 5 || m == 11) ? -S1 : /* 7, 9 */ -C1;
		constexpr float sw = (m == 1 || m == 7) ? S1 : (m == 3 || m == 5) ? C1
		                   : (m == 9 || m == 15) ? -S1 : /* 11, 13 */ -C1;
		return cmulw<DIR> (a, make_float2 (cw, -sw));
	}
}

// 16-point DFT of v[n], n = 0..15, result in natural order v[k].
// Decimation in frequency as two radix-4 stages: n = 4 al + be, k = ka + 4 la.
template <int DIR>
PROT_DEV void fft16 (float2 (&v)[16])
{
	// stage 1: for each be, DFT-4 over al; result y[be][ka] lands in v[4 ka + be]
	radix4<DIR> (v[0], v[4], v[8], v[12]);
	radix4<DIR> (v[1], v[5], v[9], v[13]);
	radix4<DIR> (v[2], v[6], v[10], v[14]);
	radix4<DIR> (v[3], v[7], v[11], v[15]);
	// twiddle y[be][ka] *= W16^{be ka}
	v[5]  = mul_w16<DIR, 1> (v[5]);
	v[6]  = mul_w16<DIR, 2> (v[6]);
	v[7]  = mul_w16<DIR, 3> (v[7]);
	v[9]  = mul_w16<DIR, 2> (v[9]);
	v[10] = mul_w16<DIR, 4> (v[10]);
	v[11] = mul_w16<DIR, 6> (v[11]);
	v[13] = mul_w16<DIR, 3> (v[13]);
	v[14] = mul_w16<DIR, 6> (v[14]);
	v[15] = mul_w16<DIR, 9> (v[15]);
	// stage 2: for each ka, DFT-4 over be; X[ka + 4 la] lands in v[4 ka + la]
	radix4<DIR> (v[0], v[1], v[2], v[3]);
	radix4<DIR> (v[4], v[5], v[6], v[7]);
	radix4<DIR> (v[8], v[9], v[10], v[11]);
	radix4<DIR> (v[12], v[13], v[14], v[15]);
	// un-permute (register renaming only): v[4 ka + la] -> X[ka + 4 la]
	float2 t;
	t = v[1];  v[1]  = v[4];  v[4]  = t;
	t = v[2];  v[2]  = v[8];  v[8]  = t;
	t = v[3];  v[3]  = v[12]; v[12] = t;
	t = v[6];  v[6]  = v[9];  v[9]  = t;
	t = v[7];  v[7]  = v[13]; v[13] = t;
	t = v[11]; v[11] = v[14]; v[14] = t;
}

} // namespace prot
