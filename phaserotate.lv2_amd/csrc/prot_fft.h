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

// complex value = one aligned VGPR pair; .x = re, .y = im
typedef float f2 __attribute__ ((ext_vector_type (2)));
PROT_DEV f2 mkf2 (float x, float y) { return f2{ x, y }; }

// a * w for DIR > 0, a * conj(w) for DIR < 0   (w = forward twiddle e^{-i phi})
template <int DIR>
PROT_DEV f2 cmulw (f2 a, f2 w)
{
	if (DIR > 0) {
		return mkf2 (a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
	} else {
		return mkf2 (a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y);
	}
}

// 4-point DFT, in place.  DIR > 0: e^{-2 pi i nk/4}; DIR < 0: conjugate kernel.
template <int DIR>
PROT_DEV void radix4 (f2& a, f2& b, f2& c, f2& d)
{
	const f2 t0 = mkf2 (a.x + c.x, a.y + c.y);
	const f2 t1 = mkf2 (a.x - c.x, a.y - c.y);
	const f2 t2 = mkf2 (b.x + d.x, b.y + d.y);
	const f2 t3 = mkf2 (b.x - d.x, b.y - d.y);
	a               = mkf2 (t0.x + t2.x, t0.y + t2.y);
	c               = mkf2 (t0.x - t2.x, t0.y - t2.y);
	if (DIR > 0) { // X1 = t1 - j t3, X3 = t1 + j t3
		b = mkf2 (t1.x + t3.y, t1.y - t3.x);
		d = mkf2 (t1.x - t3.y, t1.y + t3.x);
	} else {
		b = mkf2 (t1.x - t3.y, t1.y + t3.x);
		d = mkf2 (t1.x + t3.y, t1.y - t3.x);
	}
}

// multiply by W16^m (forward) or its conjugate, m a compile-time constant
template <int DIR, int M>
PROT_DEV f2 mul_w16 (f2 a)
{
	constexpr float C1 = 0.92387953251128675613f; // cos(pi/8)
	constexpr float S1 = 0.38268343236508977173f; // sin(pi/8)
	constexpr float R  = 0.70710678118654752440f; // sqrt(1/2)
	constexpr int   m  = M & 15;
	if (m == 0) {
		return a;
	} else if (m == 4) { // -j (fwd) / +j (inv)
		return DIR > 0 ? mkf2 (a.y, -a.x) : mkf2 (-a.y, a.x);
	} else if (m == 8) {
		return mkf2 (-a.x, -a.y);
	} else if (m == 12) {
		return DIR > 0 ? mkf2 (-a.y, a.x) : mkf2 (a.y, -a.x);
	} else if (m == 2) { // R(1 - j)
		return DIR > 0 ? mkf2 (R * (a.x + a.y), R * (a.y - a.x))
		               : mkf2 (R * (a.x - a.y), R * (a.y + a.x));
	} else if (m == 6) { // R(-1 - j)
		return DIR > 0 ? mkf2 (R * (a.y - a.x), -R * (a.x + a.y))
		               : mkf2 (-R * (a.x + a.y), R * (a.x - a.y));
	} else if (m == 10) { // -W^2
		return DIR > 0 ? mkf2 (-R * (a.x + a.y), R * (a.x - a.y))
		               : mkf2 (R * (a.y - a.x), -R * (a.y + a.x));
	} else if (m == 14) { // -W^6 = R(1 + j)
		return DIR > 0 ? mkf2 (R * (a.x - a.y), R * (a.x + a.y))
		               : mkf2 (R * (a.x + a.y), R * (a.y - a.x));
	} else {
		// generic: W16^m = (cos, -sin)
		constexpr float cw = (m == 1 || m == 15) ? C1 : (m == 3 || m == 13) ? S1
		                   : (m == 5 || m == 11) ? -S1 : /* 7, 9 */ -C1;
		constexpr float sw = (m == 1 || m == 7) ? S1 : (m == 3 || m == 5) ? C1
		                   : (m == 9 || m == 15) ? -S1 : /* 11, 13 */ -C1;
		return cmulw<DIR> (a, mkf2 (cw, -sw));
	}
}

// 16-point DFT of v[n], n = 0..15, result in natural order v[k].
// Decimation in frequency as two radix-4 stages: n = 4 al + be, k = ka + 4 la.
template <int DIR>
PROT_DEV void fft16 (f2 (&v)[16])
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
	f2 t;
	t = v[1];  v[1]  = v[4];  v[4]  = t;
	t = v[2];  v[2]  = v[8];  v[8]  = t;
	t = v[3];  v[3]  = v[12]; v[12] = t;
	t = v[6];  v[6]  = v[9];  v[9]  = t;
	t = v[7];  v[7]  = v[13]; v[13] = t;
	t = v[11]; v[11] = v[14]; v[14] = t;
}


// --------------------------------------------------------------------------
// Packed-fp32 butterflies (v_pk_add/mul/fma_f32 with op_sel / neg modifiers).
// Same arithmetic as above, half the instruction count: one instruction per
// complex add, two per complex multiply, and the +-j rotations are free
// operand swizzles.  v_pk_*_f32 has the scalar ops' lane-op rate on gfx950
// (measured, tools/ubench/valu_rate.hip) but needs half the issue slots, which
// is what a barrier/LDS-stalled wave is short of.  hipcc's SLP vectoriser
// cannot produce this form: it marshals register pairs with ~500 v_mov per
// iteration instead of using the swizzle modifiers.
// --------------------------------------------------------------------------
// a + (-j) b = (a.x + b.y, a.y - b.x)
PROT_DEV f2 pk_add_mj (f2 a, f2 b)
{
	f2 d;
	asm ("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
	return d;
}
// a - (-j) b = a + j b = (a.x - b.y, a.y + b.x)
PROT_DEV f2 pk_sub_mj (f2 a, f2 b)
{
	f2 d;
	asm ("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
	return d;
}
// a * w (DIR > 0), a * conj(w) (DIR < 0)
template <int DIR>
PROT_DEV f2 pk_cmul (f2 a, f2 w)
{
	f2 r, d;
	if (DIR > 0) {
		asm ("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(w));               // (ax wx, ax wy)
		asm ("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(w), "v"(r)); // (-ay wy + ., ay wx + .)
	} else {
		asm ("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(w));  // (ax wx, -ax wy)
		asm ("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(w), "v"(r)); // (ay wy + ., ay wx + .)
	}
	return d;
}

// acc + a * w (DIR > 0) or acc + a * conj(w) (DIR < 0)
template <int DIR>
PROT_DEV f2 pk_cmul_acc (f2 acc, f2 a, f2 w)
{
	f2 r, d;
	if (DIR > 0) {
		asm ("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(acc));
		asm ("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(w), "v"(r));
	} else {
		asm ("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(acc));
		asm ("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(w), "v"(r));
	}
	return d;
}

// 4-point DFT on packed values; CROT: operand c arrives un-rotated and still
// has to be multiplied by -j (DIR > 0) / +j (DIR < 0) -- the W16^4 twiddle
// folded into the butterfly.
template <int DIR, bool CROT = false>
PROT_DEV void pk_radix4 (f2& a, f2& b, f2& c, f2& d)
{
	f2 t0, t1;
	if (!CROT) {
		t0 = a + c;
		t1 = a - c;
	} else if (DIR > 0) {
		t0 = pk_add_mj (a, c);
		t1 = pk_sub_mj (a, c);
	} else {
		t0 = pk_sub_mj (a, c);
		t1 = pk_add_mj (a, c);
	}
	const f2 t2 = b + d;
	const f2 t3 = b - d;
	a           = t0 + t2;
	c           = t0 - t2;
	if (DIR > 0) {
		b = pk_add_mj (t1, t3);
		d = pk_sub_mj (t1, t3);
	} else {
		b = pk_sub_mj (t1, t3);
		d = pk_add_mj (t1, t3);
	}
}

template <int DIR, int M>
PROT_DEV f2 pk_mul_w16 (f2 a)
{
	constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, R = 0.70710678118654752440f;
	constexpr int   m  = M & 15;
	static_assert (m == 1 || m == 2 || m == 3 || m == 6 || m == 9, "twiddle not used by fft16");
	if (m == 2) { // R(1 -+ j) a
		return (DIR > 0 ? pk_add_mj (a, a) : pk_sub_mj (a, a)) * f2{ R, R };
	} else if (m == 6) { // R(-1 -+ j) a
		return (DIR > 0 ? pk_sub_mj (a, a) : pk_add_mj (a, a)) * f2{ -R, -R };
	} else {
		constexpr float cw = (m == 1) ? C1 : (m == 3) ? S1 : /* 9 */ -C1;
		constexpr float sw = (m == 1) ? S1 : (m == 3) ? C1 : /* 9 */ -S1;
		return pk_cmul<DIR> (a, f2{ cw, -sw });
	}
}

// second half of the radix-16 (after stage 1 and its twiddles): four DFT-4 and the un-permute
template <int DIR>
PROT_DEV void pk_fft16_stage2 (f2 (&v)[16])
{
	pk_radix4<DIR> (v[0], v[1], v[2], v[3]);
	pk_radix4<DIR> (v[4], v[5], v[6], v[7]);
	pk_radix4<DIR, true> (v[8], v[9], v[10], v[11]); // v[10] carries the pending W16^4 = -+j
	pk_radix4<DIR> (v[12], v[13], v[14], v[15]);
	f2 t;
	t = v[1];  v[1]  = v[4];  v[4]  = t;
	t = v[2];  v[2]  = v[8];  v[8]  = t;
	t = v[3];  v[3]  = v[12]; v[12] = t;
	t = v[6];  v[6]  = v[9];  v[9]  = t;
	t = v[7];  v[7]  = v[13]; v[13] = t;
	t = v[11]; v[11] = v[14]; v[14] = t;
}

template <int DIR>
PROT_DEV void pk_stage1_twiddles (f2 (&v)[16])
{
	v[5]  = pk_mul_w16<DIR, 1> (v[5]);
	v[6]  = pk_mul_w16<DIR, 2> (v[6]);
	v[7]  = pk_mul_w16<DIR, 3> (v[7]);
	v[9]  = pk_mul_w16<DIR, 2> (v[9]);
	/* v[10]: W16^4 folded into stage 2 */
	v[11] = pk_mul_w16<DIR, 6> (v[11]);
	v[13] = pk_mul_w16<DIR, 3> (v[13]);
	v[14] = pk_mul_w16<DIR, 6> (v[14]);
	v[15] = pk_mul_w16<DIR, 9> (v[15]);
}

template <int DIR>
PROT_DEV void pk_fft16 (f2 (&v)[16])
{
	pk_radix4<DIR> (v[0], v[4], v[8], v[12]);
	pk_radix4<DIR> (v[1], v[5], v[9], v[13]);
	pk_radix4<DIR> (v[2], v[6], v[10], v[14]);
	pk_radix4<DIR> (v[3], v[7], v[11], v[15]);
	pk_stage1_twiddles<DIR> (v);
	pk_fft16_stage2<DIR> (v);
}

// forward radix-16 of a zero-padded column: only v[0..NNZ) are non-zero (NNZ = 4 or 8).
// The asm butterflies are opaque to constant folding, so the pruning is spelled out.
template <int NNZ>
PROT_DEV void pk_fft16_fwd_pruned (f2 (&v)[16])
{
	static_assert (NNZ == 4 || NNZ == 8, "hop is 1/4 or 1/2 of the transform");
	if (NNZ == 4) { // stage 1 degenerates to copies: y[be][ka] = x[be]
#pragma unroll
		for (int be = 0; be < 4; ++be) {
			v[4 + be] = v[8 + be] = v[12 + be] = v[be];
		}
	} else { // inputs al = 0, 1 only: DFT-4 of (a, b, 0, 0) = (a + b, a - j b, a - b, a + j b)
#pragma unroll
		for (int be = 0; be < 4; ++be) {
			const f2 a = v[be], b = v[4 + be];
			v[be]      = a + b;
			v[4 + be]  = pk_add_mj (a, b);
			v[8 + be]  = a - b;
			v[12 + be] = pk_sub_mj (a, b);
		}
	}
	pk_stage1_twiddles<1> (v);
	pk_fft16_stage2<1> (v);
}

} // namespace prot
