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
typedef float f4 __attribute__ ((ext_vector_type (4)));

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
// Packed butterflies: v_pk_{add,mul,fma}_{f32,f16} with op_sel / neg modifiers.
// Same arithmetic as above, half the instruction count: one instruction per
// complex add, two per complex multiply, and the +-j rotations are free
// operand swizzles.  v_pk_*_f32 has the scalar ops' lane-op rate on gfx950
// (measured, tools/ubench/valu_rate.hip) but needs half the issue slots, which
// is what a barrier/LDS-stalled wave is short of; v_pk_*_f16 also doubles the
// arithmetic rate.  hipcc's SLP vectoriser cannot produce this form: it
// marshals register pairs with ~500 v_mov per iteration instead of using the
// swizzle modifiers.
//
// One complex value is one f2 (aligned VGPR pair) or one h2 (one VGPR); the
// VOP3P modifiers mean the same for both widths: op_sel[i] / op_sel_hi[i] pick
// the low or high element of source i for the low / high result, neg_lo /
// neg_hi negate it.
// --------------------------------------------------------------------------
typedef _Float16 h2 __attribute__ ((ext_vector_type (2)));

#define PROT_PK_OPS(NAME, C, E, SFX)                                                                         \
	struct NAME {                                                                                            \
		typedef C cplx;                                                                                      \
		PROT_DEV C mk (float x, float y) { return C{ (E)x, (E)y }; }                                         \
		/* a * r (real): a plain product, which the compiler may contract into a following add */           \
		PROT_DEV C scale (C a, float r) { return a * C{ (E)r, (E)r }; }                                      \
		/* a + (-j) b = (a.x + b.y, a.y - b.x) */                                                            \
		PROT_DEV C add_mj (C a, C b)                                                                         \
		{                                                                                                    \
			C d;                                                                                             \
			asm ("v_pk_add_" SFX " %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); \
			return d;                                                                                        \
		}                                                                                                    \
		/* a - (-j) b = a + j b = (a.x - b.y, a.y + b.x) */                                                  \
		PROT_DEV C sub_mj (C a, C b)                                                                         \
		{                                                                                                    \
			C d;                                                                                             \
			asm ("v_pk_add_" SFX " %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b)); \
			return d;                                                                                        \
		}                                                                                                    \
		/* a * w (DIR > 0), a * conj(w) (DIR < 0) */                                                         \
		template <int DIR>                                                                                   \
		PROT_DEV C cmul (C a, C w)                                                                           \
		{                                                                                                    \
			C r, d;                                                                                          \
			if (DIR > 0) {                                                                                   \
				asm ("v_pk_mul_" SFX " %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(w)); \
				asm ("v_pk_fma_" SFX " %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"      \
				     : "=v"(d) : "v"(a), "v"(w), "v"(r));                                                    \
			} else {                                                                                         \
				asm ("v_pk_mul_" SFX " %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(w)); \
				asm ("v_pk_fma_" SFX " %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(w), "v"(r)); \
			}                                                                                                \
			return d;                                                                                        \
		}                                                                                                    \
		/* two independent products, interleaved: a0 *= w0, a1 *= w1 (conj(w) for DIR < 0) */               \
		template <int DIR>                                                                                   \
		PROT_DEV void cmul2 (C& a0, C w0, C& a1, C w1)                                                       \
		{                                                                                                    \
			C r0, r1, d0, d1;                                                                                \
			if (DIR > 0) {                                                                                   \
				asm ("v_pk_mul_" SFX " %0, %4, %5 op_sel:[0,0] op_sel_hi:[0,1]\n\t"                          \
				     "v_pk_mul_" SFX " %1, %6, %7 op_sel:[0,0] op_sel_hi:[0,1]\n\t"                          \
				     "v_pk_fma_" SFX " %2, %4, %5, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"   \
				     "v_pk_fma_" SFX " %3, %6, %7, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"        \
				     : "=&v"(r0), "=&v"(r1), "=&v"(d0), "=&v"(d1) : "v"(a0), "v"(w0), "v"(a1), "v"(w1));      \
			} else {                                                                                         \
				asm ("v_pk_mul_" SFX " %0, %4, %5 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"             \
				     "v_pk_mul_" SFX " %1, %6, %7 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"             \
				     "v_pk_fma_" SFX " %2, %4, %5, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"                  \
				     "v_pk_fma_" SFX " %3, %6, %7, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1]"                       \
				     : "=&v"(r0), "=&v"(r1), "=&v"(d0), "=&v"(d1) : "v"(a0), "v"(w0), "v"(a1), "v"(w1));      \
			}                                                                                                \
			a0 = d0;                                                                                         \
			a1 = d1;                                                                                         \
		}                                                                                                    \
		/* acc + a * w (DIR > 0) or acc + a * conj(w) (DIR < 0) */                                           \
		template <int DIR>                                                                                   \
		PROT_DEV C cmul_acc (C acc, C a, C w)                                                                \
		{                                                                                                    \
			C r, d;                                                                                          \
			if (DIR > 0) {                                                                                   \
				asm ("v_pk_fma_" SFX " %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(acc)); \
				asm ("v_pk_fma_" SFX " %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"      \
				     : "=v"(d) : "v"(a), "v"(w), "v"(r));                                                    \
			} else {                                                                                         \
				asm ("v_pk_fma_" SFX " %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"      \
				     : "=v"(r) : "v"(a), "v"(w), "v"(acc));                                                  \
				asm ("v_pk_fma_" SFX " %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(w), "v"(r)); \
			}                                                                                                \
			return d;                                                                                        \
		}                                                                                                    \
	}

PROT_PK_OPS (pk32, f2, float, "f32");
PROT_PK_OPS (pk16, h2, _Float16, "f16");
#undef PROT_PK_OPS

// pk32 with every product spelled out as an instruction of its own: olasave_kernel's C++ blocks must round exactly like
// its generated loop (tools/gen/gen_olasave.py), so nothing may be contracted into a neighbouring add
struct pk32a : pk32 {
	PROT_DEV f2 scale (f2 a, float r)
	{
		f2 d;
		asm ("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "s"(f2{ r, r }));
		return d;
	}
};

// TIMING EXPERIMENT (-DPROT_F16_MFMA_TIMING; results WRONG by design): an upper bound of what the fp16 sample
// path would gain from radix-16 passes on the matrix core, measured IN SITU -- every packed-half radix-16 butterfly network
// (two stages of DFT-4 and the W16 twiddles between them: ~100 v_pk_*_f16) is replaced by the instructions the matrix-core pass
// issues instead: 4 v_mfma_f32_32x32x16_f16 (the real 32 x 32 DFT matrix, two K halves x two column halves of the wave) and the
// 16 v_cvt_pkrtz that bring the fp32 products back to half.  The operands stand in for each other (no LDS layout change, no
// matrix constants loaded): what is timed is the instruction mix in the kernel's own schedule, occupancy and LDS traffic.
#ifdef PROT_F16_MFMA_TIMING
template <class O> struct pkt_mfma_timing { static constexpr bool value = sizeof (typename O::cplx) == 4; };
typedef _Float16 pkt_h8 __attribute__ ((ext_vector_type (8)));
typedef float    pkt_f16v __attribute__ ((ext_vector_type (16)));
PROT_DEV void pkt_mfma_dummy (h2 (&v)[16])
{
	pkt_h8 a0, a1;
	asm volatile ("" : "=v"(a0), "=v"(a1)); // (the DFT matrix halves: registers of the kernel's lifetime in the real thing)
	pkt_h8 b[4];
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		b[q] = pkt_h8{ v[4 * q].x, v[4 * q].y, v[4 * q + 1].x, v[4 * q + 1].y, v[4 * q + 2].x, v[4 * q + 2].y, v[4 * q + 3].x, v[4 * q + 3].y };
	}
	pkt_f16v z;
#pragma unroll
	for (int r = 0; r < 16; ++r) {
		z[r] = 0.f;
	}
	pkt_f16v r0 = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a0, b[0], z, 0, 0, 0);
	r0          = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a1, b[1], r0, 0, 0, 0);
	pkt_f16v r1 = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a0, b[2], z, 0, 0, 0);
	r1          = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a1, b[3], r1, 0, 0, 0);
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		v[i]     = __builtin_bit_cast (h2, __builtin_amdgcn_cvt_pkrtz (r0[2 * i], r0[2 * i + 1]));
		v[8 + i] = __builtin_bit_cast (h2, __builtin_amdgcn_cvt_pkrtz (r1[2 * i], r1[2 * i + 1]));
	}
}
template <class C> PROT_DEV void pkt_mfma_dummy (C (&)[16]) {}
#else
template <class O> struct pkt_mfma_timing { static constexpr bool value = false; };
template <class C> PROT_DEV void pkt_mfma_dummy (C (&)[16]) {}
#endif

// 4-point DFT on packed values; CROT: operand c arrives un-rotated and still
// has to be multiplied by -j (DIR > 0) / +j (DIR < 0) -- the W16^4 twiddle
// folded into the butterfly.
template <class O, int DIR, bool CROT = false>
PROT_DEV void pkt_radix4 (typename O::cplx& a, typename O::cplx& b, typename O::cplx& c, typename O::cplx& d)
{
	if (pkt_mfma_timing<O>::value) {
		return;
	}
	typename O::cplx t0, t1;
	if (!CROT) {
		t0 = a + c;
		t1 = a - c;
	} else if (DIR > 0) {
		t0 = O::add_mj (a, c);
		t1 = O::sub_mj (a, c);
	} else {
		t0 = O::sub_mj (a, c);
		t1 = O::add_mj (a, c);
	}
	const typename O::cplx t2 = b + d;
	const typename O::cplx t3 = b - d;
	a = t0 + t2;
	c = t0 - t2;
	if (DIR > 0) {
		b = O::add_mj (t1, t3);
		d = O::sub_mj (t1, t3);
	} else {
		b = O::sub_mj (t1, t3);
		d = O::add_mj (t1, t3);
	}
}

template <class O, int DIR, int M>
PROT_DEV typename O::cplx pkt_mul_w16 (typename O::cplx a)
{
	constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, R = 0.70710678118654752440f;
	constexpr int   m  = M & 15;
	static_assert (m == 1 || m == 2 || m == 3 || m == 6 || m == 9, "twiddle not used by fft16");
	if (m == 2) { // R(1 -+ j) a
		return O::scale (DIR > 0 ? O::add_mj (a, a) : O::sub_mj (a, a), R);
	} else if (m == 6) { // R(-1 -+ j) a
		return O::scale (DIR > 0 ? O::sub_mj (a, a) : O::add_mj (a, a), -R);
	} else {
		constexpr float cw = (m == 1) ? C1 : (m == 3) ? S1 : /* 9 */ -C1;
		constexpr float sw = (m == 1) ? S1 : (m == 3) ? C1 : /* 9 */ -S1;
		return O::template cmul<DIR> (a, O::mk (cw, -sw));
	}
}

// second half of the radix-16 (after stage 1 and its twiddles): four DFT-4 and the un-permute
template <class O, int DIR>
PROT_DEV void pkt_fft16_stage2 (typename O::cplx (&v)[16])
{
	if (pkt_mfma_timing<O>::value) {
		pkt_mfma_dummy (v);
		return;
	}
	pkt_radix4<O, DIR> (v[0], v[1], v[2], v[3]);
	pkt_radix4<O, DIR> (v[4], v[5], v[6], v[7]);
	pkt_radix4<O, DIR, true> (v[8], v[9], v[10], v[11]); // v[10] carries the pending W16^4 = -+j
	pkt_radix4<O, DIR> (v[12], v[13], v[14], v[15]);
	typename O::cplx t;
	t = v[1];  v[1]  = v[4];  v[4]  = t;
	t = v[2];  v[2]  = v[8];  v[8]  = t;
	t = v[3];  v[3]  = v[12]; v[12] = t;
	t = v[6];  v[6]  = v[9];  v[9]  = t;
	t = v[7];  v[7]  = v[13]; v[13] = t;
	t = v[11]; v[11] = v[14]; v[14] = t;
}

template <class O, int DIR>
PROT_DEV void pkt_stage1_twiddles (typename O::cplx (&v)[16])
{
	if (pkt_mfma_timing<O>::value) {
		return;
	}
	v[5]  = pkt_mul_w16<O, DIR, 1> (v[5]);
	v[6]  = pkt_mul_w16<O, DIR, 2> (v[6]);
	v[7]  = pkt_mul_w16<O, DIR, 3> (v[7]);
	v[9]  = pkt_mul_w16<O, DIR, 2> (v[9]);
	/* v[10]: W16^4 folded into stage 2 */
	v[11] = pkt_mul_w16<O, DIR, 6> (v[11]);
	v[13] = pkt_mul_w16<O, DIR, 3> (v[13]);
	v[14] = pkt_mul_w16<O, DIR, 6> (v[14]);
	v[15] = pkt_mul_w16<O, DIR, 9> (v[15]);
}

template <class O, int DIR>
PROT_DEV void pkt_fft16 (typename O::cplx (&v)[16])
{
	pkt_radix4<O, DIR> (v[0], v[4], v[8], v[12]);
	pkt_radix4<O, DIR> (v[1], v[5], v[9], v[13]);
	pkt_radix4<O, DIR> (v[2], v[6], v[10], v[14]);
	pkt_radix4<O, DIR> (v[3], v[7], v[11], v[15]);
	pkt_stage1_twiddles<O, DIR> (v);
	pkt_fft16_stage2<O, DIR> (v);
}

// forward radix-16 of a zero-padded column: only v[0..NNZ) are non-zero (NNZ = 4, 8, 10, 12, 14 or 16).
// The asm butterflies are opaque to constant folding, so the pruning is spelled out: stage 1 is four
// DFT-4 over (v[be], v[be+4], v[be+8], v[be+12]); with n of the four inputs non-zero it costs
// 0 (n = 1: copies), 4 (n = 2), 6 (n = 3) or 8 (n = 4) packed instructions.
template <class O, int NNZ>
PROT_DEV void pkt_fft16_fwd_pruned (typename O::cplx (&v)[16])
{
	static_assert (NNZ == 4 || (NNZ >= 8 && NNZ <= 16 && NNZ % 2 == 0), "zero-padded rows come in pairs");
#pragma unroll
	for (int be = 0; be < 4; ++be) {
		const int n = pkt_mfma_timing<O>::value ? 4 : (NNZ - be + 3) / 4; // non-zero inputs of this butterfly: rows be, be+4, ... < NNZ
		if (n == 1) { // y[al] = x[be]
			v[4 + be] = v[8 + be] = v[12 + be] = v[be];
		} else if (n == 2) { // DFT-4 of (a, b, 0, 0) = (a + b, a - j b, a - b, a + j b)
			const typename O::cplx a = v[be], b = v[4 + be];
			v[be]      = a + b;
			v[4 + be]  = O::add_mj (a, b);
			v[8 + be]  = a - b;
			v[12 + be] = O::sub_mj (a, b);
		} else if (n == 3) { // DFT-4 of (a, b, c, 0): s = a + c, d = a - c; (s + b, d - j b, s - b, d + j b)
			const typename O::cplx a = v[be], b = v[4 + be], c = v[8 + be];
			const typename O::cplx sm = a + c, df = a - c;
			v[be]      = sm + b;
			v[4 + be]  = O::add_mj (df, b);
			v[8 + be]  = sm - b;
			v[12 + be] = O::sub_mj (df, b);
		} else {
			pkt_radix4<O, 1> (v[be], v[4 + be], v[8 + be], v[12 + be]);
		}
	}
	pkt_stage1_twiddles<O, 1> (v);
	pkt_fft16_stage2<O, 1> (v);
}

// Radix-16 passes with the thread's outer twiddles W^{t k}, k = 0..15, applied in SPLIT form: k = ka + 4 la gives
// W^{t k} = A[ka] B[la] with A[i] = W^{t i}, B[i] = W^{4 t i} (i = 1..3): 6 twiddles in registers instead of 15.
// A multiplies the four stage-2 inputs of group ka (linearity of the DFT-4 over be), B the outputs: 24 multiplies
// instead of 15 -- 18 packed instructions more per pass, 18 VGPRs fewer for the whole kernel.
//   forward: X[k] W^{t k} out;   inverse: input x[n] conj(W^{t n}), n = 4 al + be -> conj(B[al]) in front, conj(A[be])
//   on the stage-1 outputs of group be.
template <class O, int NNZ>
PROT_DEV void pkt_fft16_fwd_pruned_twout (typename O::cplx (&v)[16], const typename O::cplx (&A)[4], const typename O::cplx (&B)[4])
{
	static_assert (NNZ == 4 || (NNZ >= 8 && NNZ <= 16 && NNZ % 2 == 0), "zero-padded rows come in pairs");
#pragma unroll
	for (int be = 0; be < 4; ++be) {
		const int n = pkt_mfma_timing<O>::value ? 4 : (NNZ - be + 3) / 4;
		if (n == 1) {
			v[4 + be] = v[8 + be] = v[12 + be] = v[be];
		} else if (n == 2) {
			const typename O::cplx a = v[be], b = v[4 + be];
			v[be]      = a + b;
			v[4 + be]  = O::add_mj (a, b);
			v[8 + be]  = a - b;
			v[12 + be] = O::sub_mj (a, b);
		} else if (n == 3) {
			const typename O::cplx a = v[be], b = v[4 + be], c = v[8 + be];
			const typename O::cplx sm = a + c, df = a - c;
			v[be]      = sm + b;
			v[4 + be]  = O::add_mj (df, b);
			v[8 + be]  = sm - b;
			v[12 + be] = O::sub_mj (df, b);
		} else {
			pkt_radix4<O, 1> (v[be], v[4 + be], v[8 + be], v[12 + be]);
		}
	}
	pkt_stage1_twiddles<O, 1> (v);
#pragma unroll
	for (int ka = 1; ka < 4; ++ka) { // y[be][ka] sits in v[4 ka + be]
		O::template cmul2<1> (v[4 * ka], A[ka], v[4 * ka + 1], A[ka]);
		O::template cmul2<1> (v[4 * ka + 2], A[ka], v[4 * ka + 3], A[ka]);
	}
	pkt_fft16_stage2<O, 1> (v); // natural order: v[ka + 4 la]
#pragma unroll
	for (int la = 1; la < 4; ++la) {
		O::template cmul2<1> (v[4 * la], B[la], v[4 * la + 1], B[la]);
		O::template cmul2<1> (v[4 * la + 2], B[la], v[4 * la + 3], B[la]);
	}
}

// The radix-16 in two halves, for callers that want the outputs of stage 2 one DFT-4 at a time: after DFT-4 number i
// its four results are X[i], X[i + 4], X[i + 8], X[i + 12] in natural order (the un-permute of pkt_fft16_stage2 becomes
// the sink's addressing).  `sink(std::integral_constant<int, i>, x0, x1, x2, x3)` multiplies / stores them while the
// next DFT-4 computes -- the LDS traffic of an exchange then runs under the wave's own butterflies instead of behind them.
template <class O, int DIR>
PROT_DEV void pkt_fft16_stage1 (typename O::cplx (&v)[16])
{
	pkt_radix4<O, DIR> (v[0], v[4], v[8], v[12]);
	pkt_radix4<O, DIR> (v[1], v[5], v[9], v[13]);
	pkt_radix4<O, DIR> (v[2], v[6], v[10], v[14]);
	pkt_radix4<O, DIR> (v[3], v[7], v[11], v[15]);
	pkt_stage1_twiddles<O, DIR> (v);
}
template <int I> struct pkt_idx { static constexpr int value = I; };
template <class O, int DIR, class S>
PROT_DEV void pkt_fft16_stage2_each (typename O::cplx (&v)[16], S&& sink)
{
	if (pkt_mfma_timing<O>::value) {
		pkt_mfma_dummy (v);
	}
	pkt_radix4<O, DIR> (v[0], v[1], v[2], v[3]);
	sink (pkt_idx<0>{}, v[0], v[1], v[2], v[3]);
	pkt_radix4<O, DIR> (v[4], v[5], v[6], v[7]);
	sink (pkt_idx<1>{}, v[4], v[5], v[6], v[7]);
	pkt_radix4<O, DIR, true> (v[8], v[9], v[10], v[11]); // v[10] carries the pending W16^4 = -+j
	sink (pkt_idx<2>{}, v[8], v[9], v[10], v[11]);
	pkt_radix4<O, DIR> (v[12], v[13], v[14], v[15]);
	sink (pkt_idx<3>{}, v[12], v[13], v[14], v[15]);
}
// first half of pkt_fft16_fwd_pruned_twout: pruned stage 1, its twiddles, A on the stage-2 inputs
template <class O, int NNZ>
PROT_DEV void pkt_fft16_fwd_pruned_twout_stage1 (typename O::cplx (&v)[16], const typename O::cplx (&A)[4])
{
	static_assert (NNZ == 4 || (NNZ >= 8 && NNZ <= 16 && NNZ % 2 == 0), "zero-padded rows come in pairs");
#pragma unroll
	for (int be = 0; be < 4; ++be) {
		const int n = pkt_mfma_timing<O>::value ? 4 : (NNZ - be + 3) / 4;
		if (n == 1) {
			v[4 + be] = v[8 + be] = v[12 + be] = v[be];
		} else if (n == 2) {
			const typename O::cplx a = v[be], b = v[4 + be];
			v[be]      = a + b;
			v[4 + be]  = O::add_mj (a, b);
			v[8 + be]  = a - b;
			v[12 + be] = O::sub_mj (a, b);
		} else if (n == 3) {
			const typename O::cplx a = v[be], b = v[4 + be], c = v[8 + be];
			const typename O::cplx sm = a + c, df = a - c;
			v[be]      = sm + b;
			v[4 + be]  = O::add_mj (df, b);
			v[8 + be]  = sm - b;
			v[12 + be] = O::sub_mj (df, b);
		} else {
			pkt_radix4<O, 1> (v[be], v[4 + be], v[8 + be], v[12 + be]);
		}
	}
	pkt_stage1_twiddles<O, 1> (v);
#pragma unroll
	for (int ka = 1; ka < 4; ++ka) {
		O::template cmul2<1> (v[4 * ka], A[ka], v[4 * ka + 1], A[ka]);
		O::template cmul2<1> (v[4 * ka + 2], A[ka], v[4 * ka + 3], A[ka]);
	}
}

template <class O>
PROT_DEV void pkt_fft16_inv_twin (typename O::cplx (&v)[16], const typename O::cplx (&A)[4], const typename O::cplx (&B)[4])
{
#pragma unroll
	for (int al = 1; al < 4; ++al) { // input n = 4 al + be
		O::template cmul2<-1> (v[4 * al], B[al], v[4 * al + 1], B[al]);
		O::template cmul2<-1> (v[4 * al + 2], B[al], v[4 * al + 3], B[al]);
	}
	pkt_radix4<O, -1> (v[0], v[4], v[8], v[12]);
	pkt_radix4<O, -1> (v[1], v[5], v[9], v[13]);
	pkt_radix4<O, -1> (v[2], v[6], v[10], v[14]);
	pkt_radix4<O, -1> (v[3], v[7], v[11], v[15]);
	pkt_stage1_twiddles<O, -1> (v);
#pragma unroll
	for (int be = 1; be < 4; ++be) { // y[be][ka] sits in v[4 ka + be]
		O::template cmul2<-1> (v[be], A[be], v[4 + be], A[be]);
		O::template cmul2<-1> (v[8 + be], A[be], v[12 + be], A[be]);
	}
	pkt_fft16_stage2<O, -1> (v);
}

// fp32 names used by the kernels
PROT_DEV f2 pk_add_mj (f2 a, f2 b) { return pk32::add_mj (a, b); }
PROT_DEV f2 pk_sub_mj (f2 a, f2 b) { return pk32::sub_mj (a, b); }
template <int DIR> PROT_DEV f2 pk_cmul (f2 a, f2 w) { return pk32::cmul<DIR> (a, w); }
template <int DIR> PROT_DEV f2 pk_cmul_acc (f2 acc, f2 a, f2 w) { return pk32::cmul_acc<DIR> (acc, a, w); }
template <int DIR> PROT_DEV void pk_fft16 (f2 (&v)[16]) { pkt_fft16<pk32, DIR> (v); }
template <int NNZ> PROT_DEV void pk_fft16_fwd_pruned (f2 (&v)[16]) { pkt_fft16_fwd_pruned<pk32, NNZ> (v); }

} // namespace prot
