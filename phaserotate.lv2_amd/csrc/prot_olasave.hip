// prot_olasave.hip -- olasave_kernel: the round-3 form of the 4096-point phase-rotate block for FIRs whose odd taps fit ONE
// partition (LV2 48 / 96 kHz, CLI blksiz 1024 ... 4096), sized for FOUR workgroups per CU (<= 128 VGPRs, 38 KiB of LDS).
//
// What it replaces: the block body of process_channel (src/phaserotate.c:629-720) and PhaseRotateProc::hilbert / rotate
// (cli/phase-rotate.cc:181-232) -- r2c, x FIR spectrum, c2r, overlap-add, cos/sin mix -- as olapoly_kernel does
// (prot_kernels.hip), with three changes that make the fourth workgroup fit:
//
//  1. OVERLAP-SAVE instead of overlap-add.  A block is 16 rows of 256 complex samples z[j] = x[2j-1] + i x[2j] (the
//     polyphase reading of the stream, see olapoly_kernel): NT = 16 - HS rows of context that the previous block already
//     had in registers, HS new rows.  The circular convolution is exact on the middle HS rows (NT/2 rows of context on
//     either side: the filter is applied CENTRED, below), and those rows' (q, c) input pairs are the dry path -- so what a
//     thread carries from block to block is its input (13 rows), not input + an output tail (19 rows), a run needs no
//     halo transform (one-block runs, BASELINE config 3, do one transform pair instead of two), and nothing is added up.
//  2. NO SPECTRUM TABLE.  The odd taps g[s] = h[2s+1] of the reference's Hilbert FIR (closed form SURVEY section 0) are
//     antisymmetric about their centre: H[k] = -i B[k] W^{(Lg/2) k} W^{-k/2} with B REAL, and |B| = 1 to 7e-8 outside the
//     256 bins next to DC (the filter is a half-sample-shift all-pass away from its transition; tools/model/
//     olasave_model.py prints the figures).  So: the forward transform is evaluated at the half-shifted time index n - 1/2
//     (its twiddles are W^{(t - 1/2) k1}, W256^{(c - 1/2) k2} and a constant W32^{-k3} on the last pass's outputs -- the
//     same instruction count as the 16 table multiplies they replace), W^{(Lg/2) k} is dropped (the filter centred: a
//     re-indexing of rows), -i is an operand swizzle, and B is one multiply on the rows k3 = 0 and 15 from a per-thread
//     register; its mean over the other rows (1 + 4e-8: the float taps' own rounding) rides in the rotation constant.
//     16 KiB of LDS, 16 ds_read_b64 and the table's registers are gone; the host checks the antisymmetry and the flatness
//     of B when the engine is created (prot_capi.hip) and keeps olapoly_kernel for any FIR that fails them.
//  3. The rotation constants arrive as kernel arguments (no copy in front of the launch).
//
// The transform itself is olapoly_kernel's: 16 x 16 x 16, three radix-16 passes in registers (prot_fft.h), one
// workgroup-wide and one wave-local LDS exchange per direction (prot_dev.h), packed-fp32 butterflies, split outer twiddles.
#include "prot_dev.h"

namespace prot {

typedef volatile __attribute__ ((address_space (3))) f2 LdsC;
typedef pk32a O; // (every product an instruction of its own: the C++ blocks round like the generated loop)

// a * w for a wave-uniform constant w (SGPR pair: the 14 constants of the last forward pass cost no VGPR)
PROT_DEV f2 cmul_sconst (f2 a, f2 w)
{
	f2 r, d;
	asm ("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "s"(w));
	asm ("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "s"(w), "v"(r));
	return d;
}
// a * (i rho.x): (-a.y rho.x, a.x rho.x)
PROT_DEV f2 mul_i_lo (f2 a, f2 rho)
{
	f2 d;
	asm ("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0]" : "=v"(d) : "v"(a), "v"(rho));
	return d;
}
// a * rho.x / a * rho.y (real)
PROT_DEV f2 mul_lo (f2 a, f2 rho)
{
	f2 d;
	asm ("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(rho));
	return d;
}
PROT_DEV f2 mul_hi (f2 a, f2 rho)
{
	f2 d;
	asm ("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(rho));
	return d;
}

// v[k] *= w[k] (k = 1..15, conj(w) for DIR < 0) with the W256 twiddles fetched from LDS four at a time, one batch ahead of
// its multiplies, every value stored to dst[k * STRIDE] as soon as it is done (the VGPR -> LDS transfers run under the
// remaining multiplies).  `r` = &table[tl]; q0 = batch 0 (k = 1..3), fetched by the caller ahead of the butterfly.
template <int DIR, int STRIDE>
PROT_DEV void sv_twiddle_store (f2 (&v)[16], const LdsC* r, const f2 (&q0)[4], LdsC* dst)
{
	f2 q[2][4];
#pragma unroll
	for (int i = 1; i < 4; ++i) {
		q[0][i] = q0[i];
	}
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		if (j < 3) {
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				q[(j + 1) & 1][i] = r[16 * (4 * (j + 1) + i)];
			}
		}
		if (j == 0) {
			dst[0] = v[0];
			v[1]   = O::template cmul<DIR> (v[1], q[0][1]);
			dst[STRIDE] = v[1];
			O::template cmul2<DIR> (v[2], q[0][2], v[3], q[0][3]);
			dst[2 * STRIDE] = v[2];
			dst[3 * STRIDE] = v[3];
		} else {
#pragma unroll
			for (int i = 0; i < 4; i += 2) {
				const int k = 4 * j + i;
				O::template cmul2<DIR> (v[k], q[j & 1][i], v[k + 1], q[j & 1][i + 1]);
				dst[k * STRIDE]       = v[k];
				dst[(k + 1) * STRIDE] = v[k + 1];
			}
		}
	}
}

// forward transform at the half-shifted time index, up to (not including) the last radix-16 pass.
// tw[0..3] = A (W^{(t-1/2) i}), tw[4..7] = B (W^{(t-1/2) 4 i}); twf = W256^{(c-1/2) k2} table in LDS
template <class G>
PROT_DEV void sv_forward (f2 (&v)[16], f2* lds, const f2* twf, const f2 (&A)[4], const f2 (&B)[4], int t, int tl, int th, G&& after_barrier)
{
	// own slots (k * kL1 + t): read by this thread at the end of the previous inverse -- no barrier in front
	pkt_fft16_fwd_pruned_twout<O, 16> (v, A, B);
	{
		LdsC* w = (LdsC*)&lds[t];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			w[k * kL1] = v[k];
		}
	}
	__syncthreads ();
	{
		const LdsC* r = (const LdsC*)&lds[th * kL1 + tl];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			v[k] = r[16 * k];
		}
	}
	after_barrier ();
	{
		const LdsC* rq = (const LdsC*)&twf[tl];
		f2          q0[4];
#pragma unroll
		for (int i = 1; i < 4; ++i) {
			q0[i] = rq[16 * i];
		}
		pkt_fft16<O, 1> (v);
		sv_twiddle_store<1, 1> (v, rq, q0, (LdsC*)&lds[th * kL1 + tl * kG]);
	}
	wave_sync ();
	{
		const LdsC* r = (const LdsC*)&lds[th * kL1 + tl];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			v[k] = r[k * kG];
		}
	}
	pkt_fft16<O, 1> (v);
}

// inverse radix-16 of the last pass with the conjugated split outer twiddles on its inputs, and only the output rows
// [R0, 16 - R0) computed (the other rows of an overlap-save block are circular wrap-around)
template <int R0>
PROT_DEV void sv_fft16_inv_twin_pruned (f2 (&v)[16], const f2 (&A)[4], const f2 (&B)[4])
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
	pkt_fft16_stage2<O, -1> (v); // (the unused outputs' adds are dead code: the compiler drops them)
}

PROT_DEV void sv_inverse (f2 (&v)[16], f2* lds, const f2* twm, const f2 (&A)[4], const f2 (&B)[4], int t, int tl, int th)
{
	{
		const LdsC* rq = (const LdsC*)&twm[tl];
		f2          q0[4];
#pragma unroll
		for (int i = 1; i < 4; ++i) {
			q0[i] = rq[16 * i];
		}
		pkt_fft16<O, -1> (v);
		sv_twiddle_store<-1, kG> (v, rq, q0, (LdsC*)&lds[th * kL1 + tl]);
	}
	wave_sync ();
	{
		const LdsC* r = (const LdsC*)&lds[th * kL1 + tl * kG];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			v[k] = r[k];
		}
	}
	pkt_fft16<O, -1> (v);
	{
		// the group's chunk again, in the pass-1 layout: nobody else touches it before the barrier
		wave_sync ();
		LdsC* w = (LdsC*)&lds[th * kL1 + tl];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			w[16 * k] = v[k];
		}
		__syncthreads ();
		const LdsC* r = (const LdsC*)&lds[t];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			v[k] = r[k * kL1];
		}
	}
	// the plain twiddles from the half-shifted ones: W^{t i} = W^{(t-1/2) i} W^{i/2}, W^{4 t i} = W^{(t-1/2) 4 i} W^{2 i}
	f2 Ai[4], Bi[4];
	Ai[1] = cmul_sconst (A[1], mkf2 (0.99999970586288223f, -0.00076699031874270453f));
	Ai[2] = cmul_sconst (A[2], mkf2 (0.99999882345170188f, -0.0015339801862847656f));
	Ai[3] = cmul_sconst (A[3], mkf2 (0.99999735276697821f, -0.0023009691514258073f));
	Bi[1] = cmul_sconst (B[1], mkf2 (0.99999529380957619f, -0.0030679567629659763f));
	Bi[2] = cmul_sconst (B[2], mkf2 (0.99998117528260111f, -0.0061358846491544753f));
	Bi[3] = cmul_sconst (B[3], mkf2 (0.9999576445519639f, -0.0092037547820598194f));
	pkt_fft16_inv_twin<O> (v, Ai, Bi);
}

// the 14 constants i W32^{-k3} = e^{i (pi/2 + pi k3 / 16)}, k3 = 1..15 (k3 = 8: -1)
PROT_DEV f2 sv_c32 (int k)
{
	constexpr float C[16] = { 0.f, -0.19509032201612825f, -0.38268343236508977f, -0.55557023301960222f, -0.70710678118654752f,
		                      -0.83146961230254524f, -0.92387953251128676f, -0.98078528040323045f, -1.f, -0.98078528040323045f,
		                      -0.92387953251128676f, -0.83146961230254524f, -0.70710678118654752f, -0.55557023301960222f,
		                      -0.38268343236508977f, -0.19509032201612825f };
	constexpr float S[16] = { 1.f, 0.98078528040323045f, 0.92387953251128676f, 0.83146961230254524f, 0.70710678118654752f,
		                      0.55557023301960222f, 0.38268343236508977f, 0.19509032201612825f, 0.f, -0.19509032201612825f,
		                      -0.38268343236508977f, -0.55557023301960222f, -0.70710678118654752f, -0.83146961230254524f,
		                      -0.92387953251128676f, -0.98078528040323045f };
	return mkf2 (C[k], S[k]);
}

// out = a * x + y (packed, a wave-uniform): the mix of the dry and the Hilbert branch, spelled out so that the C++ blocks
// and the generated loop round alike
PROT_DEV f2 pk_mul_s (f2 x, f2 a)
{
	f2 d;
	asm ("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(x), "s"(a));
	return d;
}
PROT_DEV f2 pk_fma_s (f2 x, f2 a, f2 y)
{
	f2 d;
	asm ("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "s"(a), "v"(y));
	return d;
}

#include "prot_olasave_asm.inc"
#define PROT_SV_CAT_(a, b, c) PROT_SV_ASM_##a##_##b##_##c
#define PROT_SV_CLOB_(a) PROT_SV_ASM_CLOBBERS_##a
#define PROT_SV_RUN(HS_, MODE_, NA_)                                                                                                  \
	if (HS == HS_ && MODE == MODE_ && NA == NA_) {                                                                                    \
		asm volatile (PROT_SV_CAT_ (HS_, MODE_, NA_)                                                                                  \
		              :                                                                                                               \
		              : "s"(a_in), "s"(a_out), "s"(nf), "s"(lds_off), "s"(a_sv), "s"(ca), "s"(sa), "v"(t0)                               \
		              : PROT_SV_CLOB_ (HS_));                                                                                        \
	}

// HS new rows (512 HS frames) per block; MODE_ROT / MODE_RAMP / MODE_HILBERT; NA = rows next to DC (on either side) whose
// |B| needs its own factor (1: FIRs of 1024 odd taps and more; 2: the 512 odd taps of CLI blksiz 1024).
// A workgroup walks its run block by block: consecutive blocks that lie wholly inside the input array (aligned streams,
// steady rotation) go to the generated loop (prot_olasave_asm.inc: 126 hand-allocated VGPRs, the next block's rows
// prefetched, the context rows carried in registers); every other block -- history or silence in reach, a partial last
// block, odd alignment, per-frame rotation -- is done here, self-contained, with the same instructions in the same order.
template <int HS, int MODE, int NA>
__global__ __launch_bounds__ (256, 4) void olasave_kernel (const OlaParams p)
{
	constexpr int BLK = 512 * HS; // output frames per block
	constexpr int NT  = 16 - HS;  // context rows
	constexpr int R0  = NT / 2;   // first valid output row = first dry row
	static_assert ((HS == 8 || HS == 10 || HS == 12) && NA == 1, "rows per block (HS = 14 needs 124 VGPRs before any temporary: olapoly_kernel)");

	__shared__ f2 smem[kLdsExchange + 512];
	f2* const lds = smem;
	f2* const twm = smem + kLdsExchange;
	f2* const twf = smem + kLdsExchange + 256;

	const int t0 = threadIdx.x;
	const int c  = blockIdx.y;
	int64_t   r0, rlen;
	run_of_workgroup (p, r0, rlen);
	if (r0 >= p.n_out) {
		hist_write_slice (p, blockIdx.y, threadIdx.x);
		return;
	}
	const int64_t r1 = (r0 + rlen < p.n_out) ? r0 + rlen : p.n_out;

	const float* __restrict__ in   = p.in + (int64_t)c * p.in_stride;
	const float* __restrict__ hist = p.hist ? p.hist + (int64_t)c * p.hist_len : nullptr;
	float* __restrict__ out        = p.out + (int64_t)c * p.out_stride;
	const int64_t n_in = p.n_in;
	const int     Dd   = p.dry_delay; // even (host guarantees)
	const bool in_al8  = (reinterpret_cast<uintptr_t> (in) & 7) == 0;
	const bool out_al8 = (reinterpret_cast<uintptr_t> (out) & 7) == 0;

	// W256 tables in LDS: plain (inverse) and at the half-shifted index (forward)
	{
		const f2* __restrict__ sv  = reinterpret_cast<const f2*> (p.sv);
		const f2* __restrict__ twp = reinterpret_cast<const f2*> (p.tw);
		twf[t0] = sv[12 * 256 + t0];
		twm[t0] = twp[16 * (((t0 & 15) * (t0 >> 4)) & 255)];
	}
	float ca = 0.f, sa = p.sv_gain;
	if (MODE == MODE_ROT) {
		ca = p.rot ? p.rot[2 * c] : p.rot_v[2 * c];
		sa = (p.rot ? p.rot[2 * c + 1] : p.rot_v[2 * c + 1]) * p.sv_gain;
	}
	ca = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (ca)));
	sa = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (sa)));
	const unsigned lds_off = (unsigned)(uintptr_t)(__attribute__ ((address_space (3))) f2*)smem;
	__syncthreads ();

	for (int64_t s = r0; s < r1;) {
		const int64_t a0 = s - Dd - 512 * R0; // frame of (row 0, thread 0).q: the dry frame of output row r is input row r
		if (MODE != MODE_RAMP && in_al8 && out_al8 && a0 >= 1 && !(p.sv_flags & 1)) {
			// blocks from here on that read only frames inside the array and are emitted whole
			const int64_t n1 = (r1 - s) / BLK;
			const int64_t n2 = n_in - 8192 - a0 >= 0 ? (n_in - 8192 - a0) / BLK + 1 : 0;
			int           nf = (int)(n1 < n2 ? n1 : n2);
			if (nf > 0) {
				nf                    = __builtin_amdgcn_readfirstlane (nf);
				const float* a_in  = uniform_ptr_generic (in + a0);
				float*       a_out = uniform_ptr_generic (out + s);
				const float* a_sv  = uniform_ptr_generic (p.sv);
				PROT_SV_RUN (8, 0, 1)
				PROT_SV_RUN (8, 2, 1)
				PROT_SV_RUN (10, 0, 1)
				PROT_SV_RUN (10, 2, 1)
				PROT_SV_RUN (12, 0, 1)
				PROT_SV_RUN (12, 2, 1)
				s += (int64_t)nf * BLK;
				continue;
			}
		}
		// ---- one block, self-contained
		int t = t0;
		asm volatile ("" : "+v"(t)); // (opaque per block: per-lane address terms are otherwise hoisted out of the loop)
		t &= 255;
		const int tl = t & 15, th = t >> 4;
		f2    inr[16]; // (q, c) pairs (x[a], x[a+1]), a = a0 + 512 row + 2 t
		float g = 0.f; // lane r of every wave: the frame in front of row r's first pair of this wave
		{
			const bool     fast = (a0 - 1 >= 0) && (a0 + 8192 <= n_in);
			const unsigned l    = (unsigned)t & 63u;
			const unsigned grow = l < 16u ? l : 0u;
			if (fast && in_al8) {
#pragma unroll
				for (int m = 0; m < 8; ++m) {
					const PROT_GLOBAL float* __restrict__ b = uniform_ptr (in + a0 + 1024 * m);
#pragma unroll
					for (int e = 0; e < 2; ++e) {
						inr[2 * m + e] = reinterpret_cast<const PROT_GLOBAL f2*> (b)[(unsigned)t + 256u * e];
					}
				}
				const PROT_GLOBAL float* __restrict__ b = uniform_ptr (in + a0 - 1);
				g = b[((unsigned)t & ~63u) * 2u + 512u * grow];
			} else {
#pragma unroll
				for (int k = 0; k < 16; ++k) {
					const int64_t a = a0 + 2 * (int64_t)(t + 256 * k);
					inr[k]          = mkf2 (ldv (in, n_in, hist, p.hist_len, a), ldv (in, n_in, hist, p.hist_len, a + 1));
				}
				g = ldv (in, n_in, hist, p.hist_len, a0 + 2 * (int64_t)(((unsigned)t & ~63u) + 256u * grow) - 1);
			}
		}
		f2 twr[8]; // [1..3] A = W^{(t-1/2) i}, [5..7] B = W^{(t-1/2) 4 i}
		f2 rho[NA];
		{
			const f2* __restrict__ sv = reinterpret_cast<const f2*> (p.sv);
#pragma unroll
			for (int i = 0; i < 6; ++i) {
				twr[(i / 3) * 4 + 1 + i % 3] = sv[i * 256 + t];
			}
#pragma unroll
			for (int i = 0; i < NA; ++i) {
				rho[i] = sv[(13 + i) * 256 + t]; // (rho[q + 256 i], rho[q + 256 (15 - i)])
			}
		}
		f2 v[16];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			// wave_shr:1 -- lane i reads lane i-1, lane 0 keeps `old` = the gathered frame of this row
			const unsigned first = (unsigned)__builtin_amdgcn_readlane ((int)__float_as_uint (g), k);
			const float    pk_   = __uint_as_float (__builtin_amdgcn_update_dpp (first, __float_as_uint (inr[k].y), 0x138, 0xf, 0xf, false));
			v[k] = mkf2 (pk_, inr[k].x);
		}
		sv_forward (v, lds, twf, reinterpret_cast<const f2 (&)[4]> (twr[0]), reinterpret_cast<const f2 (&)[4]> (twr[4]), t, tl, th, [] () {});

		// Y[k] = Z'[k] (-i B[k]) W^{-(Lg/2) k}: the constant i W32^{-k3} per row, the |B| factors next to DC
		v[0] = mul_i_lo (v[0], rho[0]);
#pragma unroll
		for (int k = 1; k < 16; ++k) {
			v[k] = k == 8 ? pk_mul_s (v[k], mkf2 (-1.f, -1.f)) : cmul_sconst (v[k], sv_c32 (k));
		}
		v[15] = mul_hi (v[15], rho[0]);
		if (NA == 2) {
			v[1]  = mul_lo (v[1], rho[NA - 1]);
			v[14] = mul_hi (v[14], rho[NA - 1]);
		}
		sv_inverse (v, lds, twm, reinterpret_cast<const f2 (&)[4]> (twr[0]), reinterpret_cast<const f2 (&)[4]> (twr[4]), t, tl, th);

		const f2 ca2 = mkf2 (ca, ca), sa2 = mkf2 (sa, sa);
		if (MODE != MODE_RAMP && s + BLK <= r1 && out_al8) {
#pragma unroll
			for (int m = 0; m < HS / 2; ++m) {
				PROT_GLOBAL f2* __restrict__ ob = reinterpret_cast<PROT_GLOBAL f2*> (uniform_ptr (out + s + 1024 * m));
#pragma unroll
				for (int e = 0; e < 2; ++e) {
					const int k = R0 + 2 * m + e;
					ob[(unsigned)t + 256u * e] = MODE == MODE_ROT ? pk_fma_s (v[k], sa2, pk_mul_s (inr[k], ca2)) : pk_mul_s (v[k], sa2);
				}
			}
		} else {
#pragma unroll
			for (int k = 0; k < HS; ++k) {
				f2 o = MODE == MODE_ROT ? pk_fma_s (v[R0 + k], sa2, pk_mul_s (inr[R0 + k], ca2)) : pk_mul_s (v[R0 + k], sa2);
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					const int64_t pos = s + 2 * (int64_t)(t + 256 * k) + h;
					if (pos < r1) {
						if (MODE == MODE_RAMP) {
							const f2 cs = reinterpret_cast<const f2*> (p.rot_s)[(int64_t)c * p.n_out + pos];
							out[pos]    = cs.x * (h ? inr[R0 + k].y : inr[R0 + k].x) + (cs.y * p.sv_gain) * (h ? v[R0 + k].y : v[R0 + k].x);
						} else {
							out[pos] = h ? o.y : o.x;
						}
					}
				}
			}
		}
		s += BLK;
	}
	hist_write_slice (p, blockIdx.y, threadIdx.x);
}

template <int HS, int NA>
static hipError_t launch_sv_m (const OlaParams& p, dim3 grid, hipStream_t s)
{
	const dim3 block (256);
	if (p.hilbert_only) {
		hipLaunchKernelGGL ((olasave_kernel<HS, MODE_HILBERT, NA>), grid, block, 0, s, p);
	} else if (p.rot_s != nullptr) {
		hipLaunchKernelGGL ((olasave_kernel<HS, MODE_RAMP, NA>), grid, block, 0, s, p);
	} else {
		hipLaunchKernelGGL ((olasave_kernel<HS, MODE_ROT, NA>), grid, block, 0, s, p);
	}
	return hipGetLastError ();
}

hipError_t launch_ola_save (const OlaParams& p, int hop_rows, int n_rho_rows, int n_chn, hipStream_t s)
{
	if ((p.dry_delay & 1) || (p.run_frames % (512 * hop_rows)) != 0 || p.accumulate || p.sv == nullptr ||
	    (n_rho_rows != 1 && n_rho_rows != 2) || 2 * (p.dry_delay - p.conv_delay) != 512 * (16 - hop_rows)) {
		return hipErrorInvalidValue;
	}
	if (p.n_out <= 0) {
		return hipSuccess;
	}
	const int64_t runs = (p.n_out + p.run_frames - 1) / p.run_frames;
	dim3          grid ((unsigned)runs, (unsigned)n_chn);
	switch (hop_rows * 4 + n_rho_rows) {
		case 8 * 4 + 1: return launch_sv_m<8, 1> (p, grid, s);
		case 10 * 4 + 1: return launch_sv_m<10, 1> (p, grid, s);
		case 12 * 4 + 1: return launch_sv_m<12, 1> (p, grid, s);
		default: return hipErrorInvalidValue;
	}
}

} // namespace prot
