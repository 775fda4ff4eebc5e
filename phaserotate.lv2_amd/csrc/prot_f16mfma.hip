// prot_f16mfma.hip -- the fp16 sample path (BASELINE configs[4]) with its radix-16 passes on the MATRIX CORE (round 6).
//
// The block is olapoly_kernel<T16, 1, HS, MODE_ROT>'s (prot_kernels.hip): the polyphase overlap-add of one partition of the odd
// taps g on a 4096-point transform pair, HS rows of 256 complex samples (= 512 HS stream frames) new per block, half-precision
// samples in HBM (2 B in + 2 B out per frame), overlap-add tail and rotation in fp32.  What is different is WHERE the six
// radix-16 passes of a transform pair are computed: a 16-point DFT of complex values is a real 32 x 32 matrix applied to 32 reals
// per column, and 64 columns -- one wave's share of a pass -- are four v_mfma_f32_32x32x16_f16 (two K halves x two groups of 32
// columns), fp32 accumulate, against ~100 v_pk_*_f16 of the butterfly network (tools/ubench/mfma_dft16.hip: 125 against 180 ns
// per wave-pass; in situ with the operands standing in for each other -21 % of the launch: profiles/r06_experiments.txt).
//
// The matrix core dictates who holds what (operand layout of the instruction, verified by tools/ubench/mfma_layout.hip):
//   lane l = 32 h + n of a wave supplies 8 consecutive K values of column n  ->  a lane holds 8 of the 16 elements of TWO columns
//   (groups g = 0, 1 of 32 columns) instead of 16 elements of one, and gets back 8 of the 16 outputs of each.
// Which 8 is free -- the matrix's rows and columns may be permuted -- and every LDS exchange is only a matter of addresses:
// tools/model/f16mfma_model.py is the lane-level model of exactly this file (every slot, address and twiddle index; checked
// against numpy's FFT for HS = 8, 10, 12, 14), the names below are its names.
//   pass A   columns t = j0 + 16 j1 (= 64 w + 32 g + n: consecutive lanes = consecutive samples, coalesced), elements = rows k
//   pass B   columns (j0, a), elements j1      exchange 1: slot a * 272 + t               (workgroup barrier)
//   pass C   columns (a, b),  elements j0      exchange 2: slot a * 272 + b * 17 + j0     (wave-local: rows 4 w .. 4 w + 3 are the wave's)
//   bins kappa = a + 16 b + 256 c in the lane that holds column (a, b), outputs c; the spectrum multiply; the inverse mirrors it
//   with NO exchange between forward C and inverse C (the input slots of inverse C are the output slots of forward C: a column
//   permutation of its matrix).  Five matrices (forward A with the row set of the block's non-zero input rows; forward B / C;
//   inverse C; inverse B; inverse A with the output rows paired for the overlap-add: row k < 16 - HS sits in the lane of row
//   k + HS), 2 KiB each in LDS, fetched as two ds_read_b128 per pass.  Outer twiddles W4096^{t a} (16 per lane) and W256^{j0 b}
//   (8 per lane) live in registers as half pairs; both serve the inverse conjugated (same index sets by construction).
// The stream: z[j] = (x[2j - 1], x[2j]) is one frame off the aligned pairs (x[2j], x[2j + 1]) -- two aligned dword loads per
// sample (the second one is the first one of the next sample: L1 hits) and one v_alignbit instead of a lane shift, which in this
// layout would cross the half-wave.  Dry frames are re-read as aligned pairs (L2: the workgroup fetched them a block ago).
// Steady rotation only (MODE_ROT), one partition: per-frame ramps, the Hilbert branch alone and FIRs of more than 4096 odd taps
// keep olapoly_kernel<T16>.  Bounds-checked per-frame paths for blocks that touch the ends of the stream or the history.
#include "prot_dev.h"

namespace prot {

typedef _Float16 mf_h8 __attribute__ ((ext_vector_type (8)));
typedef float    mf_f16 __attribute__ ((ext_vector_type (16)));
typedef unsigned mf_u4 __attribute__ ((ext_vector_type (4)));
typedef unsigned mf_u2 __attribute__ ((ext_vector_type (2)));
typedef mf_u2    mf_u2a __attribute__ ((aligned (4))); // an 8-byte access that is only dword-aligned (conv_delay = 2 mod 4 frames)
typedef volatile __attribute__ ((address_space (3))) h2 mf_lds;
typedef volatile __attribute__ ((address_space (3))) mf_u2 mf_lds2; // both columns of a lane in one 8-byte access (exchange 1 and its mirror)

// Rows by lane half (tools/model/f16mfma_model.py, rows_for / cxx_tables -- printed by `f16mfma_model.py --tables`):
//   input rows of the first forward pass: E_h = [HS/2 h, HS/2 (h + 1)) in slot order, zero rows behind them;
//   O[h][u]: output row of slot u of the last inverse pass = the lane's input rows DH later (mod HS) -- the dry frame of output row k
//   is the input pair of row k - DH, of this block or the one before: the lane already holds it -- rows with an overlap-add tail
//   first, their tails (row + HS) in slots HS/2 .. 7 in the same order;
//   DSRC[h][u]: where the dry pair of output slot u comes from: >= 0 the current block's input slot, < 0 the carried pair -1 - x of
//   the block before (the carried pairs are the block's last DH input rows: slots HS/2 - DH .. HS/2 - 1 of lane half 1).
template <int HS> struct MfRows;
template <> struct MfRows<8> {
	static constexpr int DH = 4;
	static constexpr int O[2][8] = { { 4, 5, 6, 7, 12, 13, 14, 15 }, { 0, 1, 2, 3, 8, 9, 10, 11 } };
	static constexpr int DSRC[2][4] = { { 0, 1, 2, 3 }, { -1, -2, -3, -4 } };
};
template <> struct MfRows<10> {
	static constexpr int DH = 3;
	static constexpr int O[2][8] = { { 3, 4, 5, 6, 7, 13, 14, 15 }, { 0, 1, 2, 8, 9, 10, 11, 12 } };
	static constexpr int DSRC[2][5] = { { 0, 1, 2, 3, 4 }, { -1, -2, -3, 0, 1 } };
};
template <> struct MfRows<12> {
	static constexpr int DH = 2;
	static constexpr int O[2][8] = { { 2, 3, 4, 5, 6, 7, 14, 15 }, { 0, 1, 8, 9, 10, 11, 12, 13 } };
	static constexpr int DSRC[2][6] = { { 0, 1, 2, 3, 4, 5 }, { -1, -2, 0, 1, 2, 3 } };
};
template <> struct MfRows<14> {
	static constexpr int DH = 1;
	static constexpr int O[2][8] = { { 1, 2, 3, 4, 5, 6, 7, 15 }, { 0, 8, 9, 10, 11, 12, 13, 14 } };
	static constexpr int DSRC[2][7] = { { 0, 1, 2, 3, 4, 5, 6 }, { -1, 0, 1, 2, 3, 4, 5 } };
};
template <int HS>
PROT_DEV constexpr int mf_in_row (int h, int s)
{
	constexpr int HH = HS / 2, NZ = 8 - HH; // zero rows per half (their order matters to the matrix only: the model's)
	return s < HH ? s + HH * h : (h == 0 ? HS + (s - HH) : 16 - NZ + (s - HH));
}
PROT_DEV constexpr int mf_onat (int h, int u) { return 4 * (u >> 1) + 2 * h + (u & 1); }
PROT_DEV constexpr int mf_enat (int h, int s) { return 8 * (s >> 2) + 4 * h + (s & 3); }

constexpr int kMfMat   = 5;       // matrices
constexpr int kMfMatH  = 32 * 32; // halfs per matrix
constexpr int kMfHsLen = 2049;
#ifndef MF_WPS
#define MF_WPS 4
#endif
constexpr int kMfWgPerCu = MF_WPS;    // 124-126 VGPRs, 37 KiB of LDS: four workgroups per CU (the host sizes its one-round launches for
                                      // that; three -- 150-168 VGPRs -- measure the same: 0.1885-0.1933 against 0.1897-0.1920 ms)

// The 16 complex values of a lane: for column group g the elements / outputs of slots 0..3 and 4..7 as the two vectors of four
// dwords the matrix core takes as B operands of its two K halves -- kept in that form through the whole block, so that no pass
// gathers its operands from scattered registers (h2 v[2][8]: 250 v_mov per block; vectors of eight halves: every half packed and
// unpacked on its own, 850 shifts per block).
// COMPILER TRAP (clang 22, host and device, -O0 and -O2): __builtin_bit_cast (h2, vec[i]) and __builtin_bit_cast (h2, vec.y) of an
// ext_vector_type element return ELEMENT 0 whatever i is -- an afternoon of "deterministic garbage whatever the schedule".  Through
// a scalar temporary it is right: never bit_cast a vector element expression directly.
struct MfV {
	mf_u4 b[2][2];
};
PROT_DEV h2 mf_get (const MfV& v, int g, int s)
{
	const unsigned x = v.b[g][s >> 2][s & 3];
	return __builtin_bit_cast (h2, x);
}
PROT_DEV void mf_set (MfV& v, int g, int s, h2 x) { v.b[g][s >> 2][s & 3] = __builtin_bit_cast (unsigned, x); }
PROT_DEV unsigned mf_bits (const MfV& v, int g, int s) { return v.b[g][s >> 2][s & 3]; }

// one radix-16 pass of this wave's 64 columns: element slots -> output slots; LAST = the last pass of a block: its products stay
// in fp32 (acc[g][2 u], acc[g][2 u + 1] = output slot u of column group g) -- the overlap-add and the rotation take them as they
// are: no rounding to half and back, and 50 conversions fewer per block
template <bool LAST = false>
PROT_DEV void mf_pass (const __attribute__ ((address_space (3))) mf_h8* mat, int l, MfV& v, mf_f16* acc_out = nullptr)
{
	const mf_h8 a0 = mat[l];      // [kh][h'][m'][8]: K half 0 (lane l = 32 h' + m' reads entry l: 1 KiB of LDS in lane order)
	const mf_h8 a1 = mat[64 + l]; // K half 1
	mf_f16 z;
#pragma unroll
	for (int r = 0; r < 16; ++r) {
		z[r] = 0.f;
	}
	mf_f16 acc[2];
#pragma unroll
	for (int g = 0; g < 2; ++g) {
		acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a0, __builtin_bit_cast (mf_h8, v.b[g][0]), z, 0, 0, 0);
	}
#pragma unroll
	for (int g = 0; g < 2; ++g) {
		acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a1, __builtin_bit_cast (mf_h8, v.b[g][1]), acc[g], 0, 0, 0);
	}
	if (LAST) {
		acc_out[0] = acc[0];
		acc_out[1] = acc[1];
		return;
	}
#pragma unroll
	for (int g = 0; g < 2; ++g) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			mf_set (v, g, u, __builtin_bit_cast (h2, __builtin_amdgcn_cvt_pkrtz (acc[g][2 * u], acc[g][2 * u + 1])));
		}
	}
}

template <int HS>
__global__ __launch_bounds__ (256, kMfWgPerCu) void olapolyh_mfma_kernel (const OlaParamsH p)
{
	typedef _Float16 S;
	typedef pk16     O;
	typedef MfRows<HS> R;
	constexpr int BLK = 512 * HS;
	constexpr int NT  = 16 - HS;
	constexpr int HH  = HS / 2; // non-zero input rows = stored output rows per lane and column
	constexpr int NTH = NT / 2; // tail rows per lane and column
	constexpr int DH  = R::DH;  // dry delay - convolution delay, in rows of 512 frames (the host checks that the engine's is this)
	static_assert (HS == 8 || HS == 10 || HS == 12 || HS == 14, "rows per block");
	static_assert (DH <= HH && NTH <= HH, "the carried rows are lane half 1's");

	__shared__ __attribute__ ((aligned (16))) h2 lds[kLdsExchange];
	__shared__ h2 Hs[kMfHsLen + 3];
	__shared__ __attribute__ ((aligned (16))) _Float16 mats[kMfMat * kMfMatH];
	__shared__ h2 tw2s[2 * 8 * 16]; // W256^{j0 b}, b = onat (h, u), at [h][u][j0]: the lanes n and n + 16 read the same word (a broadcast)

	const int t0 = threadIdx.x;
	const int c  = blockIdx.y;
	int64_t   r0, rlen;
	run_of_workgroup (p, r0, rlen);
	if (r0 >= p.n_out) {
		hist_write_slice (p, blockIdx.y, threadIdx.x);
		return;
	}
	const int64_t r1 = (r0 + rlen < p.n_out) ? r0 + rlen : p.n_out;

	const S* __restrict__ in   = p.in + (int64_t)c * p.in_stride;
	const S* __restrict__ hist = p.hist ? p.hist + (int64_t)c * p.hist_len : nullptr;
	S* __restrict__ out        = p.out + (int64_t)c * p.out_stride;
	const int64_t n_in = p.n_in;
	const int     Dc = p.conv_delay, Dd = p.dry_delay; // both even, Dd - Dc = 512 DH (the host checks)
	const bool in_al4  = (reinterpret_cast<uintptr_t> (in) & 3) == 0;
	const bool out_al4 = (reinterpret_cast<uintptr_t> (out) & 3) == 0;
	const float ca = p.rot ? p.rot[2 * c] : p.rot_v[2 * c], sa = p.rot ? p.rot[2 * c + 1] : p.rot_v[2 * c + 1];

	// lane geometry: lane l = 32 h + n of wave w.  Pass A / inverse A: columns t = 64 w + 2 n + g (the lane's two columns are
	// neighbours: one 8-byte access loads, and one stores, both); passes B / C: column (j0, ab + 2 g) resp. (ab + 2 g, j0 as b)
	const int l = t0 & 63, w = t0 >> 6, h = l >> 5, n = l & 31;
	const int j0 = n & 15, ab = 4 * w + (n >> 4);

	// ---- tables: the five matrices, the spectrum and the W256 twiddles into LDS, the outer twiddles into registers
	const f2* __restrict__ twp = reinterpret_cast<const f2*> (p.tw);
	{
		// matrix m, entry (M', K') -> half at (((m * 2 + kh) * 2 + h') * 32 + M') * 8 + i with K' = 16 kh + 8 h' + i
		// rows: output index P[M' >> 1], part M' & 1; columns: element Q[K' >> 1], part K' & 1
		for (int e = t0; e < kMfMat * kMfMatH; e += 256) {
			const int m = e >> 10, kh = (e >> 9) & 1, hp = (e >> 8) & 1, mp = (e >> 3) & 31, i = e & 7;
			const int kp = 16 * kh + 8 * hp + i;
			const int rs = mp >> 1, cs = kp >> 1;                     // row / column slot numbers 0..15
			// slot number -> (half, slot-in-half): rows 4 (u >> 1) + 2 h + (u & 1), columns 8 (s >> 2) + 4 h + (s & 3)
			const int rh = (rs >> 1) & 1, ru = 2 * (rs >> 2) + (rs & 1);
			const int chh = (cs >> 2) & 1, csl = 4 * (cs >> 3) + (cs & 3);
			// P: identity except the last inverse pass (the block's output rows); Q: identity (B, C forward; B inverse), the block's input
			// rows (A forward), the natural OUTPUT order (C and A inverse: their input slots are what output slots held)
			const int o  = m == 4 ? R::O[rh][ru] : rs;
			const int el = m == 0 ? mf_in_row<HS> (chh, csl) : (m == 2 || m == 4) ? mf_onat (chh, csl) : cs;
			const f2   wv  = twp[256 * ((o * el) & 15)];               // W16^{o el} = cos - i sin
			const float cw = wv.x, sw = m >= 2 ? wv.y : -wv.y;       // forward: s = +sin; inverse: -sin   (wv.y = -sin)
			const int  po = mp & 1, pe = kp & 1;
			mats[e] = (_Float16)(po == pe ? cw : (po == 0 ? sw : -sw));
		}
		for (int i = t0; i <= 2048; i += 256) {
			const f2 hv = reinterpret_cast<const f2*> (p.H)[i];
			Hs[(i & ~255) | ((i & 15) << 4) | ((i >> 4) & 15)] = O::mk (hv.x, hv.y); // low hex digits swapped: lanes read neighbours
		}
		const int th = t0 >> 7, tu = (t0 >> 4) & 7, tj = t0 & 15; // 256 threads = 2 x 8 x 16 entries
		const f2  wv = twp[16 * ((tj * (mf_onat (0, tu) + 2 * th)) & 255)];
		tw2s[t0]     = O::mk (wv.x, wv.y);
	}
	h2 tw1[2][8];
#pragma unroll
	for (int u = 0; u < 8; ++u) {
		const int a = mf_onat (0, u) + 2 * h;
#pragma unroll
		for (int g = 0; g < 2; ++g) {
			const int t  = 64 * w + 2 * n + g;
			const f2  wv = twp[(t * a) & 4095];
			tw1[g][u]    = O::mk (wv.x, wv.y);
		}
	}
	const __attribute__ ((address_space (3))) mf_h8* const matp = (const __attribute__ ((address_space (3))) mf_h8*)mats;

	// mirrored bin of (ab + 2 g) + 16 j0 for c >= 8, digit-swapped (per g)
	int hq[2], hm[2];
#pragma unroll
	for (int g = 0; g < 2; ++g) {
		const int a = ab + 2 * g, q = a + 16 * j0;
		const int m = 256 - q;
		hq[g] = j0 + 16 * a; // swapped position of q
		hm[g] = q == 0 ? 256 : ((m & ~255) | ((m & 15) << 4) | ((m >> 4) & 15));
	}

	// the overlap-add tail of the block before (fp32: the last pass's products as the matrix core accumulated them) and the block's
	// last DH input pairs (lane half 1's: the dry frames of the next block's first rows)
	float    tail[2][NTH][2];
	unsigned carry[2][DH];
#pragma unroll
	for (int g = 0; g < 2; ++g) {
#pragma unroll
		for (int i = 0; i < NTH; ++i) {
			tail[g][i][0] = tail[g][i][1] = 0.f;
		}
	}

	const int n_it = 1 + (int)((r1 - r0 + BLK - 1) / BLK); // one halo block: the overlap-add tail
	int64_t   s    = r0 - BLK;

	// The block's input.  z[j] = (x[2j - 1], x[2j]) is one frame off the aligned pairs (x[2j], x[2j + 1]): for row slot i (row
	// k = i + HH h) one 8-byte load fetches the pairs AT the lane's two samples j, j + 1 (nx_cur: also the dry frames DH rows on) and
	// one dword load the pair in front of sample j (nx_prev0); the pair in front of sample j + 1 is the pair at sample j.
	unsigned nx_cur[2][HH], nx_prev0[HH];
	auto load_block = [&] (int64_t a0, int tt) {
		const bool fast = in_al4 && (a0 - 2 >= 0) && (a0 + BLK <= n_in);
		const int  lw = tt >> 6, ll = tt & 63;
		if (fast) {
			const PROT_GLOBAL unsigned* __restrict__ b = reinterpret_cast<const PROT_GLOBAL unsigned*> (uniform_ptr (in + a0 - 2));
			const unsigned                           o = (unsigned)(64 * lw + 2 * (ll & 31) + 256 * HH * (ll >> 5));
#pragma unroll
			for (int i = 0; i < HH; ++i) {
				nx_prev0[i]   = b[o + 256u * i];
				const mf_u2 q = *reinterpret_cast<const PROT_GLOBAL mf_u2a*> (b + o + 256u * i + 1u);
				nx_cur[0][i]  = q.x;
				nx_cur[1][i]  = q.y;
			}
		} else {
#pragma unroll
			for (int i = 0; i < HH; ++i) {
				const int     t = 64 * lw + 2 * (ll & 31);
				const int64_t a = a0 + 2 * (int64_t)(t + 256 * (i + HH * (ll >> 5)));
				nx_prev0[i]  = __builtin_bit_cast (unsigned, O::mk (0.f, ldv (in, n_in, hist, p.hist_len, a - 1)));
				nx_cur[0][i] = __builtin_bit_cast (unsigned, O::mk (ldv (in, n_in, hist, p.hist_len, a), ldv (in, n_in, hist, p.hist_len, a + 1)));
				nx_cur[1][i] = __builtin_bit_cast (unsigned, O::mk (ldv (in, n_in, hist, p.hist_len, a + 2), ldv (in, n_in, hist, p.hist_len, a + 3)));
			}
		}
	};
	// the pairs DH rows in front of the first block: the dry frames of its first rows (lane half 1's slots HH - DH .. HH - 1 of the
	// block before, had there been one)
#pragma unroll
	for (int i = 0; i < DH; ++i) {
		const int64_t a = s - Dc - 2 * 256 * (int64_t)DH + 2 * (int64_t)(64 * w + 2 * n + 256 * i);
#pragma unroll
		for (int g = 0; g < 2; ++g) {
			carry[g][i] = __builtin_bit_cast (unsigned, O::mk (ldv (in, n_in, hist, p.hist_len, a + 2 * g), ldv (in, n_in, hist, p.hist_len, a + 2 * g + 1)));
		}
	}
	load_block (s - Dc, t0);
	__syncthreads ();

	for (int it = 0; it < n_it; ++it, s += BLK) {
		int tt = t0;
		// (the thread id opaque per iteration: per-lane address terms are recomputed, not carried in registers -- hoisted as loop
		// invariants they spill: 76-252 bytes of scratch per lane and 0.217 ms instead of 0.190)
		asm volatile ("" : "+v"(tt));
		tt &= 255;
		const int l = tt & 63, w = tt >> 6, h = l >> 5, n = l & 31;
		const int j0 = n & 15, ab = 4 * w + (n >> 4);
		MfV      v;
		unsigned cur[2][HH];
		// ---- z = (x[2j - 1], x[2j]): high half of the pair in front, low half of the pair at the sample
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			if (i < HH) {
				const int ii = i < HH ? i : 0;
				cur[0][ii]   = nx_cur[0][ii];
				cur[1][ii]   = nx_cur[1][ii];
				mf_set (v, 0, i, __builtin_bit_cast (h2, __builtin_amdgcn_alignbit (nx_cur[0][ii], nx_prev0[ii], 16u)));
				mf_set (v, 1, i, __builtin_bit_cast (h2, __builtin_amdgcn_alignbit (nx_cur[1][ii], nx_cur[0][ii], 16u)));
			} else {
				mf_set (v, 0, i, O::mk (0.f, 0.f));
				mf_set (v, 1, i, O::mk (0.f, 0.f));
			}
		}
		// ---- forward pass A, outer twiddles, exchange 1: slot a * 272 + t
		mf_pass (matp + 0 * 128, l, v);
		{
			// (the lane's two columns are neighbours in the row: ONE 8-byte write per output -- as two dword writes the lanes n and
			// n + 16 met in one bank, SQ_LDS_BANK_CONFLICT a quarter of the LDS cycles)
			mf_lds2* wr = (mf_lds2*)&lds[(2 * h) * kL1 + 64 * w + 2 * n];
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				h2 x0 = mf_get (v, 0, u), x1 = mf_get (v, 1, u);
				O::template cmul2<1> (x0, tw1[0][u], x1, tw1[1][u]);
				wr[mf_onat (0, u) * (kL1 / 2)] = mf_u2{ __builtin_bit_cast (unsigned, x0), __builtin_bit_cast (unsigned, x1) };
			}
		}
		__syncthreads ();
		{
			const mf_lds* rd = (const mf_lds*)&lds[ab * kL1 + j0 + 64 * h];
#pragma unroll
			for (int g = 0; g < 2; ++g) {
#pragma unroll
				for (int sl = 0; sl < 8; ++sl) {
					mf_set (v, g, sl, rd[2 * g * kL1 + 16 * mf_enat (0, sl)]);
				}
			}
		}
		// the next block's frames go out behind the first barrier, three quarters of a block ahead of their use
		if (it + 1 < n_it) {
			load_block (s + BLK - Dc, tt);
		}
		// ---- forward pass B, W256 twiddles, wave-local exchange 2: slot a * 272 + b * 17 + j0
		mf_pass (matp + 1 * 128, l, v);
		{
			mf_lds*       wr = (mf_lds*)&lds[ab * kL1 + j0 + 2 * kG * h];
			const mf_lds* tq = (const mf_lds*)&tw2s[128 * h + j0];
#pragma unroll
			for (int u = 0; u < 8; u += 2) {
				const h2 q0 = tq[16 * u], q1 = tq[16 * (u + 1)];
#pragma unroll
				for (int g = 0; g < 2; ++g) {
					h2 x0 = mf_get (v, g, u), x1 = mf_get (v, g, u + 1);
					O::template cmul2<1> (x0, q0, x1, q1);
					wr[2 * g * kL1 + kG * mf_onat (0, u)]     = x0;
					wr[2 * g * kL1 + kG * mf_onat (0, u + 1)] = x1;
				}
			}
		}
		wave_sync ();
		{
			const mf_lds* rd = (const mf_lds*)&lds[ab * kL1 + kG * j0 + 4 * h];
#pragma unroll
			for (int g = 0; g < 2; ++g) {
#pragma unroll
				for (int sl = 0; sl < 8; ++sl) {
					mf_set (v, g, sl, rd[2 * g * kL1 + mf_enat (0, sl)]);
				}
			}
		}
		// ---- forward pass C; the lane holds bins (ab + 2 g) + 16 j0 + 256 c, c = onat (h, u): times the spectrum (/ 4096)
		mf_pass (matp + 1 * 128, l, v);
#pragma unroll
		for (int g = 0; g < 2; ++g) {
			const mf_lds* lo = (const mf_lds*)&Hs[hq[g] + 512 * h];
			const mf_lds* hi = (const mf_lds*)&Hs[hm[g] + 512 * (1 - h)];
			h2 hv[8];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				hv[u] = lo[256 * mf_onat (0, u)];
			}
#pragma unroll
			for (int u = 4; u < 8; ++u) { // c = 8 + 4 x + 2 h + y: bin 256 (15 - c) + mirrored low part
				hv[u] = hi[256 * (5 - 4 * ((u >> 1) & 1) - (u & 1))];
			}
#pragma unroll
			for (int u = 0; u < 8; u += 2) {
				h2 x0 = mf_get (v, g, u), x1 = mf_get (v, g, u + 1);
				if (u < 4) {
					O::template cmul2<1> (x0, hv[u], x1, hv[u + 1]);
				} else {
					O::template cmul2<-1> (x0, hv[u], x1, hv[u + 1]);
				}
				mf_set (v, g, u, x0);
				mf_set (v, g, u + 1, x1);
			}
		}
		// ---- inverse pass C (no exchange: its input slots are these output slots), conj W256, wave-local exchange
		mf_pass (matp + 2 * 128, l, v);
		{
			mf_lds*       wr = (mf_lds*)&lds[ab * kL1 + kG * j0 + 2 * h];
			const mf_lds* tq = (const mf_lds*)&tw2s[128 * h + j0];
#pragma unroll
			for (int u = 0; u < 8; u += 2) {
				const h2 q0 = tq[16 * u], q1 = tq[16 * (u + 1)];
#pragma unroll
				for (int g = 0; g < 2; ++g) {
					h2 x0 = mf_get (v, g, u), x1 = mf_get (v, g, u + 1);
					O::template cmul2<-1> (x0, q0, x1, q1);
					wr[2 * g * kL1 + mf_onat (0, u)]     = x0;
					wr[2 * g * kL1 + mf_onat (0, u + 1)] = x1;
				}
			}
		}
		wave_sync ();
		{
			const mf_lds* rd = (const mf_lds*)&lds[ab * kL1 + j0 + 4 * kG * h];
#pragma unroll
			for (int g = 0; g < 2; ++g) {
#pragma unroll
				for (int sl = 0; sl < 8; ++sl) {
					mf_set (v, g, sl, rd[2 * g * kL1 + kG * mf_enat (0, sl)]);
				}
			}
		}
		// ---- inverse pass B, exchange 1' (the outer twiddle belongs to (t, a): the reader applies it)
		mf_pass (matp + 3 * 128, l, v);
		{
			wave_sync ();
			mf_lds* wr = (mf_lds*)&lds[ab * kL1 + j0 + 32 * h];
#pragma unroll
			for (int g = 0; g < 2; ++g) {
#pragma unroll
				for (int u = 0; u < 8; ++u) {
					wr[2 * g * kL1 + 16 * mf_onat (0, u)] = mf_get (v, g, u);
				}
			}
		}
		__syncthreads ();
		{
			const mf_lds2* rd = (const mf_lds2*)&lds[(2 * h) * kL1 + 64 * w + 2 * n];
#pragma unroll
			for (int sl = 0; sl < 8; ++sl) {
				const mf_u2    q  = rd[mf_onat (0, sl) * (kL1 / 2)];
				const unsigned qx = q.x, qy = q.y; // (NOT __builtin_bit_cast (h2, q.y): see mf_get)
				h2 x0 = __builtin_bit_cast (h2, qx), x1 = __builtin_bit_cast (h2, qy);
				O::template cmul2<-1> (x0, tw1[0][sl], x1, tw1[1][sl]);
				mf_set (v, 0, sl, x0);
				mf_set (v, 1, sl, x1);
			}
		}
		// ---- inverse pass A: rows O_h[u] of the lane's columns
		mf_f16 yo[2];
		mf_pass<true> (matp + 4 * 128, l, v, yo);
		// ---- overlap-add (fp32), dry frames from the input pairs the lane holds, rotation, store (both columns in one 8-byte access)
		const bool emit = it >= 1;
		const bool full = s + BLK <= r1;
		unsigned   o2[HH][2];
#pragma unroll
		for (int g = 0; g < 2; ++g) {
#pragma unroll
			for (int i = 0; i < HH; ++i) {
				const float u0 = yo[g][2 * i] + (i < NTH ? tail[g][i < NTH ? i : 0][0] : 0.f), u1 = yo[g][2 * i + 1] + (i < NTH ? tail[g][i < NTH ? i : 0][1] : 0.f);
				// (the two lane halves take their dry pairs from different slots: a select where they differ)
				const int s0 = R::DSRC[0][i], s1 = R::DSRC[1][i]; // (compile-time after unrolling)
				const unsigned d0 = s0 >= 0 ? cur[g][s0 >= 0 ? s0 : 0] : carry[g][s0 < 0 ? -1 - s0 : 0];
				const unsigned d1 = s1 >= 0 ? cur[g][s1 >= 0 ? s1 : 0] : carry[g][s1 < 0 ? -1 - s1 : 0];
				const h2       d  = __builtin_bit_cast (h2, s0 == s1 ? d0 : (h ? d1 : d0));
				o2[i][g] = __builtin_bit_cast (unsigned, O::mk (ca * (float)d.x + sa * u0, ca * (float)d.y + sa * u1));
			}
#pragma unroll
			for (int i = 0; i < NTH; ++i) {
				tail[g][i][0] = yo[g][2 * (HH + i)];
				tail[g][i][1] = yo[g][2 * (HH + i) + 1];
			}
#pragma unroll
			for (int i = 0; i < DH; ++i) {
				carry[g][i] = cur[g][HH - DH + i];
			}
		}
		if (emit) {
			if (full && out_al4) {
				PROT_GLOBAL unsigned* __restrict__ ob = reinterpret_cast<PROT_GLOBAL unsigned*> (uniform_ptr (out + s));
#pragma unroll
				for (int i = 0; i < HH; ++i) {
					const unsigned row = h ? (unsigned)R::O[1][i] : (unsigned)R::O[0][i];
					*reinterpret_cast<PROT_GLOBAL mf_u2a*> (ob + (unsigned)(64 * w + 2 * n) + 256u * row) = mf_u2{ o2[i][0], o2[i][1] };
				}
			} else {
#pragma unroll
				for (int i = 0; i < HH; ++i) {
					const int row = h ? R::O[1][i] : R::O[0][i];
#pragma unroll
					for (int g = 0; g < 2; ++g) {
						const int64_t pos = s + 2 * (int64_t)(64 * w + 2 * n + g + 256 * row);
						const h2      o   = __builtin_bit_cast (h2, o2[i][g]);
						if (pos < r1) {
							out[pos] = o.x;
						}
						if (pos + 1 < r1) {
							out[pos + 1] = o.y;
						}
					}
				}
			}
		}
	}
	hist_write_slice (p, blockIdx.y, threadIdx.x);
}

int mfma_f16_wg_per_cu () { return kMfWgPerCu; }

// same contract as launch_ola_poly_f16 (n_part = 1, steady rotation, no accumulate) plus: dry_delay - conv_delay = 512 DH of the
// block (the dry frames come from the input pairs in registers) -- mfma_f16_takes () says whether a call qualifies
bool mfma_f16_takes (const OlaParamsH& p, int hop_rows)
{
	const int dh = hop_rows == 8 ? MfRows<8>::DH : hop_rows == 10 ? MfRows<10>::DH : hop_rows == 12 ? MfRows<12>::DH : hop_rows == 14 ? MfRows<14>::DH : -1;
	return dh > 0 && !(p.conv_delay & 1) && p.dry_delay - p.conv_delay == 512 * dh && !p.accumulate && !p.hilbert_only && p.rot_s == nullptr &&
	       p.surv == nullptr && p.run_stride <= 0;
}

hipError_t launch_ola_poly_f16_mfma (const OlaParamsH& p, int hop_rows, int n_chn, hipStream_t s)
{
	if (p.n_out <= 0) {
		return hipSuccess;
	}
	if (!mfma_f16_takes (p, hop_rows) || (p.run_frames % (512 * hop_rows)) != 0) {
		return hipErrorInvalidValue;
	}
	const int64_t runs = (p.n_out + p.run_frames - 1) / p.run_frames;
	dim3          grid ((unsigned)runs, (unsigned)n_chn), block (256);
	switch (hop_rows) {
		case 8: hipLaunchKernelGGL ((olapolyh_mfma_kernel<8>), grid, block, 0, s, p); break;
		case 10: hipLaunchKernelGGL ((olapolyh_mfma_kernel<10>), grid, block, 0, s, p); break;
		case 12: hipLaunchKernelGGL ((olapolyh_mfma_kernel<12>), grid, block, 0, s, p); break;
		case 14: hipLaunchKernelGGL ((olapolyh_mfma_kernel<14>), grid, block, 0, s, p); break;
		default: return hipErrorInvalidValue;
	}
	return hipGetLastError ();
}

} // namespace prot
