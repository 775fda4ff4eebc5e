// prot_olapoly.hip -- olapoly_kernel: the POLYPHASE overlap-add kernel with a spectrum table (FIRs of 8192 taps and more, CLI
// blksiz 1024, PROT_CFG_OVERLAP_ADD, the vector-pipe fp16 path) and its launchers.  Split from prot_kernels.hip in round 6 (same
// code); the kernels of the path by file are listed at the top of prot_kernels.hip.
#include "prot_fft4096.h"

#include <algorithm>
#include <cstdlib>

namespace prot {

// --------------------------------------------------------------------------------------
// olapoly_kernel -- POLYPHASE form: the default for every FIR length except 4096 taps.
//
// The Hilbert FIR is zero at every even tap (src/phaserotate.c:374-401, cli/phase-rotate.cc:
// 128-165: h[n] ~ cot(pi m/N) for odd m = n - N/2 only, N/2 even), so
//     v[2j]   = sum_s g[s] x[2(j-s) - 1],     v[2j+1] = sum_s g[s] x[2(j-s)],     g[s] = h[2s+1]:
// even and odd output frames are the SAME half-length filter g applied to the odd resp. even
// input frames.  Reading the stream as complex samples  z[j] = x[2j-1] + i x[2j]  (a plain
// reinterpretation of consecutive frames) turns the convolution into ONE complex convolution
// g * z = v[2j] + i v[2j+1]  -- so the two real signals that ride in the re/im of the complex
// transform are the two phases of one stream, the filter is half as long (3072 / 4096 taps fit
// ONE partition: no second spectrum multiply, no delayed spectrum in registers, half the
// table), and a block is 512 HS consecutive frames of one stream (one halo block per run):
// the zero padding only has to cover the tail of g, so 1536 / 1024 / 512 taps take HS = 10 /
// 12 / 14 rows of 256 new samples per 4096-point transform instead of 8.
// FIRs of 8192 taps and more take NP = 2 partitions of g per launch (frequency-domain delay
// line as in ola4096p2_kernel), i.e. half the launches of the direct form.
// The dry frames (x delayed by dry_delay) pair up as (x[2j], x[2j+1]), one frame off the
// transform's input pairs (x[2j-1], x[2j]): a thread needs p = x[a-1], q = x[a], c = x[a+1].
// The stream is read exactly once as aligned 8-byte pairs (q, c); p is the c of lane t-1 (one
// DPP wave shift at the point of use; lane 0 of a wave gets it from a scalar load).  (p, q) feeds
// the transform, (q, c) is kept for the dry path.
// --------------------------------------------------------------------------------------
// WPS = workgroups per CU (= waves per SIMD) the register budget is sized for: 2 (<= 256 VGPRs, tables fetched a whole
// butterfly ahead) or 3 (<= 168 VGPRs: tables fetched in small batches, 3 x 53 KiB of LDS is the whole CU)
// (LEAN_: the register-lean arithmetic of the three-per-CU form -- split outer twiddles, smaller table batches; a template
// parameter of its own so that a two-per-CU instantiation can compute the SAME BITS as its three-per-CU twin: MODE_SWEEP of
// CLI blksiz 1024 against the separate Hilbert launch)
template <class T, int NP, int HS, int MODE, int DH, bool ACC, int WPS = 2, bool LEAN_ = (WPS >= 3)>
__global__ __launch_bounds__ (256, WPS) void olapoly_kernel (const OlaParamsT<typename T::samp> p)
{
	constexpr bool LEAN = LEAN_;
	typedef typename T::cplx C;  // one complex of the transform = one PAIR of stream frames in HBM
	typedef typename T::ops  O;
	typedef typename T::samp S;
	// frame pairs, dry path, overlap-add tail and rotation are fp32 registers for both precisions
	auto to_f2 = [] (C v) { return mkf2 ((float)v.x, (float)v.y); };
	auto to_sp = [] (f2 v) { return C{ (S)v.x, (S)v.y }; };
	typedef PROT_LDS_PTR (T) VC;
	// HS rows of 256 complex samples are new per block, 16 - HS rows are the overlap-add tail: g may
	// have up to 256 (16 - HS) + 1 taps.  HS = 8 for 2048-tap partitions; a shorter g (1536 taps for
	// the 3072-tap LV2 filter: HS = 10; 1024: 12; 512: 14) takes proportionally more frames per
	// transform pair -- the zero padding of the OLA is only as long as the filter needs.
	constexpr int BLK = 512 * HS; // frames per block
	constexpr int NT  = 16 - HS;  // tail rows
	static_assert (NP == 1 || NP == 2, "one or two partitions of g per launch");
	static_assert (HS == 8 || ((HS == 10 || HS == 12 || HS == 14) && NP == 1), "long hops only for a single short partition");
	static_assert (DH >= 0 && DH <= 8, "dry delay difference / 512 within one block, or 0");
	constexpr int NDH = DH > 0 ? DH : 1;

	__shared__ C lds[kLdsRegion];
	__shared__ C twm[256];
	__shared__ C Hs[NP][2049];

	const int     t  = threadIdx.x;
	const int     c  = blockIdx.y;
#ifdef PROT_STAMPS
	const uint64_t st_entry = __builtin_amdgcn_s_memrealtime ();
#endif
	// The run of this workgroup.  One-round launches (every workgroup resident at once, p.rank_cu > 0) give the runs
	// unequal lengths by DISPATCH RANK: the hardware arbitrates oldest-first, so of the three workgroups of a CU the one
	// dispatched first runs ahead and the last one finishes alone, long after (ends at 196 / 281 / 347 us for equal
	// runs at the headline).  Workgroups are dealt out in id order, one per CU per round, so id / CUs is the rank;
	// if a dispatcher ever does it differently the lengths are merely not ideal -- the runs still tile the stream.
	// Which run: not simply blockIdx.x.  Workgroups are dealt round-robin over the 8 XCDs, so XCD k would own the runs
	// k, k + 8, ... -- with runs of about 1 MiB that is one fixed residue of the address modulo 8 MiB per XCD.  Rotating
	// the position inside each group of eight consecutive runs by the group's number gives every XCD all eight residues
	// in turn: measured -2 % (interleaved A/B of the shipped kernel).  (The 1.5x spread between XCDs that an earlier
	// version of this comment quoted came from a stamped diagnostic build whose end-of-kernel atomics queued on one
	// cache line; the timeline-only build shows the XCDs within +-3 %, tools/attic/xcd_timeline.py.)
	int bx = (int)blockIdx.x;
	if ((bx | 7) < (int)gridDim.x) {
		bx = (bx & ~7) | ((bx + (bx >> 3)) & 7);
	}
	int64_t r0, rlen = p.run_frames;
	if (p.rank_cu > 0) {
		const int id = (int)blockIdx.y * (int)gridDim.x + bx, id0 = (int)blockIdx.y * (int)gridDim.x;
		r0           = 0;
		const int last = p.n_rank - 1;
#pragma unroll
		for (int k = 0; k < 4; ++k) { // runs of rank k in front of this one, within the channel
			const int lo = k * p.rank_cu > id0 ? k * p.rank_cu : id0;
			const int hi = (k >= last || (k + 1) * p.rank_cu > id) ? id : (k + 1) * p.rank_cu;
			r0 += (k <= last && hi > lo) ? (int64_t)(hi - lo) * p.rank_frames[k] : 0;
		}
		const int rk = id / p.rank_cu > last ? last : id / p.rank_cu;
		rlen         = p.rank_frames[rk];
	} else {
		r0 = (int64_t)bx * p.run_frames;
	}
	if (p.run_stride > 0) { // a sparse sample of the stream (the fused sweep's bound-seeding pass): run g starts at g * run_stride
		r0   = (int64_t)blockIdx.x * p.run_stride;
		rlen = p.run_frames;
	}
	if (r0 >= p.n_out) {
		hist_write_slice (p, blockIdx.y, threadIdx.x);
		if (MODE == MODE_SWEEP && (t & 63) == 0) {
			p.surv_cnt[((int64_t)c * gridDim.x + blockIdx.x) * 4 + (t >> 6)] = 0u;
		}
		return;
	}
	const int64_t r1 = (r0 + rlen < p.n_out) ? r0 + rlen : p.n_out;

	const S* __restrict__ in   = p.in + (int64_t)c * p.in_stride;
	const S* __restrict__ hist = p.hist ? p.hist + (int64_t)c * p.hist_len : nullptr;
	S* __restrict__ out        = p.out + (int64_t)c * p.out_stride;
	const int64_t n_in             = p.n_in;
	const int     Dc = p.conv_delay, Dd = p.dry_delay; // Dc even (host guarantees)
	// 8-byte alignment of the frame pairs (block starts and Dc are even): uniform per workgroup
	const bool in_al8  = (reinterpret_cast<uintptr_t> (in) & (2 * sizeof (S) - 1)) == 0;  // a frame pair is one aligned access
	const bool out_al8 = (reinterpret_cast<uintptr_t> (out) & (2 * sizeof (S) - 1)) == 0;

	// MODE_SWEEP (round 4; the LAST launch of the Hilbert branch of a long FIR, CLI blksiz 16384 / 32768): the complete Hilbert
	// value of a frame -- what the earlier launches left in `out` plus this launch's share -- is not stored but TESTED with its dry
	// frame against the call's bound, and the surviving (dry, hilbert) pairs go to this wave's list (prot_olasave.hip's MODE_SWEEP
	// does the same in generated code; the lists, the plan and the evaluation are prot_sweep.hip's)
	__shared__ float sw_lo_s[4];
	float            sw_thr2 = 0.f, sw_raw = 0.f;
	unsigned         sw_cnt  = 0u;
	if constexpr (MODE == MODE_SWEEP) {
		float lo = 1.0e18f;
		if (!p.sw_thr_zero) {
			for (int i = t; i < p.sw_n_angles; i += 256) {
				lo = fminf (lo, __builtin_nontemporal_load (&p.sw_peaks[(int64_t)c * kMaxSample + p.sw_angles[i]]));
			}
		} else {
			lo = 0.f;
		}
		for (int m = 1; m < 64; m <<= 1) {
			lo = fminf (lo, __shfl_xor (lo, m));
		}
		if ((t & 63) == 0) {
			sw_lo_s[t >> 6] = lo;
		}
		__syncthreads ();
		lo      = fminf (fminf (sw_lo_s[0], sw_lo_s[1]), fminf (sw_lo_s[2], sw_lo_s[3]));
		sw_thr2 = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (lo * lo * 0.9999f)));
		if (p.sw_raw_slot >= 0 && p.sw_raw_tail && blockIdx.x == 0) {
			// the call's last dry_delay input frames: the rows reach them only as the next chunk's history (see prot_sweep.hip)
			for (int64_t i = n_in - Dd + t; i < n_in; i += 256) {
				sw_raw = fmaxf (sw_raw, i >= 0 ? fabsf ((float)in[i]) : 0.f);
			}
		}
	}
	const f2* __restrict__ twp = reinterpret_cast<const f2*> (p.tw);
	C tw1[16];
	// the tables of this workgroup (outer twiddles in registers, W256 and the filter spectrum in LDS): set up BEHIND the
	// first block's stream loads (below), so that the two latencies of the prologue overlap
	auto setup_tables = [&] () {
	if (LEAN) { // split form (pkt_fft16_fwd_pruned_twout): [1..3] = W^{t i}, [5..7] = W^{4 t i}; the rest is never read
#pragma unroll
		for (int i = 1; i < 4; ++i) {
			tw1[i]     = O::mk (twp[(t * i) & 4095].x, twp[(t * i) & 4095].y);
			tw1[4 + i] = O::mk (twp[(4 * t * i) & 4095].x, twp[(4 * t * i) & 4095].y);
		}
	} else {
#pragma unroll
		for (int k = 1; k < 16; ++k) {
			tw1[k] = O::mk (twp[(t * k) & 4095].x, twp[(t * k) & 4095].y);
		}
	}
	twm[t] = O::mk (twp[16 * (((t & 15) * (t >> 4)) & 255)].x, twp[16 * (((t & 15) * (t >> 4)) & 255)].y);
	for (int i = t; i <= 2048; i += 256) {
		const int j = swap_lo_digits (i);
#pragma unroll
		for (int pp = 0; pp < NP; ++pp) {
			const f2 hv = reinterpret_cast<const f2*> (p.H)[pp * 4096 + i];
			Hs[pp][j] = O::mk (hv.x, hv.y);
		}
	}
	};
	float ca = 0.f, sa = 1.f;
	if (MODE == MODE_ROT) {
		ca = p.rot ? p.rot[2 * c] : p.rot_v[2 * c];
		sa = p.rot ? p.rot[2 * c + 1] : p.rot_v[2 * c + 1];
	}

	const int tl = t & 15, th = t >> 4;
	const int q  = th + 16 * tl;
	const int mq = q == 0 ? 256 : swap_lo_digits (256 - q); // mirrored bin of (th + 16 tl + 256 k), k >= 8
	const int n_it = NP + (int)((r1 - r0 + BLK - 1) / BLK);  // NP halo blocks: OLA tail (+ X_{b-1})
	int64_t   s    = r0 - (int64_t)NP * BLK;

	f2 acc[NT];
	C  zp[NP == 2 ? 16 : 1];
#pragma unroll
	for (int i = 0; i < NT; ++i) {
		acc[i] = mkf2 (0.f, 0.f);
	}
#pragma unroll
	for (int i = 0; i < (NP == 2 ? 16 : 1); ++i) {
		zp[i] = O::mk (0.f, 0.f);
	}
	// (q, c) pairs of the DH sample rows in front of the current block: the dry path's source
	f2 dh[NDH];
#pragma unroll
	for (int i = 0; i < NDH; ++i) {
		const int64_t a = s - Dc + 2 * (int64_t)(t + 256 * (i - DH));
		dh[i]           = DH > 0 ? mkf2 (ldv (in, n_in, hist, p.hist_len, a), ldv (in, n_in, hist, p.hist_len, a + 1)) : mkf2 (0.f, 0.f);
	}

	// frames a, a+1 of sample j = t + 256 k, a = a0 + 2 j, as the pair (q, c); the frame in front, p = x[a-1], is the c of
	// the previous sample, i.e. of lane t-1 (one DPP wave shift at the point of use), except in lane 0 of a wave:
	// those HS frames (one per row) come with ONE gather load per wave, lane k fetching row k's into nxg.
	// (A vector load on purpose: scalar loads of fresh stream data sit in lgkmcnt for a whole HBM round trip, and
	// every LDS wait of the transform behind them then waits for that.)  The bounds-checked and unaligned forms
	// load the same registers, so the loop carries one representation.
	float nxg = 0.f;
	f2    nxqc[HS];
	auto  load_block = [&] (int64_t a0, int t) { // t: the thread id, passed in so that the loop can hide its loop-invariance
		const bool     fast = (a0 - 1 >= 0) && (a0 + BLK <= n_in);
		const unsigned l    = (unsigned)t & 63u;
		const unsigned grow = l < (unsigned)HS ? l : 0u; // row whose leading frame this lane gathers
		if (PROT_ABL2 & 32) { // no stream loads
#pragma unroll
			for (int k = 0; k < HS; ++k) {
				nxqc[k] = mkf2 (0.002f * (float)(t - k), 0.003f * (float)(t + 2 * k));
			}
			nxg = 0.001f * (float)t;
		} else if (fast && in_al8) {
			// the stream exactly once, as aligned 8-byte pairs
#pragma unroll
			for (int m = 0; m < HS / 2; ++m) { // one SGPR base per 1024 frames: the immediate offset reaches 4 KiB
				const PROT_GLOBAL S* __restrict__ b = uniform_ptr (in + a0 + 1024 * m);
#pragma unroll
				for (int e = 0; e < 2; ++e) {
					nxqc[2 * m + e] = to_f2 (reinterpret_cast<const PROT_GLOBAL C*> (b)[(unsigned)t + 256u * e]);
				}
			}
			const PROT_GLOBAL S* __restrict__ b = uniform_ptr (in + a0 - 1);
			nxg = (float)b[((unsigned)t & ~63u) * 2u + 512u * grow];
		} else if (fast) {
#pragma unroll
			for (int m = 0; m < HS / 2; ++m) {
				const PROT_GLOBAL S* __restrict__ b = uniform_ptr (in + a0 + 1024 * m);
#pragma unroll
				for (int e = 0; e < 2; ++e) {
					nxqc[2 * m + e] = mkf2 ((float)b[2u * (unsigned)t + 512u * e], (float)(b + 1)[2u * (unsigned)t + 512u * e]);
				}
			}
			const PROT_GLOBAL S* __restrict__ b = uniform_ptr (in + a0 - 1);
			nxg = (float)b[((unsigned)t & ~63u) * 2u + 512u * grow];
		} else {
#pragma unroll
			for (int k = 0; k < HS; ++k) {
				const int64_t a = a0 + 2 * (int64_t)(t + 256 * k);
				nxqc[k]         = mkf2 (ldv (in, n_in, hist, p.hist_len, a), ldv (in, n_in, hist, p.hist_len, a + 1));
			}
			nxg = ldv (in, n_in, hist, p.hist_len, a0 + 2 * (int64_t)(((unsigned)t & ~63u) + 256u * grow) - 1);
		}
	};
	load_block (s - Dc, t);
	setup_tables ();
	__syncthreads ();
	Stamps st;
	st.start ();
#ifdef PROT_STAMPS
	const uint64_t st_c0 = __builtin_readcyclecounter (), st_r0 = __builtin_amdgcn_s_memrealtime ();
#endif

	const int t0 = t, tl0 = tl, th0 = th;
	for (int it = 0; it < n_it; ++it, s += BLK) {
		// the thread id, opaque per iteration: per-lane address terms (64-bit pointers of the bounds-checked paths, LDS
		// addresses of the table reads) are otherwise hoisted out of the loop as loop invariants and sit in ~30 VGPRs
		// for the whole run; recomputed in place they fold into immediate offsets or stay inside their rare branch
		int t = t0;
		asm volatile ("" : "+v"(t));
		t &= 255; // known range again: stream addresses become SGPR base + 32-bit lane offset + immediate (no 64-bit lane arithmetic)
		const int tl = t & 15, th = t >> 4;
		C  v[16];
		f2 cur[HS];
#pragma unroll
		for (int k = HS; k < 16; ++k) {
			v[k] = O::mk (0.f, 0.f);
		}
#pragma unroll
		for (int k = 0; k < HS; ++k) {
			cur[k] = nxqc[k];
			// wave_shr:1 -- lane i reads lane i-1, lane 0 keeps `old` = the gathered frame of this row
			const unsigned first = (unsigned)__builtin_amdgcn_readlane ((int)__float_as_uint (nxg), k);
			const float    pk_   = __uint_as_float (__builtin_amdgcn_update_dpp (first, __float_as_uint (nxqc[k].y), 0x138, 0xf, 0xf, false));
			v[k] = O::mk (pk_, nxqc[k].x);
		}

		// spectrum tables, HB bins per batch, one batch in flight ahead of the multiplies
		constexpr int HB = (NP == 2 || LEAN) ? 2 : 4; // two partitions: half the batch, the kernel is at the 256-VGPR limit
		C             hb[2][NP * HB];
		auto          h_fetch = [&] (int j) {
#pragma unroll
			for (int i = 0; i < HB; ++i) {
				const int k   = HB * j + i;
				const int idx = k < 8 ? t + 256 * k : 256 * (15 - k) + mq;
#pragma unroll
				for (int pp = 0; pp < NP; ++pp) {
					hb[j & 1][pp * HB + i] = ((const VC*)Hs[pp])[idx];
				}
			}
		};
		st.tick (0);
		fwd4096w<T, HS, LEAN> (v, lds, twm, tw1, t, tl, th, [&] () { if (!(PROT_ABL2 & 16)) h_fetch (0); }, st, [&] () {
			if (LEAN && it + 1 < n_it) { // the next block's frames go out right behind the first barrier: three quarters of an iteration ahead
				load_block (((PROT_ABL2 & 128) ? r0 : s + BLK) - Dc, t);
			}
		});

		// Y = Z G0 (+ Z_prev G1); bins >= 2048 use the conjugate of the mirrored bin
#pragma unroll
		for (int j = 0; j < ((PROT_ABL2 & 16) ? 0 : 16 / HB); ++j) {
			if (j + 1 < 16 / HB) {
				h_fetch (j + 1);
			}
			__builtin_amdgcn_sched_barrier (0);
			if (NP == 1) {
#pragma unroll
				for (int i = 0; i < HB; i += 2) {
					const int k = HB * j + i;
					if (k < 8) {
						O::template cmul2<1> (v[k], hb[j & 1][i], v[k + 1], hb[j & 1][i + 1]);
					} else {
						O::template cmul2<-1> (v[k], hb[j & 1][i], v[k + 1], hb[j & 1][i + 1]);
					}
				}
			} else {
#pragma unroll
				for (int i = 0; i < HB; ++i) {
					const int k = HB * j + i;
					const C   z = v[k];
					if (k < 8) {
						v[k] = O::template cmul_acc<1> (O::template cmul<1> (z, hb[j & 1][i]), zp[k], hb[j & 1][HB + i]);
					} else {
						v[k] = O::template cmul_acc<-1> (O::template cmul<-1> (z, hb[j & 1][i]), zp[k], hb[j & 1][HB + i]);
					}
					zp[NP == 2 ? k : 0] = z;
				}
			}
			__builtin_amdgcn_sched_barrier (0);
		}

		st.tick (5);
		__builtin_amdgcn_sched_barrier (0);
		if (!LEAN) {
			if (it + 1 < n_it) {
				load_block (((PROT_ABL2 & 128) ? r0 : s + BLK) - Dc, t);
			}
		}
		__builtin_amdgcn_sched_barrier (0);

		// dry path: frames (s + 2j - Dd, s + 2j + 1 - Dd) = the (q, c) pair DH sample rows back
		f2 dry[HS];
		if (DH > 0) {
#pragma unroll
			for (int k = 0; k < HS; ++k) {
				dry[k] = (k >= DH) ? cur[k >= DH ? k - DH : 0] : dh[k < DH ? k : 0];
			}
#pragma unroll
			for (int i = 0; i < NDH; ++i) {
				dh[i] = cur[HS - NDH + i];
			}
		}
		// DH = 0: the dry frames are read again from the stream (L2 / Infinity Cache: this workgroup fetched them one or two
		// blocks ago) -- late, between the last exchange and the last pass of the inverse, so that no copy of the block
		// occupies 2 HS registers through both transforms
		auto dry_reread = [&] () {
			// (MODE_SWEEP with dry_delay == conv_delay -- the last launch of blksiz 16384's Hilbert branch --: the dry frames are this
			// block's own input rows, kept in `cur`; read again they were 0.8 ms of cfg4's 5.7)
			if (DH == 0 && ((MODE != MODE_HILBERT && !ACC) || (MODE == MODE_SWEEP && Dd != Dc)) && it >= NP) {
				const int64_t d0 = s - Dd;
				if (d0 >= 0 && d0 + BLK <= n_in && in_al8 && !(Dd & 1)) {
#pragma unroll
					for (int m = 0; m < HS / 2; ++m) {
						const PROT_GLOBAL S* __restrict__ b = uniform_ptr (in + d0 + 1024 * m);
#pragma unroll
						for (int e = 0; e < 2; ++e) {
							dry[2 * m + e] = to_f2 (reinterpret_cast<const PROT_GLOBAL C*> (b)[(unsigned)t + 256u * e]);
						}
					}
				} else {
#pragma unroll
					for (int k = 0; k < HS; ++k) {
						const int64_t a = d0 + 2 * (int64_t)(t + 256 * k);
						dry[k]          = mkf2 (ldv (in, n_in, hist, p.hist_len, a), ldv (in, n_in, hist, p.hist_len, a + 1));
					}
				}
			}
		};

		st.tick (6);
		inv4096w<T, LEAN> (v, lds, twm, tw1, t, tl, th, st, dry_reread);

		f2 u[HS];
#pragma unroll
		for (int k = 0; k < HS; ++k) {
			u[k] = k < NT ? acc[k < NT ? k : 0] + to_f2 (v[k]) : to_f2 (v[k]);
		}
#pragma unroll
		for (int k = 0; k < NT; ++k) {
			acc[k] = to_f2 (v[HS + k]);
		}

		// The prefetched block (issued three quarters of an iteration ago: long here) is waited for IN FRONT of the stores.
		// vmcnt retires in order, so a wait for those loads placed behind the stores -- where the next iteration first
		// reads them -- is a wait for the stores too (s_waitcnt vmcnt(0) at the loop end: every iteration then ends with a
		// full write round trip).  Waited for here, the stores drain under the next block's forward transform.
		if (!ACC || LEAN) {
#pragma unroll
			for (int k = 0; k < HS; ++k) {
				asm volatile ("" : "+v"(nxqc[k].x), "+v"(nxqc[k].y));
			}
			asm volatile ("" : "+v"(nxg));
			__builtin_amdgcn_sched_barrier (0);
		}
		if (it >= NP && MODE == MODE_SWEEP) {
			if constexpr (MODE == MODE_SWEEP) {
				const bool    full = s + BLK <= r1;
				float2* __restrict__ const list = p.surv + (((int64_t)c * gridDim.x + blockIdx.x) * 4 + (t >> 6)) * p.surv_cap;
				if (ACC) { // what the earlier launches of the Hilbert branch stored: all rows' loads in flight before the first test
					// (fetched row by row inside the loop below, each load waited behind the list stores of the row before it)
					f2 prev[HS];
#pragma unroll
					for (int k = 0; k < HS; ++k) {
						const int64_t pos0 = s + 2 * (int64_t)(t + 256 * k);
						if (full && out_al8) {
							prev[k] = to_f2 (reinterpret_cast<const PROT_GLOBAL C*> (uniform_ptr (out + s + 512 * k))[(unsigned)t]);
						} else {
							prev[k] = mkf2 (pos0 < r1 ? (float)out[pos0] : 0.f, pos0 + 1 < r1 ? (float)out[pos0 + 1] : 0.f);
						}
					}
#pragma unroll
					for (int k = 0; k < HS; ++k) {
						u[k] = u[k] + prev[k];
					}
				}
				// a block wholly inside [sw_t_begin, r1) and behind sw_dry_from -- all but the first and last of a call -- needs none of the
				// per-frame position tests (64-bit compares and selects per lane and row: a twentieth of this launch)
				auto test_rows = [&] (auto inner) {
					constexpr bool I = decltype (inner)::value;
#pragma unroll
					for (int k = 0; k < HS; ++k) {
						const int64_t pos0 = s + 2 * (int64_t)(t + 256 * k);
						const f2      hp   = u[k];
						// the same rules as the edge blocks of prot_olasave.hip: frames in front of sw_t_begin or beyond the call are not
						// tested, the dry term is 0 while t < sw_dry_from (Q1; the raw peak takes the frame all the same).  Both frames of
						// the pair are tested first: most rows of most blocks have no survivor in the whole wave and skip the compaction
						const f2   dk  = Dd == Dc ? cur[k] : dry[k];
						const bool in0 = I || (pos0 >= p.sw_t_begin && pos0 < r1), in1 = I || (pos0 + 1 >= p.sw_t_begin && pos0 + 1 < r1);
						const float d0 = (I || pos0 >= p.sw_dry_from) ? dk.x : 0.f, d1 = (I || pos0 + 1 >= p.sw_dry_from) ? dk.y : 0.f;
						sw_raw         = fmaxf (sw_raw, fmaxf (in0 ? fabsf (dk.x) : 0.f, in1 ? fabsf (dk.y) : 0.f));
						const bool     l0 = in0 && d0 * d0 + hp.x * hp.x > sw_thr2, l1 = in1 && d1 * d1 + hp.y * hp.y > sw_thr2;
						const uint64_t m0 = __builtin_amdgcn_ballot_w64 (l0), m1 = __builtin_amdgcn_ballot_w64 (l1);
						if ((m0 | m1) != 0) {
							if (l0) {
								list[sw_cnt + __builtin_amdgcn_mbcnt_hi ((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo ((unsigned)m0, 0u))] = make_float2 (d0, hp.x);
							}
							sw_cnt += (unsigned)__builtin_popcountll (m0);
							if (l1) {
								list[sw_cnt + __builtin_amdgcn_mbcnt_hi ((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo ((unsigned)m1, 0u))] = make_float2 (d1, hp.y);
							}
							sw_cnt += (unsigned)__builtin_popcountll (m1);
						}
					}
				};
				if (full && s >= p.sw_t_begin && s >= p.sw_dry_from) {
					test_rows (std::true_type {});
				} else {
					test_rows (std::false_type {});
				}
			}
		} else if (it >= NP) {
			const bool full = s + BLK <= r1;
			if (MODE == MODE_ROT && !ACC && full && out_al8 && !(PROT_ABL2 & 64)) {
				const f2 ca2 = mkf2 (ca, ca), sa2 = mkf2 (sa, sa);
#pragma unroll
				for (int m = 0; m < HS / 2; ++m) {
					PROT_GLOBAL C* __restrict__ ob = reinterpret_cast<PROT_GLOBAL C*> (uniform_ptr (out + ((PROT_ABL2 & 128) ? r0 : s) + 1024 * m));
#pragma unroll
					for (int e = 0; e < 2; ++e) {
						ob[(unsigned)t + 256u * e] = to_sp (ca2 * dry[2 * m + e] + sa2 * u[2 * m + e]);
					}
				}
			} else if (MODE == MODE_ROT && ACC && full && out_al8 && !(PROT_ABL2 & 64)) {
				// a later partition of a long FIR (16384 / 32768 taps): its share added to what the earlier launches stored
				const f2 sa2 = mkf2 (sa, sa);
#pragma unroll
				for (int m = 0; m < HS / 2; ++m) {
					PROT_GLOBAL C* __restrict__ ob = reinterpret_cast<PROT_GLOBAL C*> (uniform_ptr (out + s + 1024 * m));
					f2                          prev[2];
#pragma unroll
					for (int e = 0; e < 2; ++e) {
						prev[e] = to_f2 (ob[(unsigned)t + 256u * e]);
					}
#pragma unroll
					for (int e = 0; e < 2; ++e) {
						ob[(unsigned)t + 256u * e] = to_sp (prev[e] + sa2 * u[2 * m + e]);
					}
				}
			} else if (MODE == MODE_HILBERT && ACC && full && out_al8 && !(PROT_ABL2 & 64)) {
				// ... and of the sweep's Hilbert branch (CLI blksiz 16384 / 32768: the second and later launches): the same
				// aligned-pair load-add-store (through the per-frame path below this launch took 3.8 ms of cfg4's 7.5 at
				// blksiz 16384, against 2.0 for the launch that only stores)
#pragma unroll
				for (int m = 0; m < HS / 2; ++m) {
					PROT_GLOBAL C* __restrict__ ob = reinterpret_cast<PROT_GLOBAL C*> (uniform_ptr (out + s + 1024 * m));
					f2                          prev[2];
#pragma unroll
					for (int e = 0; e < 2; ++e) {
						prev[e] = to_f2 (ob[(unsigned)t + 256u * e]);
					}
#pragma unroll
					for (int e = 0; e < 2; ++e) {
						ob[(unsigned)t + 256u * e] = to_sp (prev[e] + u[2 * m + e]);
					}
				}
			} else if (MODE == MODE_HILBERT && !ACC && full && out_al8 && !(PROT_ABL2 & 64)) {
				// the Hilbert pass of the sweep: whole blocks as aligned pairs too
#pragma unroll
				for (int m = 0; m < HS / 2; ++m) {
					PROT_GLOBAL C* __restrict__ ob = reinterpret_cast<PROT_GLOBAL C*> (uniform_ptr (out + s + 1024 * m));
#pragma unroll
					for (int e = 0; e < 2; ++e) {
						ob[(unsigned)t + 256u * e] = to_sp (u[2 * m + e]);
					}
				}
			} else {
#pragma unroll
				for (int k = 0; k < HS; ++k) {
#pragma unroll
					for (int h = 0; h < 2; ++h) {
						const int64_t pos = s + 2 * (int64_t)(t + 256 * k) + h;
						const float   uu  = h ? u[k].y : u[k].x;
						if ((PROT_ABL2 & 64) ? (pos < r1 && uu == 123.456f) : (pos < r1)) { // 64: (almost) no stores
							if (MODE == MODE_HILBERT) {
								out[pos] = ACC ? (S)((float)out[pos] + uu) : (S)uu;
							} else {
								float cc = ca, ss = sa;
								if (MODE == MODE_RAMP) {
									const f2 cs = reinterpret_cast<const f2*> (p.rot_s)[(int64_t)c * p.n_out + pos];
									cc          = cs.x;
									ss          = cs.y;
								}
								out[pos] = ACC ? (S)((float)out[pos] + ss * uu) : (S)(cc * (h ? dry[k].y : dry[k].x) + ss * uu);
							}
						}
					}
				}
			}
		}
		st.tick (10);
	}
#ifdef PROT_STAMPS
	if ((t & 63) == 0) {
#if PROT_STAMPS == 1
#pragma unroll
		for (int i = 0; i < 11; ++i) {
			atomicAdd (&g_stamps[i], (unsigned long long)st.acc[i]);
		}
#endif
#if PROT_STAMPS == 1
		atomicAdd (&g_stamps[16], (unsigned long long)n_it);
		atomicAdd (&g_stamps[17], (unsigned long long)(__builtin_readcyclecounter () - st_c0));
		atomicAdd (&g_stamps[18], (unsigned long long)(__builtin_amdgcn_s_memrealtime () - st_r0));
		atomicAdd (&g_stamps[19], 1ull);
#else // (12 000 atomics on one cache line are 100 us of launch time: the timeline build counts workgroups only)
		if (t == 0) {
			atomicAdd (&g_stamps[16], 4ull * (unsigned long long)n_it);
			atomicAdd (&g_stamps[19], 4ull);
		}
#endif
	}
	if (t == 0 && blockIdx.x < 4096) {
		g_wg_t[4 * blockIdx.x]     = st_entry;
		g_wg_t[4 * blockIdx.x + 1] = (unsigned long long)(__builtin_readcyclecounter () - st_c0); // shader cycles, loop start to end
		g_wg_t[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime ();
		g_wg_t[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg ((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg ((3 << 11) | 20) << 32);
	}
#endif
	hist_write_slice (p, blockIdx.y, threadIdx.x);
	if constexpr (MODE == MODE_SWEEP) {
		if ((t & 63) == 0) {
			p.surv_cnt[((int64_t)c * gridDim.x + blockIdx.x) * 4 + (t >> 6)] = sw_cnt;
		}
		if (p.sw_raw_slot >= 0) {
			for (int m = 1; m < 64; m <<= 1) {
				sw_raw = fmaxf (sw_raw, __shfl_xor (sw_raw, m));
			}
			if ((t & 63) == 0 && sw_raw > 0.f) { // (non-negative floats order as their bit patterns; look before the atomic)
				unsigned* const slot = reinterpret_cast<unsigned*> (p.sw_peaks + (int64_t)c * kMaxSample + p.sw_raw_slot);
				if (__float_as_uint (sw_raw) > __hip_atomic_load (slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
					atomicMax (slot, __float_as_uint (sw_raw));
				}
			}
		}
	}
}

template <class T, int NP, int HS, int MODE>
static hipError_t launch_poly_m (const OlaParamsT<typename T::samp>& p, dim3 grid, hipStream_t s)
{
	// dry frames from registers when (dry_delay - conv_delay) is a whole number of 512-frame sample rows
	// that the hop geometry in use actually meets: LV2 48 kHz 3 (HS 10), 96 kHz / CLI 4096 4, CLI 8192 /
	// LV2 192 kHz 8 (HS 8), CLI 2048 2 (HS 12), CLI 1024 1 (HS 14); anything else re-reads the stream
	const int  diff = p.dry_delay - p.conv_delay;
	const int  dd   = (diff > 0 && diff % 512 == 0) ? diff / 512 : 0;
	constexpr int DHS = HS == 8 ? (NP == 2 ? 8 : 4) : HS == 10 ? 3 : HS == 12 ? 2 : 1;
	// fp32, one partition: the register diet of the 3-workgroups-per-CU form (3 x 53 KiB of LDS = one CU)
	// (the per-frame rotation of the ramp form with 12 or 14 rows per block does not fit 168 VGPRs: 2 per CU there)
	// fp16 samples: the host sizes its one-round launches for FOUR resident workgroups per CU (one partition) or THREE (two
	// partitions) -- the launch bounds say the same, so that a register count beyond that budget is a compile-time spill
	// report (tools/kernel_resources.py) instead of a silently two-round launch
	constexpr int WPS = sizeof (typename T::samp) == 2 ? (NP == 1 ? 4 : 3)
	                  : (NP == 1 && !(MODE == MODE_RAMP && HS >= 12)) ? 3 : 2;
	const dim3 block (256);
	if constexpr (sizeof (typename T::samp) == 4 && NP == 2 && HS == 8 && MODE == MODE_HILBERT) {
		if (p.surv != nullptr) { // the last launch of a long FIR's Hilbert branch with the sweep's test (lists instead of stores)
			if (!p.accumulate) {
				return hipErrorInvalidValue;
			}
			hipLaunchKernelGGL ((olapoly_kernel<T, NP, HS, MODE_SWEEP, 0, true, WPS>), grid, block, 0, s, p);
			return hipGetLastError ();
		}
	}
	if constexpr (sizeof (typename T::samp) == 4 && NP == 1 && HS == 14 && MODE == MODE_HILBERT) {
		if (p.surv != nullptr) { // CLI blksiz 1024: the one launch of its Hilbert branch, testing instead of storing
			if (p.accumulate) {
				return hipErrorInvalidValue;
			}
			hipLaunchKernelGGL ((olapoly_kernel<T, NP, HS, MODE_SWEEP, 0, false, 2, (WPS >= 3)>), grid, block, 0, s, p);
			return hipGetLastError ();
		}
	}
	if (p.surv != nullptr) {
		return hipErrorInvalidValue;
	}
	if (p.accumulate) {
		hipLaunchKernelGGL ((olapoly_kernel<T, NP, HS, MODE, 0, true, WPS>), grid, block, 0, s, p);
	} else if (MODE != MODE_HILBERT && dd == DHS) {
		hipLaunchKernelGGL ((olapoly_kernel<T, NP, HS, MODE, DHS, false, WPS>), grid, block, 0, s, p);
	} else {
		// (dry frames re-read from the stream: with 10 or more rows per block that form does not fit 168 VGPRs)
		constexpr int WPS0 = (HS >= 10 && MODE != MODE_HILBERT) ? 2 : WPS;
		hipLaunchKernelGGL ((olapoly_kernel<T, NP, HS, MODE, 0, false, WPS0>), grid, block, 0, s, p);
	}
	return hipGetLastError ();
}

template <class T, int NP, int HS>
static hipError_t launch_poly_n (const OlaParamsT<typename T::samp>& p, int n_chn, hipStream_t s)
{
	if (p.n_out <= 0) {
		return hipSuccess;
	}
	const int64_t unit = p.run_stride > 0 ? p.run_stride : p.run_frames; // (run_stride: every k-th block only, the fused sweep's sample pass)
	const int64_t runs = (p.n_out + unit - 1) / unit;
	dim3          grid ((unsigned)runs, (unsigned)n_chn);
	if (p.hilbert_only) {
		return launch_poly_m<T, NP, HS, MODE_HILBERT> (p, grid, s);
	}
	if (p.rot_s != nullptr) {
		return launch_poly_m<T, NP, HS, MODE_RAMP> (p, grid, s);
	}
	return launch_poly_m<T, NP, HS, MODE_ROT> (p, grid, s);
}

hipError_t launch_ola_poly (const OlaParams& p, int n_part, int hop_rows, int n_chn, hipStream_t s)
{
	if ((p.conv_delay & 1) || (p.run_frames % (512 * hop_rows)) != 0 || (n_part != 1 && n_part != 2) ||
	    (hop_rows != 8 && (n_part != 1 || (hop_rows != 10 && hop_rows != 12 && hop_rows != 14)))) {
		return hipErrorInvalidValue;
	}
	if (n_part == 2) {
		return launch_poly_n<T32, 2, 8> (p, n_chn, s);
	}
	switch (hop_rows) {
		case 8: return launch_poly_n<T32, 1, 8> (p, n_chn, s);
		case 10: return launch_poly_n<T32, 1, 10> (p, n_chn, s);
		case 12: return launch_poly_n<T32, 1, 12> (p, n_chn, s);
		default: return launch_poly_n<T32, 1, 14> (p, n_chn, s);
	}
}

// fp16 sample path in polyphase form (prot_apply_dev_f16 / _ramp): one or two partitions of g per launch, steady or
// per-frame rotation, with or without history
hipError_t launch_ola_poly_f16 (const OlaParamsH& p, int n_part, int hop_rows, int n_chn, hipStream_t s)
{
	if ((p.conv_delay & 1) || (p.run_frames % (512 * hop_rows)) != 0 || (p.accumulate && n_part != 2) || (n_part != 1 && n_part != 2) ||
	    (hop_rows != 8 && (n_part != 1 || (hop_rows != 10 && hop_rows != 12 && hop_rows != 14)))) {
		return hipErrorInvalidValue;
	}
	if (n_part == 2) {
		return launch_poly_n<T16, 2, 8> (p, n_chn, s);
	}
	switch (hop_rows) {
		case 8: return launch_poly_n<T16, 1, 8> (p, n_chn, s);
		case 10: return launch_poly_n<T16, 1, 10> (p, n_chn, s);
		case 12: return launch_poly_n<T16, 1, 12> (p, n_chn, s);
		default: return launch_poly_n<T16, 1, 14> (p, n_chn, s);
	}
}

} // namespace prot
