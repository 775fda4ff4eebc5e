// prot_fft4096.h -- the 4096-point transform pair shared by the overlap-add kernels (ola4096p2_kernel: prot_kernels.hip,
// olapoly_kernel: prot_olapoly.hip): precision traits, exchange geometry, the forward / inverse transforms with their
// wave-local inner exchanges.  Device code only; included by the two units above.
#pragma once
#include <type_traits>
#include "prot_fft.h"
#include "prot_kernels.h"
#include "prot_dev.h"
#include "prot_ablate.h"

namespace prot {

// (kL1 = 272, the exchange stride between pass 1 and 2, and the MODE_* enum live in prot_dev.h)
constexpr int kL2 = 258; // forward exchange 2 and inverse exchange A share slots (n0*kL2 + t):
                         // conflict-free b64 reads on the inverse side, 2-way on the forward
                         // side stores (8 vs 6 cycles) -- cheaper than the barrier it removes
constexpr int kL3 = kL2;
constexpr int kLdsRegion = 16 * kL1;

// exchange accesses are volatile so the backend keeps one ds_read_b64 / ds_write_b64 per element instead of fusing
// neighbours into ds_read2_b64 / ds_write2_b64 (8 and 13 LDS cycles against 2 x 2 and 2 x 6, and mod-32 banking)
#define PROT_LDSQ volatile
// precision traits: fp32 (aligned VGPR pair per complex) and fp16 (one VGPR per complex)
struct T32 {
	typedef pk32  ops;
	typedef f2    cplx;
	typedef float samp;
	static constexpr int kWaves = 2; // workgroups per CU the register budget is sized for
};
struct T16 {
	typedef pk16     ops;
	typedef h2       cplx;
	typedef _Float16 samp;
	static constexpr int kWaves = 3;
};

// ---- FFT-4096 with wave-local inner exchanges (2 workgroup barriers per transform pair) ----
// Pass-2 threads are (n1 = tl, k1 = th), so the 16 threads of one `th` hold a full n1 x k2 tile:
// forward exchange 2 (and its inverse mirror) is a 16 x 16 transposition INSIDE that 16-lane
// group.  It runs in the group's own 272-slot chunk of the pass-1 exchange buffer (the slots
// th*kL1 + [0, 272) that only this group read after the barrier), row stride 17 => conflict-free
// ds_write_b64 (16-lane groups, mod-16 slots) and ds_read_b64 (32-lane halves, mod-32 slots).
// DS instructions of one wave execute in order, so no s_barrier is needed around them; the final
// thread map is t = k2 + 16 k1, i.e. thread t holds bins  th + 16 tl + 256 k  (H is stored
// digit-swapped to keep those reads linear).
//
// Scheduling (the kernel is latency-bound at 2 waves per SIMD, not throughput-bound):
//  * all LDS accesses are volatile, address-space-3: one ds_read_b64/ds_write_b64 per element in
//    source order (the backend otherwise fuses neighbours into ds_read2_b64/ds_write2_b64: 8 and
//    13 LDS cycles against 2 x 2 and 2 x 6, with mod-32 banking);
//  * table reads (twm, H) are issued a whole butterfly ahead of their use and pinned there with
//    sched_barrier -- left alone, the register-pressure-driven scheduler sinks each ds_read next
//    to its multiply and every one of them stalls for the full LDS latency;
//  * every value goes to the exchange as soon as its twiddle multiply is done, so the
//    VGPR->LDS transfers (6 cycles each) run under the remaining multiplies.
// (kG = 17, wave_sync and PROT_LDS_PTR live in prot_dev.h)

// slot of bin i (0..2048) in the digit-swapped spectrum table
PROT_DEV int swap_lo_digits (int i)
{
	return (i & ~255) | ((i & 15) << 4) | ((i >> 4) & 15);
}


// issue the reads of W256^{k tl}, k = 1..15 (twm[16 k + tl]: neighbouring lanes, conflict-free)
template <class T>
PROT_DEV void twm_fetch (typename T::cplx (&q)[16], const typename T::cplx* twm, int tl)
{
	typedef PROT_LDS_PTR (T) VC;
	const VC* r = (const VC*)&twm[tl];
#pragma unroll
	for (int k = 1; k < 16; ++k) {
		q[k] = r[16 * k];
	}
}

// v[k] *= w[k] (k = 1..15; conj(w) for DIR < 0) and store v[k] to dst[k * STRIDE] right away;
// the multiplies go in pairs so that a lone wave still has an independent instruction to issue
template <class T, int DIR, int STRIDE>
PROT_DEV void twiddle_store (typename T::cplx (&v)[16], const typename T::cplx (&w)[16], PROT_LDS_PTR (T) * dst)
{
	typedef typename T::ops O;
	if (!(PROT_ABL2 & 1)) dst[0] = v[0];
	if (!(PROT_ABL2 & 8)) v[1] = O::template cmul<DIR> (v[1], w[1]);
	if (!(PROT_ABL2 & 1)) dst[STRIDE] = v[1];
#pragma unroll
	for (int k = 2; k < 16; k += 2) {
		if (!(PROT_ABL2 & 8)) O::template cmul2<DIR> (v[k], w[k], v[k + 1], w[k + 1]);
		if (!(PROT_ABL2 & 1)) dst[k * STRIDE] = v[k];
		if (!(PROT_ABL2 & 1)) dst[(k + 1) * STRIDE] = v[k + 1];
	}
}

// the same with the W256 twiddles fetched from LDS four at a time, one batch ahead of its multiplies (QB = 4:
// 16 VGPRs of twiddles in flight instead of 30) -- the register diet of the 3-workgroups-per-CU instantiations.
// `r` = &twm[tl]; q0 = batch 0 (k = 1..3), fetched by the caller ahead of the butterfly
template <class T, int DIR, int STRIDE>
PROT_DEV void twiddle_store_lean (typename T::cplx (&v)[16], const PROT_LDS_PTR (T) * r, const typename T::cplx (&q0)[4],
                                  PROT_LDS_PTR (T) * dst)
{
	typedef typename T::ops  O;
	typedef typename T::cplx C;
	C q[2][4];
#pragma unroll
	for (int i = 1; i < 4; ++i) {
		q[0][i] = q0[i];
	}
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		if (j < 3) {
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				if (!(PROT_ABL2 & 8)) q[(j + 1) & 1][i] = r[16 * (4 * (j + 1) + i)];
			}
		}
		if (j == 0) {
			if (!(PROT_ABL2 & 1)) dst[0] = v[0];
			if (!(PROT_ABL2 & 8)) v[1] = O::template cmul<DIR> (v[1], q[0][1]);
			if (!(PROT_ABL2 & 1)) dst[STRIDE] = v[1];
			if (!(PROT_ABL2 & 8)) O::template cmul2<DIR> (v[2], q[0][2], v[3], q[0][3]);
			if (!(PROT_ABL2 & 1)) dst[2 * STRIDE] = v[2];
			if (!(PROT_ABL2 & 1)) dst[3 * STRIDE] = v[3];
		} else {
#pragma unroll
			for (int i = 0; i < 4; i += 2) {
				const int k = 4 * j + i;
				if (!(PROT_ABL2 & 8)) O::template cmul2<DIR> (v[k], q[j & 1][i], v[k + 1], q[j & 1][i + 1]);
				if (!(PROT_ABL2 & 1)) dst[k * STRIDE] = v[k];
				if (!(PROT_ABL2 & 1)) dst[(k + 1) * STRIDE] = v[k + 1];
			}
		}
	}
}

// radix-16 + W256 twiddles (table in LDS, `r` = &twm[tl]) + store to dst[k * STRIDE], the second stage one DFT-4 at a
// time: the four results of DFT-4 number i (outputs i, i + 4, i + 8, i + 12) are multiplied and stored while the next one
// computes; the table entries of group i + 1 are fetched ahead of group i's multiplies (two groups = 16 VGPRs in flight,
// as twiddle_store_lean).  The 3-per-CU instantiations' form of pkt_fft16 + twiddle_store_lean.
template <class T, int DIR, int STRIDE>
PROT_DEV void fft16_twiddle_store_each (typename T::cplx (&v)[16], const PROT_LDS_PTR (T) * r, PROT_LDS_PTR (T) * dst)
{
	typedef typename T::ops  O;
	typedef typename T::cplx C;
	C q[2][4];
#pragma unroll
	for (int j = 1; j < 4; ++j) {
		q[0][j] = r[16 * (4 * j)];
	}
	pkt_fft16_stage1<O, DIR> (v);
	pkt_fft16_stage2_each<O, DIR> (v, [&] (auto ic, C& x0, C& x1, C& x2, C& x3) {
		constexpr int i = decltype (ic)::value;
		if (i < 3) {
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				q[(i + 1) & 1][j] = r[16 * (i + 1 + 4 * j)];
			}
		}
		if (i == 0) {
			x1 = O::template cmul<DIR> (x1, q[0][1]);
		} else {
			O::template cmul2<DIR> (x0, q[i & 1][0], x1, q[i & 1][1]);
		}
		O::template cmul2<DIR> (x2, q[i & 1][2], x3, q[i & 1][3]);
		dst[i * STRIDE]        = x0;
		dst[(i + 4) * STRIDE]  = x1;
		dst[(i + 8) * STRIDE]  = x2;
		dst[(i + 12) * STRIDE] = x3;
		__builtin_amdgcn_sched_barrier (0);
	});
}

// forward transform up to (not including) the last radix-16 pass; `before_last` runs between the
// issue of the last exchange reads and that pass (the caller fetches its first table entries there)
template <class T, int NNZ, bool LEAN, class F, class G>
PROT_DEV void fwd4096w (typename T::cplx (&v)[16], typename T::cplx* lds, const typename T::cplx* twm,
                        const typename T::cplx (&tw1)[16], int t, int tl, int th, F&& before_last, Stamps& st, G&& after_barrier)
{
	typedef typename T::cplx C;
	typedef PROT_LDS_PTR (T) VC;
	typedef typename T::ops  O;
	// own slots (k*kL1 + t) -- read by this thread at the end of the previous inverse: no barrier in front
	if (LEAN) { // split outer twiddles: tw1[1..3] = W^{t i}, tw1[5..7] = W^{4 t i}
		const C (&A)[4] = reinterpret_cast<const C (&)[4]> (tw1[0]);
		const C (&B)[4] = reinterpret_cast<const C (&)[4]> (tw1[4]);
		VC* w = (VC*)&lds[t];
		if (!(PROT_ABL2 & 4)) pkt_fft16_fwd_pruned_twout<O, NNZ> (v, A, B);
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			if (!(PROT_ABL2 & 1)) w[k * kL1] = v[k];
		}
	} else {
		if (!(PROT_ABL2 & 4)) pkt_fft16_fwd_pruned<O, NNZ> (v);
		twiddle_store<T, 1, kL1> (v, tw1, (VC*)&lds[t]);
	}
	st.tick (1);
	if (!(PROT_ABL2 & 2)) __syncthreads ();
	st.tick (2);
	{
		const VC* r = (const VC*)&lds[th * kL1 + tl];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			if (!(PROT_ABL2 & 1)) v[k] = r[16 * k];
		}
	}
	after_barrier ();
	if (LEAN) {
		const VC* rq = (const VC*)&twm[tl];
		C q0[4];
#pragma unroll
		for (int i = 1; i < 4; ++i) {
			if (!(PROT_ABL2 & 8)) q0[i] = rq[16 * i];
		}
		if (!(PROT_ABL2 & 4)) pkt_fft16<O, 1> (v);
		twiddle_store_lean<T, 1, 1> (v, rq, q0, (VC*)&lds[th * kL1 + tl * kG]);
	} else {
		C q[16];
		twm_fetch<T> (q, twm, tl);
		__builtin_amdgcn_sched_barrier (0);
		if (!(PROT_ABL2 & 4)) pkt_fft16<O, 1> (v);
		twiddle_store<T, 1, 1> (v, q, (VC*)&lds[th * kL1 + tl * kG]);
	}
	wave_sync ();
	{
		const VC* r = (const VC*)&lds[th * kL1 + tl];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			if (!(PROT_ABL2 & 1)) v[k] = r[k * kG];
		}
	}
	st.tick (3);
	before_last ();
	if (!LEAN) {
		__builtin_amdgcn_sched_barrier (0);
	}
	if (!(PROT_ABL2 & 4)) pkt_fft16<O, 1> (v);
	st.tick (4);
}
template <class T, int NNZ, class F>
PROT_DEV void fwd4096w (typename T::cplx (&v)[16], typename T::cplx* lds, const typename T::cplx* twm,
                        const typename T::cplx (&tw1)[16], int t, int tl, int th, F&& before_last)
{
	Stamps st;
	fwd4096w<T, NNZ, false> (v, lds, twm, tw1, t, tl, th, before_last, st, [] () {});
}

template <class T, bool LEAN, class F>
PROT_DEV void inv4096w (typename T::cplx (&v)[16], typename T::cplx* lds, const typename T::cplx* twm,
                        const typename T::cplx (&tw1)[16], int t, int tl, int th, Stamps& st, F&& before_last)
{
	typedef typename T::cplx C;
	typedef PROT_LDS_PTR (T) VC;
	typedef typename T::ops  O;
	if (LEAN) {
		const VC* rq = (const VC*)&twm[tl];
		C q0[4];
#pragma unroll
		for (int i = 1; i < 4; ++i) {
			if (!(PROT_ABL2 & 8)) q0[i] = rq[16 * i];
		}
		if (!(PROT_ABL2 & 4)) pkt_fft16<O, -1> (v);
		twiddle_store_lean<T, -1, kG> (v, rq, q0, (VC*)&lds[th * kL1 + tl]);
	} else {
		C q[16];
		twm_fetch<T> (q, twm, tl); // W256^{k2 n1}: the table is symmetric, so lanes read neighbours here too
		__builtin_amdgcn_sched_barrier (0);
		if (!(PROT_ABL2 & 4)) pkt_fft16<O, -1> (v);
		twiddle_store<T, -1, kG> (v, q, (VC*)&lds[th * kL1 + tl]);
	}
	wave_sync ();
	{
		const VC* r = (const VC*)&lds[th * kL1 + tl * kG];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			if (!(PROT_ABL2 & 1)) v[k] = r[k];
		}
	}
	if (!(PROT_ABL2 & 4)) pkt_fft16<O, -1> (v);
	{
		// the group's chunk again, in the pass-1 layout: nobody else touches it before the barrier
		wave_sync ();
		VC* w = (VC*)&lds[th * kL1 + tl];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			if (!(PROT_ABL2 & 1)) w[16 * k] = v[k];
		}
		st.tick (7);
		if (!(PROT_ABL2 & 2)) __syncthreads ();
		st.tick (8);
		const VC* r = (const VC*)&lds[t];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			if (!(PROT_ABL2 & 1)) v[k] = r[k * kL1];
		}
	}
	before_last ();
	if (LEAN) {
		const C (&A)[4] = reinterpret_cast<const C (&)[4]> (tw1[0]);
		const C (&B)[4] = reinterpret_cast<const C (&)[4]> (tw1[4]);
		if (!(PROT_ABL2 & 4)) pkt_fft16_inv_twin<O> (v, A, B);
	} else {
		if (!(PROT_ABL2 & 8)) {
			v[1] = O::template cmul<-1> (v[1], tw1[1]);
#pragma unroll
			for (int k = 2; k < 16; k += 2) {
				O::template cmul2<-1> (v[k], tw1[k], v[k + 1], tw1[k + 1]);
			}
		}
		if (!(PROT_ABL2 & 4)) pkt_fft16<O, -1> (v);
	}
	st.tick (9);
}
template <class T>
PROT_DEV void inv4096w (typename T::cplx (&v)[16], typename T::cplx* lds, const typename T::cplx* twm,
                        const typename T::cplx (&tw1)[16], int t, int tl, int th)
{
	Stamps st;
	inv4096w<T, false> (v, lds, twm, tw1, t, tl, th, st, [] () {});
}

} // namespace prot
