// prot_kernels.hip -- gfx950 (MI355X, CDNA4) overlap-add kernels of the phase-rotation hot path, and the small
// movers around them.  wave64 throughout; no hipFFT/rocFFT (the OLA path is a butterfly network, not a contraction).
//
// The kernels of the path, by file:
//   prot_olasave.hip  olasave_kernel   -- polyphase OVERLAP-SAVE, one partition of the odd taps, no spectrum table,
//                                         generated gfx950 assembly, four workgroups per CU: the default for every
//                                         Hilbert FIR of 1024..6144 taps (LV2 48 / 96 kHz, CLI blksiz 1024..4096) --
//                                         the headline kernel
//   here              olapoly_kernel   -- polyphase overlap-add with a spectrum table: FIRs of 8192 taps and more
//                                         (two partitions of the odd taps per launch, accumulating launches beyond),
//                                         7168-frame blocks (CLI blksiz 1024: HS = 14), PROT_CFG_OVERLAP_ADD, fp16
//   here              ola4096p2_kernel -- direct form, two 2048-tap partitions sharing a forward transform:
//                                         PROT_CFG_DIRECT_FORM (cross-check of the polyphase forms), fp16 at 4096 taps
//   here              ola4096_kernel   -- direct form, one partition: FIRs below 1024 taps (CLI blksiz 64..512),
//                                         PROT_CFG_SINGLE_FFT_PER_HOP; described below
//   prot_sweep.hip    sweep_kernel     -- K7, the per-angle abs-max on the matrix core over compacted survivors
// They share the 4096-point transform pair (fwd4096w / inv4096w, or the inline version of ola4096_kernel), the
// register-resident overlap-add tail and dry path, the halo scheme and the fused history write (prot_dev.h).
//
// ola4096_kernel -- overlap-add Hilbert convolution + rotation, one
//   workgroup (256 threads = 4 waves) per run of consecutive hops of one
//   channel-stream.  Replaces the block body of process_channel
//   (src/phaserotate.c:629-720) and PhaseRotateProc::hilbert/rotate
//   (cli/phase-rotate.cc:181-225).
//     * two consecutive hops A, B ride in one complex transform (re = A,
//       im = B): forward FFT-4096 -> x H (real FIR => acts on re and im
//       independently) -> inverse FFT-4096 gives both convolutions, no
//       real/complex split pass;
//     * FFT-4096 = 16 x 16 x 16: three radix-16 passes in registers (16
//       complex per thread), two digit transpositions through LDS
//       (34 KiB, strides 272/257/258 chosen conflict-free for ds_write_b64 /
//       ds_read_b64);  the inverse runs the mirror image so the spectrum
//       stays in registers across "x H";
//     * thread t owns frames t + 256k of every block, so HBM loads/stores are
//       coalesced dwords of contiguous frames and the overlap-add tail
//       (fftlen - hop frames) lives in registers across the run -- LDS holds
//       nothing but the FFT exchange;
//     * a run starts (fftlen/hop) halo blocks early to rebuild the tail from
//       input history, so workgroups never exchange data.
#include <type_traits>
#include "prot_fft.h"
#include "prot_kernels.h"
#include "prot_dev.h"
#include "prot_ablate.h"

#include <algorithm>
#include <cstdlib>


namespace prot {

// ------------------------------------------------------------ OLA, FFT-4096
// (kL1 = 272, the exchange stride between pass 1 and 2, and the MODE_* enum live in prot_dev.h)
constexpr int kL2 = 258; // forward exchange 2 and inverse exchange A share slots (n0*kL2 + t):
                         // conflict-free b64 reads on the inverse side, 2-way on the forward
                         // side stores (8 vs 6 cycles) -- cheaper than the barrier it removes
constexpr int kL3 = kL2;
constexpr int kLdsRegion = 16 * kL1;


// the 2*HOPSEG frames thread t contributes to the hop pair whose block A reads in[iA ...]
template <int HOPSEG>
PROT_DEV void load_pair (float (&nx)[2 * HOPSEG], const float* __restrict__ in, int64_t n_in,
                         const float* __restrict__ hist, int hist_len, int64_t iA, int t)
{
	constexpr int HOP  = HOPSEG * 256;
	const bool    fast = (iA >= 0) && (iA + 2 * HOP <= n_in);
	if (fast) {
		const float* __restrict__ q = in + iA + t;
#pragma unroll
		for (int k = 0; k < 2 * HOPSEG; ++k) {
			nx[k] = q[256 * k];
		}
	} else {
#pragma unroll
		for (int k = 0; k < 2 * HOPSEG; ++k) {
			nx[k] = ldv (in, n_in, hist, hist_len, iA + t + 256 * k);
		}
	}
}

template <int HOPSEG, int NREG, int WAVES, int MODE, bool ACC = false, int DH = 0, bool PK = true>
__global__ __launch_bounds__ (256, WAVES) void ola4096_kernel (const OlaParams p)
{
	constexpr int HOP  = HOPSEG * 256;
	constexpr int NS   = 16 / HOPSEG;        // hop-sized segments per FFT
	constexpr int NH   = NS - 1;             // halo blocks that reach into a run
	constexpr int NACC = (NS - 1) * HOPSEG;  // OLA tail registers

	__shared__ f2 lds[NREG][kLdsRegion];
	__shared__ f2 twm[256];              // twm[k1*16 + n0] = W256^{n0 k1}
	// FIR spectrum, bins 0..2048 (the FIR is real: H[4096-k] = conj H[k]).  Kept in LDS
	// (16 KiB; 3 workgroups still fit a CU) because 32 KiB of L2 reads per hop pair from
	// every workgroup was the largest single stall source (ablation: -28 % time at hop 2048)
	__shared__ f2 Hs[2049];

	const int     t  = threadIdx.x;
	const int     c  = blockIdx.y;
	const int64_t r0 = (int64_t)blockIdx.x * p.run_frames;
	if (r0 >= p.n_out) {
		hist_write_slice (p, blockIdx.y, threadIdx.x);
		return;
	}
	const int64_t r1 = (r0 + p.run_frames < p.n_out) ? r0 + p.run_frames : p.n_out;

	const float* __restrict__ in   = p.in + (int64_t)c * p.in_stride;
	const float* __restrict__ hist = p.hist ? p.hist + (int64_t)c * p.hist_len : nullptr;
	float* __restrict__ out        = p.out + (int64_t)c * p.out_stride;
	const int64_t n_in             = p.n_in;
	const int     Dc = p.conv_delay, Dd = p.dry_delay;

	// pass-1 twiddles W4096^{t k0}, k0 = 1..15, live in registers for the run
	const f2* __restrict__ twp = reinterpret_cast<const f2*> (p.tw);
	f2 tw1[16];
#pragma unroll
	for (int k = 1; k < 16; ++k) {
		tw1[k] = twp[(t * k) & 4095];
	}
	twm[t] = twp[16 * (((t & 15) * (t >> 4)) & 255)];
	for (int i = t; i <= 2048; i += 256) {
		Hs[i] = reinterpret_cast<const f2*> (p.H)[i];
	}

	float ca = 0.f, sa = 1.f;
	if (MODE == MODE_ROT) {
		ca = p.rot ? p.rot[2 * c] : p.rot_v[2 * c];
		sa = p.rot ? p.rot[2 * c + 1] : p.rot_v[2 * c + 1];
	}

	float acc[NACC];
#pragma unroll
	for (int i = 0; i < NACC; ++i) {
		acc[i] = 0.f;
	}

	const int tl = t & 15, th = t >> 4;
	const int n_pairs = (NH + 1) / 2 + (int)((r1 - r0 + 2 * HOP - 1) / (2 * HOP));
	int64_t   sA      = r0 - (int64_t)(NH + 1) * HOP;

	// software pipeline: the frames of pair it+1 are requested while pair it is
	// in its inverse transform, so HBM latency hides behind a whole half-iteration
	float nx[2 * HOPSEG];
	load_pair<HOPSEG> (nx, in, n_in, hist, p.hist_len, sA - Dc, t);
	__syncthreads ();

	// DH > 0: the dry path (x delayed by dry_delay) is served from registers: its
	// frames are the FFT input of this pair (cur) and the last DH 256-frame segments of
	// the previous pair (dh) -- (dry_delay - conv_delay) = 256 * DH.  No second read of
	// the stream, so HBM traffic stays at the algorithmic 4 B in + 4 B out per frame.
	constexpr int NDH = DH > 0 ? DH : 1;
	float         dh[NDH];
#pragma unroll
	for (int i = 0; i < NDH; ++i) {
		dh[i] = 0.f;
	}
	if (DH > 0) { // frames in front of the first (halo) pair: never emitted, but keep them exact
#pragma unroll
		for (int i = 0; i < NDH; ++i) {
			dh[i] = ldv (in, n_in, hist, p.hist_len, sA - Dc - 256 * (DH - i) + t);
		}
	}

	const int t0 = t;
	for (int it = 0; it < n_pairs; ++it, sA += 2 * HOP) {
		// the thread id, opaque per iteration (see olapoly_kernel: keeps slow-path pointers out of loop-long registers)
		int t = t0;
		asm volatile ("" : "+v"(t));
		const int tl = t & 15, th = t >> 4;
		f2 v[16];
		float  cur[2 * HOPSEG];
#pragma unroll
		for (int k = 0; k < 2 * HOPSEG; ++k) {
			cur[k] = nx[k];
		}
#pragma unroll
		for (int k = 0; k < HOPSEG; ++k) {
			v[k] = mkf2 (it == 0 ? 0.f : cur[k], cur[HOPSEG + k]); // it == 0: alignment block, contributes nothing
		}
#pragma unroll
		for (int k = HOPSEG; k < 16; ++k) {
			v[k] = mkf2 (0.f, 0.f); // zero padding (pruned by the compiler)
		}

		// ---- forward pass 1 (over n2), twiddle W^{t k0} ---------------------
		if (PK) {
			pk_fft16_fwd_pruned<HOPSEG> (v);
		} else {
			fft16<1> (v);
		}
#pragma unroll
		for (int k = 1; k < 16; ++k) {
			v[k] = PK ? pk_cmul<1> (v[k], tw1[k]) : cmulw<1> (v[k], tw1[k]);
		}
		{
			// no barrier in front of these stores: every thread overwrites exactly the
			// slots it read itself at the end of the previous pair (k*kL1 + t)
			f2* w = &lds[NREG - 1][t];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				w[k * kL1] = v[k];
			}
			__syncthreads ();
			const f2* r = &lds[NREG - 1][th * kL1 + tl];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				v[k] = r[16 * k];
			}
		}
		// ---- forward pass 2 (over n1), twiddle W256^{n0 k1} -----------------
		if (PK) {
			pk_fft16<1> (v);
		} else {
			fft16<1> (v);
		}
#pragma unroll
		for (int k = 1; k < 16; ++k) {
			v[k] = PK ? pk_cmul<1> (v[k], twm[k * 16 + tl]) : cmulw<1> (v[k], twm[k * 16 + tl]);
		}
		if (NREG == 1) {
			__syncthreads ();
		}
		{
			f2* w = &lds[0][tl * kL2 + th];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				w[16 * k] = v[k];
			}
			__syncthreads ();
			const f2* r = &lds[0][t];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				v[k] = r[k * kL2];
			}
		}
		// ---- forward pass 3 (over n0): v[k2] = X[t + 256 k2] ----------------
		if (PK) {
			pk_fft16<1> (v);
		} else {
			fft16<1> (v);
		}

		// ---- x H: Hilbert FIR spectrum, 1/fftlen folded in -------------------
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			v[k] = PK ? pk_cmul<1> (v[k], Hs[t + 256 * k]) : cmulw<1> (v[k], Hs[t + 256 * k]);
		}
#pragma unroll
		for (int k = 8; k < 16; ++k) { // bins >= 2048: conjugate of the mirrored bin
			v[k] = PK ? pk_cmul<-1> (v[k], Hs[256 * (16 - k) - t]) : cmulw<-1> (v[k], Hs[256 * (16 - k) - t]);
		}

		// ---- request the next pair's frames (after the H loads: vmcnt retires in order)
		__builtin_amdgcn_sched_barrier (0);
		if (it + 1 < n_pairs) {
			load_pair<HOPSEG> (nx, in, n_in, hist, p.hist_len, sA + 2 * HOP - Dc, t);
		}
		// ... and the delayed dry frames of this pair (L2 hits: read as FFT input earlier)
		float      dry[2 * HOPSEG];
		const bool emit     = sA >= r0; // uniform: halo blocks emit nothing
		const bool fast_out = (sA - Dd >= 0) && (sA + 2 * HOP <= n_in + Dd) && (sA + 2 * HOP <= r1);
		if (DH > 0) {
#pragma unroll
			for (int k = 0; k < 2 * HOPSEG; ++k) {
				dry[k] = (k - DH >= 0) ? cur[k - DH >= 0 ? k - DH : 0] : dh[k - DH < 0 ? NDH + (k - DH) : 0];
			}
#pragma unroll
			for (int i = 0; i < NDH; ++i) {
				dh[i] = cur[2 * HOPSEG - NDH + i];
			}
		} else if (MODE != MODE_HILBERT && !ACC && emit) {
			if (fast_out) {
				const float* __restrict__ dq = in + (sA - Dd) + t;
#pragma unroll
				for (int k = 0; k < 2 * HOPSEG; ++k) {
					dry[k] = dq[256 * k];
				}
			} else {
#pragma unroll
				for (int k = 0; k < 2 * HOPSEG; ++k) {
					dry[k] = ldv (in, n_in, hist, p.hist_len, sA + t + 256 * k - Dd);
				}
			}
		}
		__builtin_amdgcn_sched_barrier (0);

		// ---- inverse pass 1 (over k2), twiddle conj W256^{n0 k1} ------------
		if (PK) {
			pk_fft16<-1> (v);
		} else {
			fft16<-1> (v);
		}
#pragma unroll
		for (int k = 1; k < 16; ++k) {
			v[k] = PK ? pk_cmul<-1> (v[k], twm[th * 16 + k]) : cmulw<-1> (v[k], twm[th * 16 + k]);
		}
		{
			// no barrier: same slots this thread read in forward exchange 2
			f2* w = &lds[0][t];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				w[k * kL3] = v[k];
			}
			__syncthreads ();
			const f2* r = &lds[0][tl * kL3 + th];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				v[k] = r[16 * k];
			}
		}
		// ---- inverse pass 2 (over k1) ----------------------------------------
		if (PK) {
			pk_fft16<-1> (v);
		} else {
			fft16<-1> (v);
		}
		if (NREG == 1) {
			__syncthreads ();
		}
		{
			f2* w = &lds[NREG - 1][th * kL1 + tl];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				w[16 * k] = v[k];
			}
			__syncthreads ();
			const f2* r = &lds[NREG - 1][t];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				v[k] = r[k * kL1];
			}
		}
		// ---- twiddle conj W^{t k0}, inverse pass 3 (over k0) -----------------
#pragma unroll
		for (int k = 1; k < 16; ++k) {
			v[k] = PK ? pk_cmul<-1> (v[k], tw1[k]) : cmulw<-1> (v[k], tw1[k]);
		}
		if (PK) {
			pk_fft16<-1> (v);
		} else {
			fft16<-1> (v);
		}
		// v[k].x = (fir * A)[t + 256k], v[k].y = (fir * B)[t + 256k]

		// ---- overlap-add in registers ----------------------------------------
		float u[2 * HOPSEG];
#pragma unroll
		for (int j = 0; j < HOPSEG; ++j) {
			u[j]          = acc[j] + v[j].x;
			u[HOPSEG + j] = v[HOPSEG + j].x + v[j].y;
			if (NS > 2) {
				u[HOPSEG + j] += acc[HOPSEG + j];
			}
		}
#pragma unroll
		for (int q = 0; q < NS - 1; ++q) {
#pragma unroll
			for (int j = 0; j < HOPSEG; ++j) {
				float a = v[(q + 1) * HOPSEG + j].y;
				if (q + 2 < NS) {
					a += v[(q + 2) * HOPSEG + j].x;
				}
				if (q + 2 <= NS - 2) {
					a += acc[(q + 2) * HOPSEG + j];
				}
				acc[q * HOPSEG + j] = a;
			}
		}

		// ---- rotate and store --------------------------------------------------
		// ACC: a later FIR partition adds its share of the Hilbert branch to what the
		// first partition stored (partitioned convolution for FIRs longer than one FFT)
		if (emit) {
			if (MODE == MODE_HILBERT) {
#pragma unroll
				for (int k = 0; k < 2 * HOPSEG; ++k) {
					const int64_t pos = sA + t + 256 * k;
					if (pos < r1) {
						out[pos] = ACC ? out[pos] + u[k] : u[k];
					}
				}
			} else if (MODE == MODE_ROT && fast_out) {
				float* __restrict__ oq = out + sA + t;
#pragma unroll
				for (int k = 0; k < 2 * HOPSEG; ++k) {
					oq[256 * k] = ACC ? oq[256 * k] + sa * u[k] : ca * dry[k] + sa * u[k];
				}
			} else {
#pragma unroll
				for (int k = 0; k < 2 * HOPSEG; ++k) {
					const int64_t pos = sA + t + 256 * k;
					if (pos < r1) {
						float cc = ca, ss = sa;
						if (MODE == MODE_RAMP) {
							const f2 cs = reinterpret_cast<const f2*> (p.rot_s)[(int64_t)c * p.n_out + pos];
							cc              = cs.x;
							ss              = cs.y;
						}
						out[pos] = ACC ? out[pos] + ss * u[k] : cc * dry[k] + ss * u[k];
					}
				}
			}
		}
	}
	hist_write_slice (p, blockIdx.y, threadIdx.x);
}

// --------------------------------------------------------------------------------------
// ola4096p2_kernel -- FIRs of 2049..4096 taps (LV2 48 kHz: 3072, LV2 96 kHz / CLI blksiz
// 4096: 4096) as TWO 2048-tap partitions that share one forward transform: a uniformly
// partitioned overlap-add with a frequency-domain delay line,
//     Y_b = X_b H0 + X_{b-1} H1,      hop 2048, 4096-point FFT
// -- the reference LV2 engine's own scheme (src/phaserotate.c:640-657: n_segm partitions,
// one c2r per block) at a partition size that suits the GPU.  Compared with four
// zero-padded 1024-frame hops per 4096 frames it halves the transforms per frame.
// The re/im packing pairs block b of two SUB-RUNS of the workgroup's run (first and second
// half), so the delayed spectrum X_{b-1} of the pair is simply last iteration's registers.
// --------------------------------------------------------------------------------------
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

// DH: dry path from registers, (dry_delay - conv_delay) / 256 in 1..8; DH = 0: dry frames re-read
// from the stream (larger delays); ACC: add this partition pair's share to what an earlier launch stored
// (FIRs of more than 4096 taps: partitions 2s, 2s+1 per launch)
// (fp16 with the dry path in registers does not fit the 168 VGPRs of 3 workgroups per CU without scratch: 2 there)
template <class T, int MODE, int DH, bool ACC = false>
__global__ __launch_bounds__ (256, (T::kWaves == 3 && DH > 0 && MODE != MODE_HILBERT) ? 2 : T::kWaves) void ola4096p2_kernel (const OlaParamsT<typename T::samp> p)
{
	typedef typename T::cplx C;
	typedef typename T::ops  O;
	typedef typename T::samp S;
	constexpr int HOP = 2048;
	static_assert (DH >= 0 && DH <= 8, "dry delay difference within one block, or 0");
	constexpr int NDH = DH > 0 ? DH : 1;

	__shared__ C lds[kLdsRegion];
	__shared__ C twm[256];
	__shared__ C Hs0[2049];
	__shared__ C Hs1[2049];

	const int     t  = threadIdx.x;
	const int     c  = blockIdx.y;
	const int64_t r0 = (int64_t)blockIdx.x * p.run_frames;
	if (r0 >= p.n_out) {
		hist_write_slice (p, blockIdx.y, threadIdx.x);
		return;
	}
	const int64_t r1 = (r0 + p.run_frames < p.n_out) ? r0 + p.run_frames : p.n_out;
	const int64_t Lh = p.run_frames / 2;                      // sub-run length (multiple of HOP)
	const int64_t eP = (r0 + Lh < r1) ? r0 + Lh : r1;         // sub-run P = [r0, eP), Q = [r0 + Lh, r1)

	const S* __restrict__ in   = p.in + (int64_t)c * p.in_stride;
	const S* __restrict__ hist = p.hist ? p.hist + (int64_t)c * p.hist_len : nullptr;
	S* __restrict__ out        = p.out + (int64_t)c * p.out_stride;
	const int64_t n_in             = p.n_in;
	const int     Dc = p.conv_delay, Dd = p.dry_delay;

	const f2* __restrict__ twp = reinterpret_cast<const f2*> (p.tw);
	C tw1[16];
#pragma unroll
	for (int k = 1; k < 16; ++k) {
		const f2 w = twp[(t * k) & 4095];
		tw1[k]     = O::mk (w.x, w.y);
	}
	{
		const f2 w = twp[16 * (((t & 15) * (t >> 4)) & 255)];
		twm[t]     = O::mk (w.x, w.y);
	}
	for (int i = t; i <= 2048; i += 256) { // |H|/4096 = 2.4e-4 is still a normal fp16 number
		const f2  h0 = reinterpret_cast<const f2*> (p.H)[i], h1 = reinterpret_cast<const f2*> (p.H)[4096 + i];
		const int j  = swap_lo_digits (i);
		Hs0[j]       = O::mk (h0.x, h0.y);
		Hs1[j]       = O::mk (h1.x, h1.y);
	}
	float ca = 0.f, sa = 1.f;
	if (MODE == MODE_ROT) {
		ca = p.rot ? p.rot[2 * c] : p.rot_v[2 * c];
		sa = p.rot ? p.rot[2 * c + 1] : p.rot_v[2 * c + 1];
	}

	const int tl = t & 15, th = t >> 4;
	// mirrored bin of (th + 16 tl + 256 k), k >= 8, in the digit-swapped table: 256 (15 - k) + mq
	const int q  = th + 16 * tl;
	const int mq = q == 0 ? 256 : swap_lo_digits (256 - q);
	const int n_it = 2 + (int)((Lh + HOP - 1) / HOP);         // two halo blocks: OLA tail + X_{b-1}
	int64_t   sP = r0 - 2 * HOP, sQ = sP + Lh;

	float acc[16], dh[2 * NDH], nx[16];
	C     zp[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		acc[i] = 0.f;
		zp[i]  = O::mk (0.f, 0.f);
	}
#pragma unroll
	for (int i = 0; i < NDH; ++i) {
		dh[i]       = DH > 0 ? ldv (in, n_in, hist, p.hist_len, sP - Dc - 256 * (DH - i) + t) : 0.f;
		dh[NDH + i] = DH > 0 ? ldv (in, n_in, hist, p.hist_len, sQ - Dc - 256 * (DH - i) + t) : 0.f;
	}
	auto load_blocks = [&] (int64_t aP, int64_t aQ) {
		const bool fast = (aP >= 0) && (aQ + HOP <= n_in);
		if (PROT_ABL2 & 32) { // no stream loads
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				nx[k] = 0.001f * (float)(t + k);
			}
		} else if (fast) {
			// uniform base (SGPR pair) + 32-bit lane offset: saddr addressing, no 64-bit VALU address math
			// (the immediate offset reaches 4 KiB: one base per 1024 frames)
			const PROT_GLOBAL S* __restrict__ bP0 = uniform_ptr (in + aP);
			const PROT_GLOBAL S* __restrict__ bP1 = uniform_ptr (in + aP + 1024);
			const PROT_GLOBAL S* __restrict__ bQ0 = uniform_ptr (in + aQ);
			const PROT_GLOBAL S* __restrict__ bQ1 = uniform_ptr (in + aQ + 1024);
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				nx[k]      = (float)bP0[(unsigned)t + 256u * k];
				nx[4 + k]  = (float)bP1[(unsigned)t + 256u * k];
				nx[8 + k]  = (float)bQ0[(unsigned)t + 256u * k];
				nx[12 + k] = (float)bQ1[(unsigned)t + 256u * k];
			}
		} else {
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				nx[k]     = ldv (in, n_in, hist, p.hist_len, aP + t + 256 * k);
				nx[8 + k] = ldv (in, n_in, hist, p.hist_len, aQ + t + 256 * k);
			}
		}
	};
	load_blocks (sP - Dc, sQ - Dc);
	__syncthreads ();

	const int t0 = t;
	for (int it = 0; it < n_it; ++it, sP += HOP, sQ += HOP) {
		// the thread id, opaque per iteration: keeps the per-lane pointers of the bounds-checked paths and the LDS
		// addresses of the table reads from being hoisted out of the loop into registers (see olapoly_kernel)
		int t = t0;
		asm volatile ("" : "+v"(t));
		const int tl = t & 15, th = t >> 4;
		C     v[16];
		float cur[16];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			cur[k] = nx[k];
		}
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			v[k]     = O::mk (cur[k], cur[8 + k]);
			v[8 + k] = O::mk (0.f, 0.f);
		}
		// spectrum tables, 4 bins (8 reads) per batch, one batch in flight ahead of the multiplies
		typedef PROT_LDS_PTR (T) VC;
		C    hb[2][8];
		auto h_fetch = [&] (int j) {
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int k   = 4 * j + i;
				const int idx = k < 8 ? t + 256 * k : 256 * (15 - k) + mq;
				hb[j & 1][i]     = ((const VC*)Hs0)[idx];
				hb[j & 1][4 + i] = ((const VC*)Hs1)[idx];
			}
		};
		fwd4096w<T, 8> (v, lds, twm, tw1, t, tl, th, [&] () { if (!(PROT_ABL2 & 16)) h_fetch (0); });

		// Y = Z H0 + Z_prev H1; bins >= 2048 use the conjugate of the mirrored bin
#pragma unroll
		for (int j = 0; j < ((PROT_ABL2 & 16) ? 0 : 4); ++j) {
			if (j < 3) {
				h_fetch (j + 1);
			}
			__builtin_amdgcn_sched_barrier (0);
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int k = 4 * j + i;
				const C   z = v[k];
				if (k < 8) {
					v[k] = O::template cmul_acc<1> (O::template cmul<1> (z, hb[j & 1][i]), zp[k], hb[j & 1][4 + i]);
				} else {
					v[k] = O::template cmul_acc<-1> (O::template cmul<-1> (z, hb[j & 1][i]), zp[k], hb[j & 1][4 + i]);
				}
				zp[k] = z;
			}
			__builtin_amdgcn_sched_barrier (0);
		}

		__builtin_amdgcn_sched_barrier (0);
		if (it + 1 < n_it) {
			load_blocks (sP + HOP - Dc, sQ + HOP - Dc);
		}
		__builtin_amdgcn_sched_barrier (0);

		// dry path: from registers ((dry_delay - conv_delay) = 256 * DH frames back) or, for
		// larger delays, re-read from the stream (L2/MALL)
		float dry[16];
		if (DH > 0) {
#pragma unroll
			for (int h = 0; h < 2; ++h) {
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					dry[8 * h + k] = (k >= DH) ? cur[8 * h + (k >= DH ? k - DH : 0)] : dh[NDH * h + (k < DH ? k : 0)];
				}
#pragma unroll
				for (int i = 0; i < NDH; ++i) {
					dh[NDH * h + i] = cur[8 * h + 8 - NDH + i];
				}
			}
		} else if (MODE != MODE_HILBERT && !ACC && it >= 2) {
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				dry[k]     = ldv (in, n_in, hist, p.hist_len, sP + t + 256 * k - Dd);
				dry[8 + k] = ldv (in, n_in, hist, p.hist_len, sQ + t + 256 * k - Dd);
			}
		}

		inv4096w<T> (v, lds, twm, tw1, t, tl, th);

		float u[16];
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			u[j]       = acc[j] + (float)v[j].x;
			u[8 + j]   = acc[8 + j] + (float)v[j].y;
			acc[j]     = (float)v[8 + j].x;
			acc[8 + j] = (float)v[8 + j].y;
		}

		// whole blocks (all but a run's last): no per-frame bounds predicates, constant offsets
		const bool full = (sP + HOP <= eP) && (sQ + HOP <= r1);
		if (it >= 2 && full && MODE == MODE_ROT && !ACC && !(PROT_ABL2 & 64)) {
			PROT_GLOBAL S* __restrict__ oP0 = uniform_ptr (out + sP);
			PROT_GLOBAL S* __restrict__ oP1 = uniform_ptr (out + sP + 1024);
			PROT_GLOBAL S* __restrict__ oQ0 = uniform_ptr (out + sQ);
			PROT_GLOBAL S* __restrict__ oQ1 = uniform_ptr (out + sQ + 1024);
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				oP0[(unsigned)t + 256u * k] = (S)(ca * dry[k] + sa * u[k]);
			}
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				oP1[(unsigned)t + 256u * k] = (S)(ca * dry[4 + k] + sa * u[4 + k]);
			}
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				oQ0[(unsigned)t + 256u * k] = (S)(ca * dry[8 + k] + sa * u[8 + k]);
			}
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				oQ1[(unsigned)t + 256u * k] = (S)(ca * dry[12 + k] + sa * u[12 + k]);
			}
		} else if (it >= 2) {
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const int64_t s0  = h ? sQ : sP;
				const int64_t end = h ? r1 : eP;
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					const int64_t pos = s0 + t + 256 * k;
					if ((PROT_ABL2 & 64) ? (pos < end && u[8 * h + k] == 123.456f) : (pos < end)) { // 64: (almost) no stores
						if (MODE == MODE_HILBERT) {
							out[pos] = ACC ? (S)((float)out[pos] + u[8 * h + k]) : (S)u[8 * h + k];
						} else {
							float cc = ca, ss = sa;
							if (MODE == MODE_RAMP) {
								const f2 cs = reinterpret_cast<const f2*> (p.rot_s)[(int64_t)c * p.n_out + pos];
								cc          = cs.x;
								ss          = cs.y;
							}
							out[pos] = ACC ? (S)((float)out[pos] + ss * u[8 * h + k]) : (S)(cc * dry[8 * h + k] + ss * u[8 * h + k]);
						}
					}
				}
			}
		}
	}
	hist_write_slice (p, blockIdx.y, threadIdx.x);
}


template <class T, int MODE>
static hipError_t launch_p2_m (const OlaParamsT<typename T::samp>& p, dim3 grid, hipStream_t s)
{
	const int  diff = p.dry_delay - p.conv_delay;
	const int  dd   = (diff > 0 && diff % 256 == 0) ? diff / 256 : 0;
	const dim3 block (256);
	if (p.accumulate) {
		hipLaunchKernelGGL ((ola4096p2_kernel<T, MODE, 0, true>), grid, block, 0, s, p);
	} else if (MODE != MODE_HILBERT && dd == 6) {
		hipLaunchKernelGGL ((ola4096p2_kernel<T, MODE, 6>), grid, block, 0, s, p);
	} else if (MODE != MODE_HILBERT && dd == 8) {
		hipLaunchKernelGGL ((ola4096p2_kernel<T, MODE, 8>), grid, block, 0, s, p);
	} else {
		hipLaunchKernelGGL ((ola4096p2_kernel<T, MODE, 0>), grid, block, 0, s, p);
	}
	return hipGetLastError ();
}

template <class T>
static hipError_t launch_p2_t (const OlaParamsT<typename T::samp>& p, int n_chn, hipStream_t s)
{
	if (p.n_out <= 0) {
		return hipSuccess;
	}
	const int64_t runs = (p.n_out + p.run_frames - 1) / p.run_frames;
	dim3          grid ((unsigned)runs, (unsigned)n_chn);
	if (p.hilbert_only) {
		return launch_p2_m<T, MODE_HILBERT> (p, grid, s);
	}
	if (p.rot_s != nullptr) {
		return launch_p2_m<T, MODE_RAMP> (p, grid, s);
	}
	return launch_p2_m<T, MODE_ROT> (p, grid, s);
}

hipError_t launch_ola4096_p2 (const OlaParams& p, int n_chn, hipStream_t s)
{
	return launch_p2_t<T32> (p, n_chn, s);
}

hipError_t launch_ola4096_p2_f16 (const OlaParamsH& p, int n_chn, hipStream_t s)
{
	return launch_p2_t<T16> (p, n_chn, s);
}

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
	// cache line; the timeline-only build shows the XCDs within +-3 %, tools/xcd_timeline.py.)
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

template <int HOPSEG, int MODE>
static hipError_t launch_ola4096_m (const OlaParams& p, dim3 grid, hipStream_t s)
{
	dim3 block (256);
	if (p.accumulate) {
		hipLaunchKernelGGL ((ola4096_kernel<HOPSEG, 1, 3, MODE, true>), grid, block, 0, s, p);
		return hipGetLastError ();
	}
	// dry path from registers when the delay difference is a small multiple of 256 frames
	const int dd = p.dry_delay - p.conv_delay;
	if (MODE != MODE_HILBERT && dd > 0 && dd % 256 == 0) {
		switch (dd / 256) {
			case 2: hipLaunchKernelGGL ((ola4096_kernel<HOPSEG, 1, 3, MODE, false, 2>), grid, block, 0, s, p); return hipGetLastError ();
			case 4: hipLaunchKernelGGL ((ola4096_kernel<HOPSEG, 1, 3, MODE, false, 4>), grid, block, 0, s, p); return hipGetLastError ();
			case 6: hipLaunchKernelGGL ((ola4096_kernel<HOPSEG, 1, 3, MODE, false, 6>), grid, block, 0, s, p); return hipGetLastError ();
			default: break;
		}
	}
	hipLaunchKernelGGL ((ola4096_kernel<HOPSEG, 1, 3, MODE>), grid, block, 0, s, p);
	return hipGetLastError ();
}

template <int HOPSEG>
static hipError_t launch_ola4096_t (const OlaParams& p, dim3 grid, hipStream_t s)
{
	if (p.hilbert_only) {
		return launch_ola4096_m<HOPSEG, MODE_HILBERT> (p, grid, s);
	}
	if (p.rot_s != nullptr) {
		return launch_ola4096_m<HOPSEG, MODE_RAMP> (p, grid, s);
	}
	return launch_ola4096_m<HOPSEG, MODE_ROT> (p, grid, s);
}

hipError_t launch_ola4096 (const OlaParams& p, int hopseg, int n_chn, hipStream_t s)
{
	if (p.n_out <= 0) {
		return hipSuccess;
	}
	const int64_t runs = (p.n_out + p.run_frames - 1) / p.run_frames;
	dim3          grid ((unsigned)runs, (unsigned)n_chn);
	if (hopseg == 4) {
		return launch_ola4096_t<4> (p, grid, s);
	} else if (hopseg == 8) {
		return launch_ola4096_t<8> (p, grid, s);
	}
	return hipErrorInvalidValue;
}

// ------------------------------------------------------------------- reductions (the sweep itself: prot_sweep.hip)
// max over the 64 lanes of a non-negative value, result valid in lane 63
PROT_DEV float wave_max_nonneg (float v)
{
#define PROT_DPP_MAX(ctrl, rmask)                                                               \
	v = fmaxf (v, __int_as_float (__builtin_amdgcn_update_dpp (0, __float_as_int (v), ctrl,     \
	                                                           rmask, 0xf, true)))
	PROT_DPP_MAX (0x111, 0xf); // row_shr:1   (out-of-row source reads 0 = identity for |.|)
	PROT_DPP_MAX (0x112, 0xf); // row_shr:2
	PROT_DPP_MAX (0x114, 0xf); // row_shr:4
	PROT_DPP_MAX (0x118, 0xf); // row_shr:8   lane 15 of each row = row max
	PROT_DPP_MAX (0x142, 0xa); // row_bcast:15 into rows 1 and 3
	PROT_DPP_MAX (0x143, 0xc); // row_bcast:31 into rows 2 and 3
#undef PROT_DPP_MAX
	return v;
}

// ------------------------------------------------------------------ absmax
__global__ __launch_bounds__ (256) void absmax_kernel (const float* __restrict__ x, int64_t x_stride,
                                                       int64_t n, float* peaks, int slot, int chn0)
{
	const int c                   = chn0 + blockIdx.y;
	const float* __restrict__ xc = x + (int64_t)c * x_stride;
	float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
	// 16 bytes per lane, four loads in flight (a dword per lane and one dependent max per load ran at 3.2 TB/s: 0.55 ms of
	// cfg4's 3.7); the unaligned head (row pointers need not be 16-byte aligned) and the tail go frame by frame
	const int64_t head = ((16 - (reinterpret_cast<uintptr_t> (xc) & 15)) & 15) / 4;
	const int64_t h    = head < n ? head : n;
	const int64_t n4   = (n - h) / 4;
	const float4* __restrict__ x4 = reinterpret_cast<const float4*> (xc + h);
	// a workgroup takes 16 KiB of consecutive frames per iteration (four loads of 16 bytes per lane in flight), the grid one
	// dense window of the stream
	const int64_t step = (int64_t)gridDim.x * 1024;
	int64_t       i    = (int64_t)blockIdx.x * 1024 + threadIdx.x;
	for (; i + 768 < n4; i += step) {
		const float4 a = x4[i], b = x4[i + 256], d = x4[i + 512], e = x4[i + 768];
		m0 = fmaxf (fmaxf (m0, fabsf (a.x)), fmaxf (fabsf (a.y), fmaxf (fabsf (a.z), fabsf (a.w))));
		m1 = fmaxf (fmaxf (m1, fabsf (b.x)), fmaxf (fabsf (b.y), fmaxf (fabsf (b.z), fabsf (b.w))));
		m2 = fmaxf (fmaxf (m2, fabsf (d.x)), fmaxf (fabsf (d.y), fmaxf (fabsf (d.z), fabsf (d.w))));
		m3 = fmaxf (fmaxf (m3, fabsf (e.x)), fmaxf (fabsf (e.y), fmaxf (fabsf (e.z), fabsf (e.w))));
	}
	for (int k = 0; k < 4; ++k) { // (the ragged last iteration of the one workgroup that meets the end)
		if (i + 256 * k < n4 && i < n4) {
			const float4 a = x4[i + 256 * k];
			m0 = fmaxf (fmaxf (m0, fabsf (a.x)), fmaxf (fabsf (a.y), fmaxf (fabsf (a.z), fabsf (a.w))));
		}
	}
	if (blockIdx.x == 0) {
		for (int64_t k = threadIdx.x; k < h; k += 256) {
			m1 = fmaxf (m1, fabsf (xc[k]));
		}
		for (int64_t k = h + 4 * n4 + threadIdx.x; k < n; k += 256) {
			m2 = fmaxf (m2, fabsf (xc[k]));
		}
	}
	float m = wave_max_nonneg (fmaxf (fmaxf (m0, m1), fmaxf (m2, m3)));
	if ((threadIdx.x & 63) == 63) {
		atomicMax (reinterpret_cast<unsigned*> (peaks + (int64_t)c * kMaxSample + slot), __float_as_uint (m));
	}
}

hipError_t launch_absmax (const float* x, int64_t x_stride, int64_t n, float* peaks, int slot,
                          int chn0, int n_chn, hipStream_t s)
{
	if (n <= 0 || n_chn <= 0) {
		return hipSuccess;
	}
	int64_t gx = (n + 4095) / 4096; // 4096 frames per workgroup iteration
	if (gx > 2048) {
		gx = 2048;
	}
	dim3 grid ((unsigned)gx, (unsigned)n_chn), block (256);
	hipLaunchKernelGGL (absmax_kernel, grid, block, 0, s, x, x_stride, n, peaks, slot, chn0);
	return hipGetLastError ();
}

// ------------------------------------------------------------ small movers
__global__ __launch_bounds__ (256) void zero_regions_kernel (const ZeroRegions z)
{
	uint32_t* const q = z.p[blockIdx.y];
	const uint64_t  n = z.n[blockIdx.y];
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
		q[i] = 0u;
	}
}

hipError_t launch_zero_regions (const ZeroRegions& z, hipStream_t s)
{
	uint64_t most = 0;
	for (int i = 0; i < kZeroRegions; ++i) {
		most = std::max (most, z.p[i] ? z.n[i] : 0);
	}
	if (most == 0) {
		return hipSuccess;
	}
	const dim3 grid ((unsigned)std::min<uint64_t> ((most + 255) / 256, 512), kZeroRegions), block (256);
	hipLaunchKernelGGL (zero_regions_kernel, grid, block, 0, s, z);
	return hipGetLastError ();
}

template <class S>
__global__ void hist_update_kernel (const S* __restrict__ in, int64_t in_stride, int64_t n_in,
                                    const S* __restrict__ hist_in, S* __restrict__ hist_out,
                                    int hist_len, int64_t shift)
{
	const int c = blockIdx.y;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < hist_len) {
		hist_out[(int64_t)c * hist_len + i] =
		    (S)ldv (in + (int64_t)c * in_stride, n_in, hist_in + (int64_t)c * hist_len, hist_len,
		            n_in + shift - hist_len + i);
	}
}

hipError_t launch_hist_update (const float* in, int64_t in_stride, int64_t n_in, const float* hist_in,
                               float* hist_out, int hist_len, int n_chn, hipStream_t s, int64_t shift)
{
	dim3 grid ((unsigned)((hist_len + 255) / 256), (unsigned)n_chn), block (256);
	hipLaunchKernelGGL (hist_update_kernel<float>, grid, block, 0, s, in, in_stride, n_in, hist_in, hist_out, hist_len, shift);
	return hipGetLastError ();
}

hipError_t launch_hist_update_f16 (const _Float16* in, int64_t in_stride, int64_t n_in, const _Float16* hist_in,
                                   _Float16* hist_out, int hist_len, int n_chn, hipStream_t s)
{
	dim3 grid ((unsigned)((hist_len + 255) / 256), (unsigned)n_chn), block (256);
	hipLaunchKernelGGL (hist_update_kernel<_Float16>, grid, block, 0, s, in, in_stride, n_in, hist_in, hist_out, hist_len, (int64_t)0);
	return hipGetLastError ();
}

__global__ void deinterleave_kernel (const float* __restrict__ inter, float* __restrict__ planar,
                                     int64_t stride, int64_t n, int n_chn)
{
	const int64_t total = n * n_chn;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
	     i += (int64_t)gridDim.x * blockDim.x) {
		const int64_t f = i / n_chn;
		const int     c = (int)(i - f * n_chn);
		planar[(int64_t)c * stride + f] = inter[i];
	}
}

__global__ void interleave_kernel (const float* __restrict__ planar, int64_t stride,
                                   float* __restrict__ inter, int64_t n, int n_chn)
{
	const int64_t total = n * n_chn;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
	     i += (int64_t)gridDim.x * blockDim.x) {
		const int64_t f = i / n_chn;
		const int     c = (int)(i - f * n_chn);
		inter[i]        = planar[(int64_t)c * stride + f];
	}
}

static unsigned mover_grid (int64_t total)
{
	int64_t g = (total + 255) / 256;
	return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

hipError_t launch_deinterleave (const float* inter, float* planar, int64_t stride, int64_t n,
                                int n_chn, hipStream_t s)
{
	if (n <= 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL (deinterleave_kernel, dim3 (mover_grid (n * n_chn)), dim3 (256), 0, s, inter,
	                    planar, stride, n, n_chn);
	return hipGetLastError ();
}

hipError_t launch_interleave (const float* planar, int64_t stride, float* inter, int64_t n, int n_chn,
                              hipStream_t s)
{
	if (n <= 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL (interleave_kernel, dim3 (mover_grid (n * n_chn)), dim3 (256), 0, s, planar,
	                    stride, inter, n, n_chn);
	return hipGetLastError ();
}

} // namespace prot

#ifdef PROT_STAMPS
extern "C" int prot_debug_stamps (unsigned long long* out, int n, int reset)
{
	unsigned long long h[32];
	if (hipMemcpyFromSymbol (h, HIP_SYMBOL (prot::g_stamps), sizeof (h)) != hipSuccess) {
		return -1;
	}
	for (int i = 0; i < n && i < 32; ++i) {
		out[i] = h[i];
	}
	if (reset) {
		unsigned long long z[32] = {};
		if (hipMemcpyToSymbol (HIP_SYMBOL (prot::g_stamps), z, sizeof (z)) != hipSuccess) {
			return -1;
		}
	}
	return 0;
}
extern "C" int prot_debug_wg_times (unsigned long long* out, int n_wg)
{
	return hipMemcpyFromSymbol (out, HIP_SYMBOL (prot::g_wg_t), sizeof (unsigned long long) * 4 * (size_t)n_wg) == hipSuccess ? 0 : -1;
}
#endif
