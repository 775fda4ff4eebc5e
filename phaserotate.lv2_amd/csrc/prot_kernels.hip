// prot_kernels.hip -- gfx950 (MI355X, CDNA4) overlap-add kernels of the phase-rotation hot path, and the small
// movers around them.  wave64 throughout; no hipFFT/rocFFT (the OLA path is a butterfly network, not a contraction).
//
// The kernels of the path, by file:
//   prot_olasave.hip  olasave_kernel   -- polyphase OVERLAP-SAVE, one partition of the odd taps, no spectrum table,
//                                         generated gfx950 assembly, four workgroups per CU: the default for every
//                                         Hilbert FIR of 1024..6144 taps (LV2 48 / 96 kHz, CLI blksiz 1024..4096) --
//                                         the headline kernel
//   prot_olapoly.hip  olapoly_kernel   -- polyphase overlap-add with a spectrum table: FIRs of 8192 taps and more
//                                         (two partitions of the odd taps per launch, accumulating launches beyond),
//                                         7168-frame blocks (CLI blksiz 1024: HS = 14), PROT_CFG_OVERLAP_ADD, fp16
//   here              ola4096p2_kernel -- direct form, two 2048-tap partitions sharing a forward transform:
//                                         PROT_CFG_DIRECT_FORM (cross-check of the polyphase forms), fp16 at 4096 taps
//   here              ola4096_kernel   -- direct form, one partition: FIRs below 1024 taps (CLI blksiz 64..512),
//                                         PROT_CFG_SINGLE_FFT_PER_HOP; described below
//   prot_sweep.hip    sweep_kernel     -- K7, the per-angle abs-max on the matrix core over compacted survivors
// They share the 4096-point transform pair (fwd4096w / inv4096w in prot_fft4096.h, or the inline version of ola4096_kernel), the
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
#include "prot_fft4096.h"

#include <algorithm>
#include <cstdlib>


namespace prot {

// ------------------------------------------------------------ OLA, FFT-4096


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
