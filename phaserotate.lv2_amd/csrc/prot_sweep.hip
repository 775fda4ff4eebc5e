// prot_sweep.hip -- K7, the CLI's per-angle abs-max (calc_rotated_peak, cli/phase-rotate.cc:98-121, called for
// every angle of the search at :409-428; cli/dsp_peak_calc.h) as rank-2 matrix-core products over the frames that
// can still matter.
//
//   |ca_a d_i + sa_a h_i| for 32 angles x 32 frames is the rank-2 product [ca sa] (32x2) . [d; h] (2x32): one
//   v_mfma_f32_32x32x2_f32.  f32 MFMA is an exact k-ordered fmaf chain, fma(sa, h, ca*d) -- the arithmetic of the
//   oracle's loop -- and the value of a frame does not depend on the column it sits in.  Lane l of the 32x32 result
//   holds column (l & 31) for 16 of the 32 angles, so running maxima stay lane-local for the whole launch: no per-tile
//   cross-lane reduction.  A wave owns up to 3 blocks of 32 angles (48 VGPRs of maxima); the 4 waves of a workgroup
//   cover the 360 angles (12 blocks) and read the same frames.
//
//   Pruning (exact): |ca d + sa h| <= sqrt(d^2 + h^2) =: r for every angle, so a frame whose r is below a LOWER BOUND
//   of the smallest final peak among this wave's angles cannot change any maximum.  Every frame is tested (a lane
//   holds d and h of its own frame: r^2, one compare); the survivors of 64 frames are COMPACTED -- ballot, v_mbcnt,
//   one ds_write_b64 of (d, h) per survivor into the wave's own LDS ring -- and the matrix core only ever sees full
//   tiles of survivors (two at a time, so v_max3 folds both).  What is left in the ring at the end is
//   padded with zeros (|0| never raises a maximum >= 0).  The table is bit-identical to evaluating everything: max is
//   exact, order-free, and skipped frames would have lost every comparison; 1e-4 of relative margin on r^2 covers the
//   rounding of the float evaluation against the exact inequality.
//   The bound: the smallest of the table entries as they stand at launch (earlier chunks; they only grow), the
//   smallest of this wave's own lane-local running maxima, refreshed once per super-tile, and the largest bound any
//   wave of the same angle group has published (one atomicMax per refresh).
//   Material on which nearly everything survives (a signal that keeps growing; the first tiles of the first chunk,
//   when no bound exists yet) would pay for the test and the LDS round trip on top of the full evaluation: a wave
//   that saw more than 7/8 of a 16-tile window survive feeds tiles straight from registers for a while, then refreshes
//   its bound and probes one window again -- the pause doubles while probes keep failing and resets when one
//   succeeds.  Tiles that went through untested are counted separately (prot_sweep_stats_ex).
#include "prot_dev.h"
#include "prot_kernels.h"

#include <algorithm>

namespace prot {

typedef float f32x16 __attribute__ ((ext_vector_type (16)));

constexpr int kSuper = 256;  // 32-frame tiles per workgroup iteration (8192 frames)
constexpr int kNL    = 4;    // 64-frame groups per step; the frames of step s + 1 are requested before the work of step s
constexpr int kSteps = kSuper * 32 / (64 * kNL); // steps per workgroup iteration
constexpr int kRing  = 512;  // (d, h) entries of one wave's survivor ring: < 64 left over + the 64 kNL frames of one step
static_assert (kRing >= 64 + 64 * kNL && (kRing & (kRing - 1)) == 0, "ring too small");
constexpr int kWin   = 16;   // tiles per decision window of the switch-off (a workgroup sees only 256..512 tiles per launch)
constexpr int kPause0 = 4, kPauseMax = 1024; // steps (of 2 kNL tiles) fed untested after a failed window: first, cap

template <int NBLK, bool PRUNE>
__global__ __launch_bounds__ (256, 3) void sweep_kernel (const SweepParams p)
{
	__shared__ float2 ring_s[PRUNE ? 4 * kRing : 1];

	const int tid  = threadIdx.x;
	const int w    = __builtin_amdgcn_readfirstlane (tid >> 6); // wave-uniform by construction
	const int lane = tid & 63;
	const int j    = lane & 31; // frame within a tile / angle within a block (A operand)
	const int kk   = lane >> 5; // k index of the rank-2 product: 0 -> (cos, dry), 1 -> (sin, hilbert)
	const int c    = p.chn0 + blockIdx.y;

	const float* __restrict__ x    = p.x + (int64_t)c * p.x_stride;
	const float* __restrict__ hist = p.hist ? p.hist + (int64_t)c * p.hist_len : nullptr;
	const float* __restrict__ u    = p.u + (int64_t)c * p.u_stride;

	const int nb   = (p.n_angles + 31) >> 5;      // angle blocks in this call (<= 12)
	const int G    = nb >= 3 ? 4 : nb;            // wave groups: 4, 2 or 1
	const int g    = w % G;                       // my group -> blocks g, g + G, g + 2G
	const int sub  = w / G;                       // waves of one group split the tiles
	const int nsub = 4 / G;
	if (g >= nb) {
		return; // 65..96 angles put three blocks on four groups: nothing to do (no workgroup barrier anywhere below)
	}

	// NBLK = blocks per wave (1..3); a wave whose group has fewer blocks multiplies by an
	// all-zero A operand there (maxima stay 0, nothing is published for them)
	float a_op[NBLK];
#pragma unroll
	for (int q = 0; q < NBLK; ++q) {
		const int idx = (g + q * G) * 32 + j;
		float     v   = 0.f;
		if (g + q * G < nb && idx < p.n_angles) {
			const int a = p.angles[idx];
			v           = p.lut[2 * a + (kk ? 0 : 1)]; // lut = {sin, cos}
		}
		a_op[q] = v;
	}
	f32x16 m[NBLK];
#pragma unroll
	for (int q = 0; q < NBLK; ++q) {
#pragma unroll
		for (int r = 0; r < 16; ++r) {
			m[q][r] = 0.f;
		}
	}
	const f32x16 zero = m[0];

	// m[q][r] of this lane belongs to a requested angle (C/D layout of 32x32: row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
	auto valid = [&] (int q, int r) {
		const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
		const int idx = (g + q * G) * 32 + row;
		return g + q * G < nb && idx < p.n_angles;
	};
	auto uniform_f = [] (float v) { return __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (v))); };
	auto wave_min = [] (float v) {
		v = fminf (v, __shfl_xor (v, 1));
		v = fminf (v, __shfl_xor (v, 2));
		v = fminf (v, __shfl_xor (v, 4));
		v = fminf (v, __shfl_xor (v, 8));
		v = fminf (v, __shfl_xor (v, 16));
		v = fminf (v, __shfl_xor (v, 32));
		return v;
	};
	// two tiles of frames through the matrix core and into the running maxima
	auto evaluate = [&] (float b0, float b1) {
#pragma unroll
		for (int q = 0; q < NBLK; ++q) {
			const f32x16 r0 = __builtin_amdgcn_mfma_f32_32x32x2f32 (a_op[q], b0, zero, 0, 0, 0);
			const f32x16 r1 = __builtin_amdgcn_mfma_f32_32x32x2f32 (a_op[q], b1, zero, 0, 0, 0);
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				m[q][r] = fmaxf (m[q][r], fmaxf (fabsf (r0[r]), fabsf (r1[r])));
			}
		}
	};

	// ---- pruning state: wave-uniform, kept in SGPRs (with NBLK = 3 the kernel sits at the 168-VGPR boundary of
	// 3 waves per SIMD).  This file is compiled with -ffast-math (finite-math-only): no infinities as sentinels; every
	// wave that gets here has at least one valid entry, so kBig never survives a wave_min.
	constexpr float kBig = 1.0e18f; // finite, and its square still is
	float thr2 = 0.f;               // frames with r^2 <= thr2 cannot matter (margin included)
	if (PRUNE) {
		float lo = kBig;
#pragma unroll
		for (int q = 0; q < NBLK; ++q) {
			const int idx = (g + q * G) * 32 + j;
			if (g + q * G < nb && idx < p.n_angles) {
				lo = fminf (lo, __builtin_nontemporal_load (&p.peaks[(int64_t)c * kMaxSample + p.angles[idx]]));
			}
		}
		lo   = wave_min (lo);
		thr2 = uniform_f (lo * lo * 0.9999f);
	}
	// Every wave of angle group g (of this channel) bounds the SAME quantity -- the smallest final peak among the group's
	// angles -- from below, so the largest bound any of them has found is valid for all: one atomicMax per refresh on
	// p.thr[c][g] (non-negative floats order as their bit patterns; zeroed by the host before the launch).  Without it a
	// wave whose own stripes are quiet never learns that another wave has already seen the peaks.
	unsigned* const thr_slot = p.thr ? reinterpret_cast<unsigned*> (p.thr) + ((int64_t)c * 4 + g) : nullptr;
	auto share = [&] (float lo) { // lo: this wave's bound (not squared); returns the best bound known
		if (thr_slot != nullptr) {
			float best = lo;
			if (lane == 0) {
				const unsigned old = atomicMax (thr_slot, __float_as_uint (lo));
				best               = fmaxf (lo, __uint_as_float (old));
			}
			lo = __shfl (best, 0);
		}
		return lo;
	};
	auto refresh = [&] () { // raise thr2 to what the running maxima (mine, and my group's) prove by now
		float lo = kBig;
#pragma unroll
		for (int q = 0; q < NBLK; ++q) {
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				if (valid (q, r)) {
					lo = fminf (lo, m[q][r]);
				}
			}
		}
		lo   = share (wave_min (lo));
		thr2 = uniform_f (fmaxf (thr2, lo * lo * 0.9999f));
	};
	if (PRUNE && thr2 > 0.f) {
		const float b = share (sqrtf (thr2));
		thr2          = uniform_f (fmaxf (thr2, b * b * 0.9999f));
	}

	float* const ring = reinterpret_cast<float*> (ring_s + (PRUNE ? w * kRing : 0));
	unsigned     rd = 0, wr = 0;                    // ring cursors (entries, free-running)
	bool         testing   = true;
	int          pause     = kPause0, pause_left = 0;
	int          win_tiles = 0, win_surv = 0;
	unsigned     n_seen = 0, n_eval = 0, n_untested = 0, n_surv = 0; // tiles, tiles, tiles, frames

	// A lane owns one frame of each 64-frame group: d from the dry stream, h from the Hilbert branch, both in its own
	// registers -- the test needs no cross-lane traffic, and one v_permlane32_swap turns a group into the B operands of
	// its two 32-frame tiles ([d_0..31 | h_0..31] and [d_32..63 | h_32..63]) for the untested feed.
	const int64_t span  = (int64_t)kSuper * 32;
	bool          first = true;
	for (int64_t base = p.t_begin + (int64_t)blockIdx.x * span; base < p.t_end; base += (int64_t)gridDim.x * span) {
		if (PRUNE && !first) {
			refresh ();
		}
		first = false;
		const bool fast = (base + span <= p.t_end) && (base - p.delay >= 0) && (base + span - p.delay <= p.n_x) &&
		                  (base + p.t_abs0 >= p.dry_from);
		auto load_groups = [&] (float (&dt)[kNL], float (&ht)[kNL], int sp) {
			const int64_t t0 = base + (int64_t)sp * (64 * kNL) + lane;
			if (fast) {
				const float* __restrict__ xs = x + t0 - p.delay;
				const float* __restrict__ us = u + t0;
#pragma unroll
				for (int i = 0; i < kNL; ++i) {
					dt[i] = xs[64 * i];
					ht[i] = us[64 * i];
				}
			} else {
				// edges of the stream (history, silence, the dry-less first half block): the same loads, branch-free -- an
				// element outside everything reads a frame that certainly exists (u[t_begin]) and is replaced by 0; no
				// control flow between the loads, so the s_waitcnt before the work counts loads instead of draining
				const float* const safe = u + p.t_begin;
#pragma unroll
				for (int i = 0; i < kNL; ++i) {
					const int64_t ti  = t0 + 64 * i;
					const int64_t ix  = ti - p.delay, ih = ix + p.hist_len;
					const bool    in  = ti < p.t_end;
					const bool    okx = ix >= 0 && ix < p.n_x, okh = ix < 0 && ih >= 0 && hist != nullptr;
					const bool    okd = in && (okx || okh) && ti + p.t_abs0 >= p.dry_from;
					const float*  qd  = okx ? x + ix : (okh ? hist + ih : safe);
					const float*  qh  = in ? u + ti : safe;
					const float   vd = *qd, vh = *qh;
					dt[i]             = okd ? vd : 0.f;
					ht[i]             = in ? vh : 0.f;
				}
			}
		};
		float dcur[kNL], hcur[kNL], dnxt[kNL], hnxt[kNL];
		load_groups (dcur, hcur, sub);
		for (int sp = sub; sp < kSteps; sp += nsub) {
			if (sp + nsub < kSteps) {
				load_groups (dnxt, hnxt, sp + nsub);
			}
			if (!PRUNE) {
#pragma unroll
				for (int i = 0; i < kNL; ++i) {
					const auto sw = __builtin_amdgcn_permlane32_swap (__float_as_uint (dcur[i]), __float_as_uint (hcur[i]), false, false);
					evaluate (__uint_as_float (sw[0]), __uint_as_float (sw[1]));
				}
			} else {
				// one instance of the evaluation serves both feeds (an unrolled register feed next to the ring loop pushed
				// the three-block kernel over 168 VGPRs, i.e. down to two waves per SIMD)
				int direct = 0; // groups still to be fed straight from dcur / hcur
				if (testing) {
					int surv = 0;
#pragma unroll
					for (int i = 0; i < kNL; ++i) {
						const float    d = dcur[i], h = hcur[i];
						const bool     live = d * d + h * h > thr2;
						const uint64_t mask = __builtin_amdgcn_ballot_w64 (live);
						if (live) {
							const unsigned at = (wr + __builtin_amdgcn_mbcnt_hi ((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo ((unsigned)mask, 0u))) & (kRing - 1);
							*reinterpret_cast<float2*> (ring + 2 * at) = make_float2 (d, h);
						}
						const int nl = __builtin_popcountll (mask);
						wr += (unsigned)nl;
						surv += nl;
					}
					n_surv += (unsigned)surv;
					win_tiles += 2 * kNL;
					win_surv += surv;
				} else {
					direct = kNL;
					n_untested += 2 * kNL;
					n_eval += 2 * kNL;
				}
				n_seen += 2 * kNL;
				__builtin_amdgcn_wave_barrier ();
				for (;;) {
					float b0, b1;
					if (direct > 0) {
						const auto sw = __builtin_amdgcn_permlane32_swap (__float_as_uint (dcur[0]), __float_as_uint (hcur[0]), false, false);
						b0 = __uint_as_float (sw[0]);
						b1 = __uint_as_float (sw[1]);
#pragma unroll
						for (int i = 0; i + 1 < kNL; ++i) {
							dcur[i] = dcur[i + 1];
							hcur[i] = hcur[i + 1];
						}
						--direct;
					} else if (wr - rd >= 64u) {
						b0 = ring[2 * ((rd + j) & (kRing - 1)) + kk];
						b1 = ring[2 * ((rd + 32 + j) & (kRing - 1)) + kk];
						rd += 64u;
						n_eval += 2;
					} else {
						break;
					}
					evaluate (b0, b1);
				}
				__builtin_amdgcn_wave_barrier ();
				if (!testing) {
					if (--pause_left <= 0) {
						testing = true;
						refresh ();
					}
				} else if (win_tiles >= kWin) {
					if (win_surv * 8 > win_tiles * 32 * 7) {
						testing    = false;
						pause_left = pause;
						pause      = pause * 2 < kPauseMax ? pause * 2 : kPauseMax;
					} else {
						pause = kPause0;
					}
					win_tiles = win_surv = 0;
				}
			}
#pragma unroll
			for (int i = 0; i < kNL; ++i) {
				dcur[i] = dnxt[i];
				hcur[i] = hnxt[i];
			}
		}
	}
	if (PRUNE) {
		const unsigned left = wr - rd; // < 64
		if (left > 0u) {
			float b0 = ring[2 * ((rd + j) & (kRing - 1)) + kk];
			float b1 = ring[2 * ((rd + 32 + j) & (kRing - 1)) + kk];
			b0       = (unsigned)j < left ? b0 : 0.f;
			b1       = (unsigned)j + 32u < left ? b1 : 0.f;
			evaluate (b0, b1);
			n_eval += left > 32u ? 2 : 1;
		}
		if (p.stats != nullptr && lane == 0) {
			atomicAdd (&p.stats[0], (unsigned long long)n_eval);
			atomicAdd (&p.stats[1], (unsigned long long)n_seen);
			atomicAdd (&p.stats[2], (unsigned long long)n_untested);
			atomicAdd (&p.stats[3], (unsigned long long)n_surv);
		}
	}
	// reduce over the 32 columns (lanes of one half), then one atomic per angle
	unsigned* gp = reinterpret_cast<unsigned*> (p.peaks + (int64_t)c * kMaxSample);
#pragma unroll
	for (int q = 0; q < NBLK; ++q) {
#pragma unroll
		for (int r = 0; r < 16; ++r) {
			float v = m[q][r];
			v       = fmaxf (v, __shfl_xor (v, 1));
			v       = fmaxf (v, __shfl_xor (v, 2));
			v       = fmaxf (v, __shfl_xor (v, 4));
			v       = fmaxf (v, __shfl_xor (v, 8));
			v       = fmaxf (v, __shfl_xor (v, 16));
			const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
			const int idx = (g + q * G) * 32 + row;
			if (j == 0 && g + q * G < nb && idx < p.n_angles) {
				atomicMax (&gp[p.angles[idx]], __float_as_uint (v));
			}
		}
	}
}

hipError_t launch_sweep (const SweepParams& p, int n_chn, hipStream_t s)
{
	if (p.t_end <= p.t_begin || p.n_angles <= 0 || n_chn <= 0) {
		return hipSuccess;
	}
	const int nb   = (p.n_angles + 31) / 32;
	const int nblk = nb >= 3 ? (nb + 3) / 4 : 1;
	typedef void (*kern_t) (SweepParams);
	static const kern_t kern[2][4] = {
		{ nullptr, sweep_kernel<1, false>, sweep_kernel<2, false>, sweep_kernel<3, false> },
		{ nullptr, sweep_kernel<1, true>, sweep_kernel<2, true>, sweep_kernel<3, true> },
	};
	const int    pr = p.prune ? 1 : 0;
	const kern_t fn = kern[pr][nblk];
	// persistent grid: exactly the workgroups the chip keeps resident, so the grid-stride loop ends without a ragged
	// last round
	static int slots[2][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
	if (slots[pr][nblk] == 0) {
		int dev = 0, cus = 256, occ = 2;
		(void)hipGetDevice (&dev);
		(void)hipDeviceGetAttribute (&cus, hipDeviceAttributeMultiprocessorCount, dev);
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor (&occ, (const void*)fn, 256, 0) != hipSuccess || occ < 1) {
			occ = 2;
		}
		slots[pr][nblk] = cus * occ;
	}
	const int64_t span  = (int64_t)kSuper * 32;
	const int64_t tiles = (p.t_end - p.t_begin + span - 1) / span;
	const int64_t gx    = std::min<int64_t> (tiles, std::max (1, slots[pr][nblk] / n_chn));
	const dim3    grid ((unsigned)gx, (unsigned)n_chn), block (256);
	hipLaunchKernelGGL (fn, grid, block, 0, s, p);
	return hipGetLastError ();
}

} // namespace prot
