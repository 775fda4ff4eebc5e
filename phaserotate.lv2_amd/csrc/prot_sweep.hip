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
//   of the smallest final peak among the workgroup's angles cannot change any maximum.  Every frame is tested ONCE per
//   workgroup (wave w its quarter of a step: a lane holds d and h of its own frame: r^2, one compare); the survivors of
//   64 frames are COMPACTED -- ballot, v_mbcnt, one ds_write_b64 of (d, h) per survivor into the wave's own LDS ring --
//   and after a barrier every wave runs the full tiles of all four rings through the matrix core with its own angles (two
//   tiles at a time, so v_max3 folds both).  What is left in the rings at the end is padded with zeros (|0| never raises
//   a maximum >= 0).  The table is bit-identical to evaluating everything: max is exact, order-free, and skipped frames
//   would have lost every comparison; 1e-4 of relative margin on r^2 covers the rounding of the float evaluation against
//   the exact inequality.
//   The bound: per wave the smallest of the table entries of its angles as they stand at launch (earlier chunks; they
//   only grow), the smallest of its own lane-local running maxima, and the largest bound any wave of the same angle
//   group has published (read first, atomicMax only when it improves), refreshed every 8 steps; the test compares
//   with the smallest of the four waves' bounds.
//   Material on which nearly everything survives (a steady tone; the first steps of the first chunk, when no bound
//   exists yet) would pay for the test and the LDS round trip on top of the full evaluation: when more than 7/8 of a
//   step's frames survive the workgroup feeds the next steps straight from the stage, untested, then refreshes its
//   bounds and tests one step again -- the pause doubles while that keeps failing and resets when it succeeds.  Tiles
//   that went through untested are counted separately (prot_sweep_stats_ex).
#include "prot_dev.h"
#include "prot_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace prot {

typedef float f32x16 __attribute__ ((ext_vector_type (16)));

// max over the 32 lanes of each half of the wave by DPP (quad_perm [1 0 3 2], [2 3 0 1], row_ror 4 and 8, then row_bcast15 into
// the odd rows): five VALU instructions, the result valid in lanes 16-31 and 48-63.  (__shfl_xor compiles to ds_bpermute_b32 and
// a wait each: the 240 of a wave's final 48-value reduction were ~12 us of every sweep launch.)
template <int CTRL, int ROW_MASK = 0xf>
PROT_DEV float dpp_move (float old, float v)
{
	return __uint_as_float ((unsigned)__builtin_amdgcn_update_dpp ((int)__float_as_uint (old), (int)__float_as_uint (v), CTRL, ROW_MASK, 0xf, false));
}
PROT_DEV float half_max_odd_rows (float v)
{
	v = fmaxf (v, dpp_move<0xB1> (v, v));
	v = fmaxf (v, dpp_move<0x4E> (v, v));
	v = fmaxf (v, dpp_move<0x124> (v, v));
	v = fmaxf (v, dpp_move<0x128> (v, v));
	return fmaxf (v, dpp_move<0x142, 0xa> (v, v)); // rows 1 and 3: max (own row, the row before)
}

constexpr int kStage  = 1024;             // frames per step: loaded ONCE per workgroup, shared through LDS
constexpr int kSteps  = 8;                // steps between two looks at the shared bound
constexpr int kBatch  = kStage / 4;       // frames of a step each wave tests (4 groups of 64 = 8 tiles)
constexpr int kRing   = 512;              // (d, h) entries of one wave's survivor ring: < 64 left over + one batch
static_assert (kRing >= 64 + kBatch && (kRing & (kRing - 1)) == 0, "ring too small");
constexpr unsigned kCtr = 32;             // step counters per channel (a cache line each): workgroup b draws from counter b mod kCtr
constexpr int kPause0 = 2, kPauseMax = 256; // steps fed untested after a step on which > 7/8 of the frames survived: first, cap

template <int NBLK, bool PRUNE>
__global__ __launch_bounds__ (256, PRUNE ? 4 : 3) void sweep_kernel (const SweepParams p)
{
	__shared__ float stage_s[2 * 2 * kStage]; // [buffer][d | h][frame]: planar, so that a loaded dwordx4 is parked as it is
	__shared__ float2 ring_s[PRUNE ? 4 * kRing : 1];

	const int tid  = threadIdx.x;
	const int w    = __builtin_amdgcn_readfirstlane (tid >> 6); // wave-uniform by construction
	const int lane = tid & 63;
	const int j    = lane & 31; // column within a tile / angle within a block (A operand)
	const int kk   = lane >> 5; // k index of the rank-2 product: 0 -> (cos, dry), 1 -> (sin, hilbert)
	const int c    = p.chn0 + blockIdx.y;

	const float* __restrict__ x    = p.x + (int64_t)c * p.x_stride;
	const float* __restrict__ hist = p.hist ? p.hist + (int64_t)c * p.hist_len : nullptr;
	const float* __restrict__ u    = p.u + (int64_t)c * p.u_stride;

	const int nb   = (p.n_angles + 31) >> 5;      // angle blocks in this call (<= 12)
	const int G    = nb >= 3 ? 4 : nb;            // wave groups: 4, 2 or 1
	// which group a wave serves rotates with the workgroup: wave w of every resident workgroup sits on SIMD w, and the
	// groups' survivor counts differ (the group that holds the smallest peak evaluates several times more tiles)
	const int wr_  = (w + (int)blockIdx.x) & 3;
	const int g    = wr_ % G;                     // my group -> blocks g, g + G, g + 2G
	const int sub  = wr_ / G;                     // waves of one group split the batches
	const int nsub = 4 / G;
	const bool idle = g >= nb; // 65..96 angles put three blocks on four groups: this wave only helps loading

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
#pragma unroll
	for (int q = 0; q < NBLK; ++q) {
		// the A operands are in their registers from here on: without this use the first s_waitcnt that covers their loads
		// lands INSIDE the main loop, in front of the first MFMA, as vmcnt(0) -- and drains the prefetch of every step
		asm volatile ("" : "+v"(a_op[q]));
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

	// ---- pruning state: wave-uniform.  This file is compiled with -ffast-math (finite-math-only): no infinities as
	// sentinels; every wave that is not idle has at least one valid entry, so kBig never survives its wave_min.
	constexpr float kBig = 1.0e18f; // finite, and its square still is
	float thr2 = 0.f;               // frames with r^2 <= thr2 cannot matter (margin included)
	if (PRUNE && !idle) {
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
			// read first, write only a bound that improves on what is published: every wave of the launch aims at the
			// same cache line (n_chn x 4 floats), and same-line atomics are served one at a time by one L2 channel --
			// with an unconditional atomicMax per wave and iteration that queue, not the stream, set the kernel's time
			float best = lo;
			if (lane == 0) {
				float cur = __uint_as_float (__hip_atomic_load (thr_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
				if (lo > cur) {
					cur = __uint_as_float (atomicMax (thr_slot, __float_as_uint (lo)));
				}
				best = fmaxf (lo, cur);
			}
			lo = uniform_f (best);
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
	if (PRUNE && !idle && thr2 > 0.f) {
		const float b = share (sqrtf (thr2));
		thr2          = uniform_f (fmaxf (thr2, b * b * 0.9999f));
	}

	// ---- survivors: wave w tests ITS quarter of a step's frames -- once per workgroup, against the smallest of the four
	// groups' bounds -- and compacts them into its own ring; after a barrier every wave evaluates the full tiles of all four
	// rings with its own angles (waves of one group split them).  (Every wave testing every frame against its own bound,
	// the first compacting version, spent 0.40 of its 0.50 ms per 115 M channel-frames in the test.)
	float* const ring = reinterpret_cast<float*> (ring_s + (PRUNE ? w * kRing : 0));
	__shared__ unsigned sh_surv[4];    // survivors each wave found in the current step (every wave keeps all four ring cursors itself)
	__shared__ float    sh_thr[4];     // each wave's own bound (squared, margin included; idle waves: huge)
	unsigned            wr = 0;                 // my ring's write cursor (entries, free-running)
	unsigned            cw[4] = { 0u, 0u, 0u, 0u }; // all four rings' cursors as of the last barrier (same in every wave)
	bool                testing = true;
	int                 pause = kPause0, pause_left = 0;
	unsigned            pair_no = 0;   // tile pairs drained so far (same in every wave): pair_no % nsub == sub are mine
	unsigned            n_seen = 0, n_eval = 0, n_untested = 0, n_surv = 0; // tiles, tiles, tiles, frames
	float               thr_min = 0.f; // what the test compares with: the smallest of the four waves' bounds
	if (PRUNE) {
		if (lane == 0) {
			sh_surv[w]  = 0u;
			sh_thr[w]   = idle ? kBig : thr2;
		}
		__syncthreads ();
		thr_min = fminf (fminf (sh_thr[0], sh_thr[1]), fminf (sh_thr[2], sh_thr[3]));
	}

	// ---- the stream: steps of 1024 frames, handed out by a counter -- at any moment the workgroups of a channel read ONE
	// dense window of the stream (what keeps DRAM pages open: with 8192-frame iterations cut into steps the same loads were
	// spread over an 8 x wider window and the bare stream ran at 2.4 TB/s instead of 4.4), and no workgroup is left with
	// steps to do when the others are done: the hardware serves the oldest wave first, so of the four workgroups of a CU
	// the first dispatched runs ahead, and with a fixed share per workgroup (grid-stride) the CUs spent the last third of
	// a launch with fewer and fewer workgroups to overlap.  seq (s) = the step this workgroup does as its s-th: fetched
	// four steps ahead (the loads of step s + 3 go out during step s) by a wave that is not loading in that step, passed
	// on through LDS; the counters are p.ctr[channel][kCtr], zero at launch (the merge kernel of every launch leaves it so).
	// A step's frames are loaded once per workgroup and parked in LDS, where all four waves test them; with every wave
	// loading every frame itself (rounds 1-2) the loads in flight per CU were 4 x redundant.
	__shared__ unsigned sh_seq[8];
	const unsigned  n_total = p.t_end > p.t_begin ? (unsigned)((p.t_end - p.t_begin + kStage - 1) / kStage) : 0u;
	// (kCtr counters per channel, each on a cache line of its own, each handing out every kCtr-th step to the workgroups
	// with that blockIdx.x mod kCtr: ONE counter serves its atomics one at a time -- 675 K of them made cfg4 9 ms instead of 3)
	// (a launch of fewer than kCtr workgroups per channel -- many channels, a small device -- uses as many counters as it has
	// workgroups: a counter nobody draws from would leave its steps undone)
	const unsigned  ngrp = gridDim.x < kCtr ? gridDim.x : kCtr;
	const unsigned  grp  = blockIdx.x % ngrp;
	unsigned* const ctr  = p.ctr + ((size_t)c * kCtr + grp) * 32;
	if (tid == 0) {
		const unsigned b0 = atomicAdd (ctr, 4u);
		sh_seq[0]         = ngrp * b0 + grp;
		sh_seq[1]         = ngrp * (b0 + 1u) + grp;
		sh_seq[2]         = ngrp * (b0 + 2u) + grp;
		sh_seq[3]         = ngrp * (b0 + 3u) + grp;
	}
	__syncthreads ();
	auto seq = [&] (int s) { return (unsigned)__builtin_amdgcn_readfirstlane ((int)sh_seq[s & 7]); };

	// Loading is split by step parity: waves 0-1 bring in the odd steps, waves 2-3 the even ones, each wave half a step
	// (512 frames: a lane four consecutive frames of each stream, twice -- dwordx4 loads).  At the start of step s the
	// waves whose turn it is park the frames of step s + 1 (requested two steps ago) in the stage and request step s + 3:
	// two steps are in flight per workgroup at any time, with one set of registers per wave and one instance of the code
	// (one step in flight, 4 x 8 KB per CU, streamed at 2.4 TB/s with nothing else to do; two register sets and the loop
	// unrolled by two ran the wave-uniform state out of SGPRs).
	const int ld_par  = w >> 1;        // parity of the steps s at whose start this wave parks / requests
	const int ld_half = (w & 1) * 512; // my half of a step
	float4    dv[2], hv[2];
	// the raw peak (angle 0, Q4: abs-max of the un-rotated input, cli/phase-rotate.cc:413-414) rides along: every input frame
	// passes through exactly one wave's registers on its way to the stage (a separate pass over the input was 0.49 ms of
	// cfg4's 3.7).  What the loaders see is x[t - delay] for t in [t_begin, t_end): every frame of the call's input from its
	// first on -- at a stream start t_begin - delay = 0; the last `delay` frames of a chunk that is not the last are seen
	// by the next chunk as history, frames before a shard (primed history) belong to the same file -- beyond it zeros
	float amax = 0.f;
	auto locate = [&] (unsigned gs, int64_t& t0) { // gs: the step's number in the stream (seq of the local step) // first frame of my lane in step s; true: the whole step is plain stream
		const int64_t base = p.t_begin + (int64_t)gs * kStage;
		t0                 = base + ld_half + 4 * lane;
		return (base + kStage <= p.t_end) && (base - p.delay >= 0) && (base + kStage - p.delay <= p.n_x) &&
		       (base + p.t_abs0 >= p.dry_from);
	};
	auto request = [&] (unsigned s) { // frames of step s -> dv, hv (plain stream only; the edges are fetched when they are parked)
		int64_t t0;
		if (locate (s, t0)) {
			typedef float f4u __attribute__ ((ext_vector_type (4), aligned (4)));
#pragma unroll
			for (int i = 0; i < 2; ++i) {
				const f4u d4 = *reinterpret_cast<const f4u*> (x + t0 + 256 * i - p.delay);
				const f4u h4 = *reinterpret_cast<const f4u*> (u + t0 + 256 * i);
				dv[i]        = make_float4 (d4[0], d4[1], d4[2], d4[3]);
				hv[i]        = make_float4 (h4[0], h4[1], h4[2], h4[3]);
			}
		}
	};
	auto park = [&] (int buf, unsigned s) { // frames of step s: dv, hv -> (d, h) pairs of the stage
		int64_t t0;
		if (locate (s, t0)) {
#pragma unroll
			for (int i = 0; i < 2; ++i) {
				float* const q                            = stage_s + buf * 2 * kStage + ld_half + 256 * i + 4 * lane;
				*reinterpret_cast<float4*> (q)            = dv[i];
				*reinterpret_cast<float4*> (q + kStage)   = hv[i];
				amax = fmaxf (fmaxf (amax, fabsf (dv[i].x)), fmaxf (fabsf (dv[i].y), fmaxf (fabsf (dv[i].z), fabsf (dv[i].w))));
			}
		} else {
			// edges of the stream (history, silence, the dry-less first half block): fetched here, frame by frame, latency
			// exposed -- a few iterations per launch; kept out of request() so that its loads land in registers no other
			// path writes (a merge of two producers costs copies, and a copy of a register in flight is a wait)
#pragma unroll
			for (int i = 0; i < 2; ++i) {
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					const int64_t ti  = t0 + 256 * i + e;
					const int64_t ix  = ti - p.delay;
					float         d = 0.f, h = 0.f;
					if (ti < p.t_end) {
						h               = u[ti];
						const float raw = ldv (x, p.n_x, hist, p.hist_len, ix);
						amax            = fmaxf (amax, fabsf (raw)); // (the raw peak sees the frames the first block's dry term skips)
						d               = ti + p.t_abs0 >= p.dry_from ? raw : 0.f;
					}
					float* const q = stage_s + buf * 2 * kStage + ld_half + 256 * i + 4 * lane + e;
					q[0]           = d;
					q[kStage]      = h;
				}
			}
		}
	};
	if (p.raw_slot >= 0 && p.raw_tail && blockIdx.x == 0) {
		// the call's last `delay` input frames: the loaders reach them only as the next chunk's history (x[t - delay], t < t_end),
		// the reference's calc_peak has them in this block -- folded here so that peaks[c][0] is complete when the call returns
		for (int64_t i = p.n_x - p.delay + tid; i < p.n_x; i += 256) {
			amax = fmaxf (amax, fabsf (ldv (x, p.n_x, hist, p.hist_len, i)));
		}
	}
	// prologue: waves 2-3 fetch step 0 and park it (the only exposed latency), then request step 2; waves 0-1 request step 1
	{
		const unsigned g0 = seq (0), g1 = seq (1), g2 = seq (2);
		if (ld_par == 1) {
			if (g0 < n_total) {
				request (g0);
				park (0, g0);
			}
			if (g2 < n_total) {
				request (g2);
			}
		} else if (g1 < n_total) {
			request (g1);
		}
	}
	__syncthreads ();
	for (int s = 0;; ++s) {
		const unsigned gs0 = seq (s), gs1 = seq (s + 1), gs3 = seq (s + 3); // this step, the one to park, the one to request
		if (gs0 >= n_total) {
			break;
		}
		const int buf = s & 1;
		// the step after the ones that are known (s + 4): asked for by the first wave of the pair that does not load now
		const bool asks = tid == ((s & 1) << 7 ^ 128);
		unsigned   got  = 0u;
		if (asks) {
			got = ngrp * atomicAdd (ctr, 1u) + grp;
		}
		if ((s & 1) == ld_par) {
			if (gs1 < n_total) {
				park (buf ^ 1, gs1); // everybody finished reading that half of the stage before the barrier that ended step s - 1
			}
			if (gs3 < n_total) {
				request (gs3);
			}
		}
		const float* const st = stage_s + buf * 2 * kStage; // d at st[f], h at st[kStage + f]
		if (!PRUNE) {
			if (!idle) {
				// (one instance of the evaluation in a plain loop, like the pruned kernel's untested feed: unrolled it needs
				// ~160 VGPRs, three waves per SIMD; at four the other waves cover the matrix core's latency just as well)
#pragma unroll 1
				for (int it = sub; it < kStage / 64; it += nsub) {
					evaluate (st[kk * kStage + it * 64 + j], st[kk * kStage + it * 64 + 32 + j]);
				}
			}
			if (asks) {
				sh_seq[(s + 4) & 7] = got;
			}
			__syncthreads ();
			continue;
		}
		// ---- test my quarter (all four waves, idle or not: an idle wave has no angles, but frames to test)
		if (testing) {
			float2 v[kBatch / 64];
#pragma unroll
			for (int i = 0; i < kBatch / 64; ++i) { // all reads in front of the ring writes (the backend keeps that order)
				v[i] = make_float2 (st[w * kBatch + i * 64 + lane], st[kStage + w * kBatch + i * 64 + lane]);
			}
			unsigned surv = 0;
#pragma unroll
			for (int i = 0; i < kBatch / 64; ++i) {
				const bool     live = v[i].x * v[i].x + v[i].y * v[i].y > thr_min;
				const uint64_t mask = __builtin_amdgcn_ballot_w64 (live);
				if (live) {
					const unsigned at = (wr + __builtin_amdgcn_mbcnt_hi ((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo ((unsigned)mask, 0u))) & (kRing - 1);
					*reinterpret_cast<float2*> (ring + 2 * at) = v[i];
				}
				const unsigned nl = (unsigned)__builtin_popcountll (mask);
				wr += nl;
				surv += nl;
			}
			n_surv += surv;
			if (lane == 0) {
				sh_surv[w] = surv;
			}
		}
		__syncthreads (); // the rings and the survivor counts are complete for this step
		// ---- evaluate: the full tiles of the four rings, or -- test switched off -- all 32 tiles of the step from the stage
		unsigned sv_step = 0;
		{
			unsigned r0v[4], cnt[4], total = 0, sv = 0; // first entry and number of tile pairs to take from each ring (wave-uniform)
#pragma unroll
			for (int v4 = 0; v4 < 4; ++v4) {
				const unsigned add = testing ? (unsigned)__builtin_amdgcn_readfirstlane ((int)sh_surv[v4]) : 0u;
				r0v[v4]            = cw[v4] & ~63u; // drained up to a multiple of 64 last time
				cw[v4] += add;
				cnt[v4] = (cw[v4] - r0v[v4]) >> 6;
				cnt[v4] = testing ? cnt[v4] : 0u;
				total += cnt[v4];
				sv += add;
			}
			if (!testing) {
				total = kStage / 64;
			}
			const unsigned b1 = cnt[0], b2 = b1 + cnt[1], b3 = b2 + cnt[2];
			const float* const rings = reinterpret_cast<const float*> (ring_s);
			for (unsigned it = 0; it < total; ++it) {
				const bool mine = !idle && (pair_no + it) % (unsigned)nsub == (unsigned)sub;
				if (mine) {
					float b0, b1v;
					if (testing) {
						const unsigned v4 = (it >= b1) + (it >= b2) + (it >= b3);
						const unsigned lo = v4 == 0 ? 0u : v4 == 1 ? b1 : v4 == 2 ? b2 : b3;
						const unsigned e0 = (v4 == 0 ? r0v[0] : v4 == 1 ? r0v[1] : v4 == 2 ? r0v[2] : r0v[3]) + 64u * (it - lo);
						const float* const rg = rings + 2 * kRing * v4;
						b0  = rg[2 * ((e0 + j) & (kRing - 1)) + kk];
						b1v = rg[2 * ((e0 + 32 + j) & (kRing - 1)) + kk];
					} else {
						b0  = st[kk * kStage + it * 64 + j];
						b1v = st[kk * kStage + it * 64 + 32 + j];
					}
					evaluate (b0, b1v);
					n_eval += 2;
				}
			}
			pair_no += total;
			sv_step = sv;
			if (!idle) {
				// (statistics in the units of the unpruned kernel: tiles this wave's share of the step would have cost)
				n_seen += (kStage / 32) / nsub;
				n_untested += testing ? 0 : (kStage / 32) / nsub;
			}
		}
		// ---- the switch (workgroup-uniform: every wave sees the same counts), and the bound
		bool look = (s + 1) % kSteps == 0; // time to look at the bounds again
		if (testing) {
			if (sv_step * 8 > (unsigned)kStage * 7) {
				testing    = false;
				pause_left = pause;
				pause      = pause * 2 < kPauseMax ? pause * 2 : kPauseMax;
			} else {
				pause = kPause0;
			}
		} else if (--pause_left <= 0) {
			testing = true;
			look    = true;
		}
		if (look) {
			if (!idle) {
				refresh ();
			}
			if (lane == 0) {
				sh_thr[w] = idle ? kBig : thr2;
			}
		}
		if (asks) {
			sh_seq[(s + 4) & 7] = got; // (not one of the three slots this step reads: s, s + 1, s + 3)
		}
		__syncthreads (); // everybody is done with this half of the stage and with the rings' drained entries
		if (look) {
			thr_min = fminf (fminf (sh_thr[0], sh_thr[1]), fminf (sh_thr[2], sh_thr[3]));
		}
	}
	if (PRUNE && !idle) {
		// what is left in the four rings (< 64 entries each), padded with zeros
		const float* const rings = reinterpret_cast<const float*> (ring_s);
		for (unsigned v4 = 0; v4 < 4u; ++v4) {
			const unsigned c1 = v4 == 0 ? cw[0] : v4 == 1 ? cw[1] : v4 == 2 ? cw[2] : cw[3], e0 = c1 & ~63u, left = c1 - e0;
			if (left > 0u && (pair_no + v4) % (unsigned)nsub == (unsigned)sub) {
				const float* const rg = rings + 2 * kRing * v4;
				float b0 = rg[2 * ((e0 + j) & (kRing - 1)) + kk];
				float b1 = rg[2 * ((e0 + 32 + j) & (kRing - 1)) + kk];
				b0       = (unsigned)j < left ? b0 : 0.f;
				b1       = (unsigned)j + 32u < left ? b1 : 0.f;
				evaluate (b0, b1);
				n_eval += left > 32u ? 2 : 1;
			}
		}
	}
	// statistics: one set of atomics per workgroup, not per wave (every atomic of the launch lands on the same cache line)
	if (PRUNE && p.stats != nullptr) {
		unsigned* const acc = reinterpret_cast<unsigned*> (stage_s); // the stage is free: everybody passed the last barrier
		if (tid < 4) {
			acc[tid] = 0u;
		}
		__syncthreads ();
		if (lane == 0 && !idle) {
			atomicAdd (&acc[0], n_eval);
			atomicAdd (&acc[1], n_seen);
			atomicAdd (&acc[2], n_untested);
			atomicAdd (&acc[3], n_surv);
		}
		__syncthreads ();
		if (tid < 4) {
			atomicAdd (&p.stats[tid], (unsigned long long)acc[tid]);
		}
	}
	// reduce over the 32 columns (lanes of one half); one value per angle and wave goes to the launch's scratch table
	// [channel][workgroup][wave's place g + G sub][angle slot | raw] with a plain store (a wave fills the slots of its own
	// angle blocks, and the raw slot), and sweep_merge_kernel folds that into the peak table.  (One atomicMax per wave and
	// angle on the table itself -- 196 K atomics on 45 cache lines at the end of a full launch -- cost 100-200 us per
	// launch however short; looking first made each wave walk 48 dependent L2 round trips instead.)
	float* const part = p.part + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wr_) * (p.n_angles + 1);
	{
		float v = amax;
		v       = fmaxf (v, __shfl_xor (v, 1));
		v       = fmaxf (v, __shfl_xor (v, 2));
		v       = fmaxf (v, __shfl_xor (v, 4));
		v       = fmaxf (v, __shfl_xor (v, 8));
		v       = fmaxf (v, __shfl_xor (v, 16));
		v       = fmaxf (v, __shfl_xor (v, 32));
		if (lane == 0) {
			part[p.n_angles] = v;
		}
	}
	if (idle) {
		return;
	}
	// (the lane's coordinates made opaque: the compiler otherwise carries what the prologue derived from them through the
	// whole kernel for this one use -- seven VGPRs spilled at the 128 of four waves per SIMD)
	int kk_e = kk, j_e = j;
	asm volatile ("" : "+v"(kk_e), "+v"(j_e));
#pragma unroll
	for (int q = 0; q < NBLK; ++q) {
#pragma unroll
		for (int r = 0; r < 16; ++r) {
			const float v   = half_max_odd_rows (m[q][r]); // (valid in lanes 16-31, 48-63)
			const int   row = (r & 3) + 8 * (r >> 2) + 4 * kk_e;
			const int   idx = (g + q * G) * 32 + row;
			if (j_e == 16 && g + q * G < nb && idx < p.n_angles) {
				part[idx] = v;
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// sweep_eval_kernel (round 4): the matrix-core evaluation of the LISTS the fused Hilbert pass left (olasave_kernel MODE_SWEEP:
// the (d, h) pairs of the frames whose r exceeds the smallest peak of the whole table).  A listed frame matters to few angles:
// |ca d + sa h| = r |cos (a - phi)| (phi = the frame's phase), so against a BLOCK of 32 angles that all lie within D of psi
// (modulo pi) it can reach at most r cos (max (0, delta - D)), delta = the distance of phi from psi folded into [0, pi / 2]:
// with u = |d cos psi + h sin psi| = r |cos (phi - psi)| and v = |h cos psi - d sin psi| = r |sin (phi - psi)| that is r when
// v <= u tan D, and u cos D + v sin D otherwise -- ten VALU operations per frame and block.  Every wave tests the listed frames
// against each of its (up to three) blocks' own bounds -- the smallest table entry of the block's angles at launch, raised by
// what the block's running maxima and the other workgroups (p.thr_blk) prove since -- and compacts the survivors into a ring
// per block; the matrix core sees full tile pairs of one block's survivors.  On the 60-minute material 3-6 % of the (listed
// frame, block) pairs survive (harmonic stack 4.6 %, stationary noise 3.4 %: a frame barely above the smallest peak of the
// table is below the bound of most blocks, and above it only near its own phase).
// Exact for the same reason as the list test: a (frame, block) pair that fails could not have raised a maximum of the block;
// the margins (D widened by 2e-3 rad, 2e-6 (u + v) added to the reach, 1e-4 relative on the bounds) cover the float rounding of the test, of the
// table's sines and cosines and of the product itself (each ~1e-7 r).  A block whose angles span the half circle (a coarse pass: every 24th
// angle) has D = pi / 2: the test is r > bound.
constexpr int kERing = 128; // (d, h) entries of one block's ring: < 64 left over + 64 new

#ifdef PROT_EVAL_STAMPS
// diagnostic build (tools/ab_build.py sw:<tag>:-DPROT_EVAL_STAMPS, tools/attic/eval_stamps.py): where a workgroup of sweep_eval_kernel spends
// its time -- thread 0 stamps the 100 MHz clock at entry, after the prologue, the first claim, the first stage, each of its first
// twelve steps, the loop's end, the rings' rest and the end
__device__ unsigned long long g_eval_stamps[4096 * 24];
#define EVAL_STAMP(k)                                                                                              \
	do {                                                                                                           \
		if (threadIdx.x == 0) {                                                                                    \
			g_eval_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) % 4096 * 24 + (k)] = wall_clock64 ();       \
		}                                                                                                          \
	} while (0)
#else
#define EVAL_STAMP(k) \
	do {              \
	} while (0)
#endif

template <int NBLK>
__global__ __launch_bounds__ (256, 4) void sweep_eval_kernel (const SweepParams p)
{
	__shared__ float    stage_s[2 * 2 * kStage];
	__shared__ float2   ring_s[4 * NBLK * kERing];
	__shared__ unsigned sh_seq[8];
	__shared__ int      sh_cnt[2]; // entries of the step parked in each half of the stage

	const int tid  = threadIdx.x;
	const int w    = __builtin_amdgcn_readfirstlane (tid >> 6);
	const int lane = tid & 63;
	const int j    = lane & 31;
	const int kk   = lane >> 5;
	const int c    = p.chn0 + blockIdx.y;

	EVAL_STAMP (0);
	const int  nb   = (p.n_angles + 31) >> 5;
	const int  G    = nb >= 3 ? 4 : nb;
	const int  wr_  = (w + (int)blockIdx.x) & 3; // (as in sweep_kernel: the place in the scratch table the merge kernel expects)
	const int  g    = wr_ % G;
	const int  sub  = wr_ / G;
	const int  nsub = 4 / G;
	const bool idle = g >= nb;

	auto uniform_f = [] (float v) { return __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (v))); };
	auto wave_min  = [] (float v) {
		v = fminf (v, __shfl_xor (v, 1));
		v = fminf (v, __shfl_xor (v, 2));
		v = fminf (v, __shfl_xor (v, 4));
		v = fminf (v, __shfl_xor (v, 8));
		v = fminf (v, __shfl_xor (v, 16));
		v = fminf (v, __shfl_xor (v, 32));
		return v;
	};
	constexpr float kBig = 1.0e18f;

	float a_op[NBLK];
	float cps[NBLK], sps[NBLK], cdl[NBLK], sdl[NBLK], tdl[NBLK]; // the blocks' sectors (wave-uniform)
	float bnd[NBLK], bnd2[NBLK];                                 // the blocks' bounds, margin included (wave-uniform)
	unsigned* slot[NBLK];
#pragma unroll
	for (int q = 0; q < NBLK; ++q) {
		const int  blk = g + q * G;
		const int  idx = blk * 32 + j;
		const bool ok  = blk < nb && idx < p.n_angles;
		float      v = 0.f, lo = kBig;
		if (ok) {
			const int a = p.angles[idx];
			v           = p.lut[2 * a + (kk ? 0 : 1)]; // lut = {sin, cos}
			lo          = __builtin_nontemporal_load (&p.peaks[(int64_t)c * kMaxSample + a]);
		}
		a_op[q]      = v;
		lo           = wave_min (lo); // (a block without angles keeps kBig: nothing passes its test)
		bnd[q]       = uniform_f (lo * 0.99995f);
		bnd2[q]      = uniform_f (lo * lo * 0.9999f);
		const int bs = blk < 12 ? blk : 0;
		cps[q]       = p.sect[bs][0];
		sps[q]       = p.sect[bs][1];
		cdl[q]       = p.sect[bs][2] + 2e-6f;
		sdl[q]       = p.sect[bs][3] + 2e-6f;
		tdl[q]       = p.sect[bs][2] > 1e-6f ? p.sect[bs][3] / p.sect[bs][2] : 1e30f;
		slot[q]      = reinterpret_cast<unsigned*> (p.thr_blk) + ((int64_t)c * 16 + bs);
	}
#pragma unroll
	for (int q = 0; q < NBLK; ++q) {
		asm volatile ("" : "+v"(a_op[q])); // (see sweep_kernel: keeps the first s_waitcnt of these loads out of the loop)
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

	// what the block's running maxima prove by now, shared with every workgroup that serves the block (read first, atomicMax only
	// when it improves: sweep_kernel's rule)
	// (a lane's smallest maximum among its 16 rows bounds the smallest final peak of those rows from below, so does the LARGEST such
	// value among the 32 lanes of the half -- one loud frame raises all 32 angles of a block to within cos 8 deg of its r, so this is
	// nearly the min over rows of the max over lanes, for 16 v_min and 5 DPP instead of 80 reductions --; the block's bound is the
	// smaller of the two halves')
	// Round 5: ALL of a wave's blocks at once -- lane q looks up block q's shared bound and, when the wave has something better,
	// raises it: two round trips at most.  One block after the other, each with a load and then an atomic whose result it needed,
	// a refresh was three to six dependent round trips: 5-6 us, four times in a shard's eight steps -- with the refreshes compiled
	// out the evaluation of a shard took 59 us instead of 82, the whole file's 243 instead of 292, stationary noise the same
	// (tools/ab_build.py sw:noref:-DPROT_EVAL_NO_REFRESH; the whole pass on four materials, ms: harmonic 1.42 -> 1.38, noise 2.22 ->
	// 2.21, a level rising in eight steps 1.41 -> 1.41, one loud passage the sample missed 2.22 -> 2.25): the lists were made against the
	// seeding pass's bound, which is what the table holds at this launch's start, and the maxima found while evaluating them add
	// little to it.  So: a refresh behind every 16th step (a workgroup with that many steps is in a long evaluation, where bounds
	// that rise during it are all that prunes), and behind the first step of a launch that starts without a bound.
	auto refresh_all = [&] () -> bool {
		float lo3[NBLK];
#pragma unroll
		for (int q = 0; q < NBLK; ++q) {
			const int blk = g + q * G;
			float     lo  = kBig;
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
				if (blk < nb && blk * 32 + row < p.n_angles) {
					lo = fminf (lo, m[q][r]);
				}
			}
			lo            = lo >= kBig ? -1.f : lo; // (a half without rows must not win the max)
			const float v = half_max_odd_rows (lo);
			const float h0 = __uint_as_float ((unsigned)__builtin_amdgcn_readlane ((int)__float_as_uint (v), 16));
			const float h1 = __uint_as_float ((unsigned)__builtin_amdgcn_readlane ((int)__float_as_uint (v), 48));
			lo3[q]         = h0 < 0.f ? h1 : h1 < 0.f ? h0 : fminf (h0, h1);
		}
		float     mine = -1.f;
		unsigned* sl   = slot[0];
#pragma unroll
		for (int q = 0; q < NBLK; ++q) {
			if (lane == q) {
				mine = lo3[q];
				sl   = slot[q];
			}
		}
		float cur = 0.f;
		if (lane < NBLK && mine > 0.f) {
			cur = __uint_as_float (__hip_atomic_load (sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
			if (mine > cur) {
				cur = __uint_as_float (atomicMax (sl, __float_as_uint (mine))); // (non-negative floats order as their bit patterns)
			}
		}
		bool rose = false;
#pragma unroll
		for (int q = 0; q < NBLK; ++q) {
			const float other = __uint_as_float ((unsigned)__builtin_amdgcn_readlane ((int)__float_as_uint (cur), q));
			if (lo3[q] > 0.f) { // (wave-uniform)
				const float best = fmaxf (lo3[q], other) * 0.99995f;
				rose             = rose || best > bnd[q] * 1.02f;
				bnd[q]           = fmaxf (bnd[q], best);
				bnd2[q]          = fmaxf (bnd2[q], best * best * (0.9999f / (0.99995f * 0.99995f)));
			}
		}
		return rose;
	};
	bool unseeded = false; // a block of this wave starts without a bound (a call too short for a seeding pass)
#pragma unroll
	for (int q = 0; q < NBLK; ++q) {
		unseeded = unseeded || (g + q * G < nb && bnd[q] <= 0.f);
	}

	// ---- the steps of the lists: handed out by counters, loaded once per workgroup, parked planar in the stage (sweep_kernel's scheme)
	const unsigned  n_total = (unsigned)__builtin_amdgcn_readfirstlane ((int)p.n_steps[blockIdx.y]);
	const unsigned  ngrp    = gridDim.x < kCtr ? gridDim.x : kCtr;
	const unsigned  grp     = blockIdx.x % ngrp;
	unsigned* const ctr     = p.ctr + ((size_t)c * kCtr + grp) * 32;
	EVAL_STAMP (1);
	if (tid == 0) {
		const unsigned b0 = atomicAdd (ctr, 4u);
		sh_seq[0]         = ngrp * b0 + grp;
		sh_seq[1]         = ngrp * (b0 + 1u) + grp;
		sh_seq[2]         = ngrp * (b0 + 2u) + grp;
		sh_seq[3]         = ngrp * (b0 + 3u) + grp;
	}
	__syncthreads ();
	EVAL_STAMP (2);
	auto seq = [&] (int s) { return (unsigned)__builtin_amdgcn_readfirstlane ((int)sh_seq[s & 7]); };
	const int          ld_par  = w >> 1;
	const int          ld_half = (w & 1) * 512;
	float4             dv[2], hv[2];
	const uint2* const dsc = p.desc + (int64_t)blockIdx.y * p.desc_stride;
	auto request = [&] (unsigned s) { // entries [first, first + n) of the lists; a lane takes four consecutive (d, h) pairs twice
		const uint2         ds = dsc[s];
		const float4* const q  = reinterpret_cast<const float4*> (p.surv + (int64_t)ds.x + ld_half + 4 * lane);
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			if (ld_half + 256 * i + 4 * lane < (int)ds.y) { // (a short step -- the plan cuts them to 128 entries when there are few: park zeroes the rest)
				const float4 a = q[128 * i], b = q[128 * i + 1];
				dv[i]          = make_float4 (a.x, a.z, b.x, b.z);
				hv[i]          = make_float4 (a.y, a.w, b.y, b.w);
			}
		}
	};
	auto park = [&] (int buf, unsigned s) { // (what lies beyond n -- the rest of a list's last step -- is allocated memory: zeroed here)
		const int n = (int)dsc[s].y;
		if (tid == ld_par * 128) {
			sh_cnt[buf] = n; // (read at the top of the step: a load from the step table there stood exposed once per step)
		}
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			const int    e0 = ld_half + 256 * i + 4 * lane;
			float* const q  = stage_s + buf * 2 * kStage + e0;
			float4       d = dv[i], h = hv[i];
			if (e0 + 4 > n) {
				d = make_float4 (e0 < n ? d.x : 0.f, e0 + 1 < n ? d.y : 0.f, e0 + 2 < n ? d.z : 0.f, 0.f);
				h = make_float4 (e0 < n ? h.x : 0.f, e0 + 1 < n ? h.y : 0.f, e0 + 2 < n ? h.z : 0.f, 0.f);
			}
			*reinterpret_cast<float4*> (q)          = d;
			*reinterpret_cast<float4*> (q + kStage) = h;
		}
	};
	{
		const unsigned g0 = seq (0), g1 = seq (1), g2 = seq (2);
		if (ld_par == 1) {
			if (g0 < n_total) {
				request (g0);
				park (0, g0);
			}
			if (g2 < n_total) {
				request (g2);
			}
		} else if (g1 < n_total) {
			request (g1);
		}
	}
	__syncthreads ();
	EVAL_STAMP (3);

	float2* const ring0 = ring_s + w * NBLK * kERing;
	unsigned      wrc[NBLK], rdc[NBLK];
#pragma unroll
	for (int q = 0; q < NBLK; ++q) {
		wrc[q] = rdc[q] = 0u;
	}
	unsigned n_seen = 0, n_eval = 0, n_surv = 0;
	for (int s = 0;; ++s) {
		const unsigned gs0 = seq (s), gs1 = seq (s + 1), gs3 = seq (s + 3);
		if (gs0 >= n_total) {
			break;
		}
		const int  buf  = s & 1;
		const bool asks = tid == ((s & 1) << 7 ^ 128);
		unsigned   got  = 0u;
		if (asks) {
			got = ngrp * atomicAdd (ctr, 1u) + grp;
		}
		if ((s & 1) == ld_par) {
			if (gs1 < n_total) {
				park (buf ^ 1, gs1);
			}
			if (gs3 < n_total) {
				request (gs3);
			}
		}
		if (!idle) {
			const float* const st = stage_s + buf * 2 * kStage;
			const int          nch = (__builtin_amdgcn_readfirstlane (sh_cnt[buf]) + 63) >> 6; // (a list's last step is short: the chunks beyond its entries are zeros)
			// (fetching the next chunk's frames from the stage while this one is tested was measured: 1.5-3 % slower -- the other
			// waves of the SIMD cover the LDS round trip already; tools/attic/sweep_ab_libs.py)
			// Two chunks of 64 frames per turn, and ALL their tests -- 2 x NBLK sector tests of ten vector operations, three compares each
			// as lane masks combined on the scalar unit -- before the first branch on any of them (round 5: with a test, a branch and a
			// compaction per block in turn, every block's compare -> scalar -> branch chain stood exposed behind the one before: 10 us
			// per 1024-frame step where the vector operations account for 4; tools/attic/eval_stamps.py).  (Written with && / || the compiler
			// nests two EXEC-masked regions per block, ~20 scalar instructions of bookkeeping around ten vector ones; the compaction is
			// skipped when nothing of the chunk survives.)
#pragma unroll 1
			for (int it = sub; it < nch; it += 2 * nsub) {
				const int  it2 = it + nsub;
				const bool two = it2 < nch;
				float      dd[2], hh[2];
				dd[0] = st[it * 64 + lane];
				hh[0] = st[kStage + it * 64 + lane];
				dd[1] = 0.f;
				hh[1] = 0.f;
				if (two) {
					dd[1] = st[it2 * 64 + lane];
					hh[1] = st[kStage + it2 * 64 + lane];
				}
				uint64_t mk[2][NBLK];
#pragma unroll
				for (int cn = 0; cn < 2; ++cn) {
					const float d = dd[cn], h = hh[cn];
					const float r2 = d * d + h * h;
#pragma unroll
					for (int q = 0; q < NBLK; ++q) {
						const float    u  = fabsf (d * cps[q] + h * sps[q]);
						const float    v  = fabsf (h * cps[q] - d * sps[q]);
						const float    t  = u * cdl[q] + v * sdl[q];
						const uint64_t m1 = __builtin_amdgcn_ballot_w64 (r2 > bnd2[q]);
						const uint64_t mi = __builtin_amdgcn_ballot_w64 (v <= u * tdl[q]);
						const uint64_t m2 = __builtin_amdgcn_ballot_w64 (t > bnd[q]);
						mk[cn][q]         = m1 & (mi | m2);
					}
				}
#pragma unroll
				for (int cn = 0; cn < 2; ++cn) {
					const float d = dd[cn], h = hh[cn];
#pragma unroll
					for (int q = 0; q < NBLK; ++q) {
						const uint64_t mask = mk[cn][q];
						float2* const  ring = ring0 + q * kERing;
						if (mask != 0) {
							if ((mask >> lane) & 1) {
								const unsigned at = (wrc[q] + __builtin_amdgcn_mbcnt_hi ((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo ((unsigned)mask, 0u))) & (kERing - 1);
								ring[at]          = make_float2 (d, h);
							}
							const unsigned nl = (unsigned)__builtin_popcountll (mask);
							wrc[q] += nl;
							n_surv += nl;
						}
						if (wrc[q] - rdc[q] >= 64u) { // a full tile pair of this block's survivors (wave-uniform)
							wave_sync ();
							const float* const rg = reinterpret_cast<const float*> (ring);
							const float        b0 = rg[2 * ((rdc[q] + j) & (kERing - 1)) + kk];
							const float        b1 = rg[2 * ((rdc[q] + 32 + j) & (kERing - 1)) + kk];
							const f32x16       r0 = __builtin_amdgcn_mfma_f32_32x32x2f32 (a_op[q], b0, zero, 0, 0, 0);
							const f32x16       r1 = __builtin_amdgcn_mfma_f32_32x32x2f32 (a_op[q], b1, zero, 0, 0, 0);
#pragma unroll
							for (int r = 0; r < 16; ++r) {
								m[q][r] = fmaxf (m[q][r], fmaxf (fabsf (r0[r]), fabsf (r1[r])));
							}
							rdc[q] += 64u;
							n_eval += 2;
							wave_sync ();
						}
					}
				}
			}
			n_seen += (unsigned)(2 * nch) / nsub;
			// the bounds: behind every 16th step; a launch that starts without a bound also behind its steps 1, 2, 4, 8 (round 4's schedule)
#ifdef PROT_EVAL_NO_REFRESH // (timing experiment: the bounds of the launch's start only -- still exact, more survivors)
			if (false) {
#else
			if ((s & 15) == 15 || (unseeded && ((s + 1) & s) == 0)) {
#endif
				(void)refresh_all ();
			}
		}
		if (asks) {
			sh_seq[(s + 4) & 7] = got;
		}
		__syncthreads ();
		if (s < 12) {
			EVAL_STAMP (4 + s);
		}
#ifdef PROT_EVAL_STAMPS
		if (threadIdx.x == 0) {
			g_eval_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) % 4096 * 24 + 19] = (unsigned long long)(s + 1);
		}
#endif
	}
	EVAL_STAMP (16);
	if (!idle) {
		// what is left in the rings (< 64 entries each), padded with zeros
#pragma unroll
		for (int q = 0; q < NBLK; ++q) {
			const unsigned left = wrc[q] - rdc[q];
			if (left > 0u) {
				wave_sync ();
				const float* const rg = reinterpret_cast<const float*> (ring0 + q * kERing);
				float              b0 = rg[2 * ((rdc[q] + j) & (kERing - 1)) + kk];
				float              b1 = rg[2 * ((rdc[q] + 32 + j) & (kERing - 1)) + kk];
				b0                    = (unsigned)j < left ? b0 : 0.f;
				b1                    = (unsigned)j + 32u < left ? b1 : 0.f;
				const f32x16 r0 = __builtin_amdgcn_mfma_f32_32x32x2f32 (a_op[q], b0, zero, 0, 0, 0);
				const f32x16 r1 = __builtin_amdgcn_mfma_f32_32x32x2f32 (a_op[q], b1, zero, 0, 0, 0);
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					m[q][r] = fmaxf (m[q][r], fmaxf (fabsf (r0[r]), fabsf (r1[r])));
				}
				n_eval += left > 32u ? 2 : 1;
			}
		}
	}
	EVAL_STAMP (17);
	// statistics in sweep_kernel's units (tiles of a wave's share, all its blocks at once): a tile of ONE block counts 1 / NBLK
	if (p.stats != nullptr) {
		unsigned* const acc = reinterpret_cast<unsigned*> (stage_s);
		__syncthreads ();
		if (tid < 4) {
			acc[tid] = 0u;
		}
		__syncthreads ();
		if (lane == 0 && !idle) {
			atomicAdd (&acc[0], n_eval / (unsigned)NBLK);
			atomicAdd (&acc[1], n_seen);
			atomicAdd (&acc[3], n_surv / (unsigned)NBLK);
		}
		__syncthreads ();
		if (tid < 4) {
			atomicAdd (&p.stats[tid], (unsigned long long)acc[tid]);
		}
	}
	float* const part = p.part + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wr_) * (p.n_angles + 1);
	if (lane == 0) {
		part[p.n_angles] = 0.f; // (the raw peak went into the table from the fused pass)
	}
	if (idle) {
		return;
	}
	int kk_e = kk, j_e = j;
	asm volatile ("" : "+v"(kk_e), "+v"(j_e));
#pragma unroll
	for (int q = 0; q < NBLK; ++q) {
#pragma unroll
		for (int r = 0; r < 16; ++r) {
			const float v   = half_max_odd_rows (m[q][r]); // (valid in lanes 16-31, 48-63)
			const int   row = (r & 3) + 8 * (r >> 2) + 4 * kk_e;
			const int   idx = (g + q * G) * 32 + row;
			if (j_e == 16 && g + q * G < nb && idx < p.n_angles) {
				part[idx] = v;
			}
		}
	}
	EVAL_STAMP (18);
}

#ifdef PROT_EVAL_STAMPS
extern "C" int prot_debug_eval_stamps (unsigned long long* out, int clear)
{
	if (clear) {
		static unsigned long long z[4096 * 24];
		return (int)hipMemcpyToSymbol (HIP_SYMBOL (g_eval_stamps), z, sizeof (z));
	}
	return (int)hipMemcpyFromSymbol (out, HIP_SYMBOL (g_eval_stamps), sizeof (unsigned long long) * 4096 * 24);
}
#endif

// peaks[c][angle] = max (peaks[c][angle], max over the launch's workgroups and the waves that served the angle's block):
// one thread per (angle slot, slice of workgroups), up to 64 slices -> at most 64 atomics per table entry
__global__ __launch_bounds__ (64) void sweep_merge_kernel (const float* __restrict__ part, int n_wg, int n_angles, const int* __restrict__ angles,
                                                             float* peaks, int chn0, int raw_slot, float* thr, unsigned* ctr, float* thr_blk)
{
	if (blockIdx.x == 0 && blockIdx.z == 0) { // what the launch that just ended shared: zero for the next one
		if (thr != nullptr && threadIdx.x < 4) {
			thr[(chn0 + blockIdx.y) * 4 + threadIdx.x] = 0.f;
		}
		if (thr_blk != nullptr && threadIdx.x < 16) {
			thr_blk[(chn0 + blockIdx.y) * 16 + threadIdx.x] = 0.f;
		}
		if (threadIdx.x < kCtr) {
			ctr[((size_t)(chn0 + blockIdx.y) * kCtr + threadIdx.x) * 32] = 0u;
		}
	}
	const int idx = blockIdx.x * 64 + threadIdx.x;
	if (idx > n_angles || (idx == n_angles && raw_slot < 0)) {
		return;
	}
	const int nb = (n_angles + 31) >> 5, G = nb >= 3 ? 4 : nb, nsub = 4 / G, g = (idx >> 5) % G;
	const float* const q = part + (int64_t)blockIdx.y * n_wg * 4 * (n_angles + 1) + idx;
	float              v = 0.f;
	if (idx == n_angles) { // the raw peak: every wave of every workgroup has seen some of the frames
		for (int b = blockIdx.z; b < n_wg; b += gridDim.z) {
			for (int wv = 0; wv < 4; ++wv) {
				v = fmaxf (v, q[((int64_t)b * 4 + wv) * (n_angles + 1)]);
			}
		}
	} else {
		for (int b = blockIdx.z; b < n_wg; b += gridDim.z) {
			for (int sub = 0; sub < nsub; ++sub) {
				v = fmaxf (v, q[((int64_t)b * 4 + g + G * sub) * (n_angles + 1)]);
			}
		}
	}
	unsigned* const slot = reinterpret_cast<unsigned*> (peaks + (int64_t)(chn0 + blockIdx.y) * kMaxSample + (idx == n_angles ? raw_slot : angles[idx]));
	if (__float_as_uint (v) > *slot) {
		atomicMax (slot, __float_as_uint (v));
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
	static const kern_t kern_eval[4] = { nullptr, sweep_eval_kernel<1>, sweep_eval_kernel<2>, sweep_eval_kernel<3> };
	const bool compact = p.surv != nullptr;
	if (compact && (!p.prune || p.desc == nullptr || p.n_steps == nullptr || p.thr_blk == nullptr || p.angles_host == nullptr)) {
		return hipErrorInvalidValue;
	}
	const int    pr = compact ? 2 : p.prune ? 1 : 0;
	const kern_t fn = compact ? kern_eval[nblk] : kern[p.prune ? 1 : 0][nblk];
	float        sect[12][4];
	if (compact) {
		// the sector of each block of 32 requested angles: LUT index a is the angle -a pi / 360 (SinCosLut, cli/phase-rotate.cc:46),
		// the table's 360 entries cover half a circle -- all that matters to |ca d + sa h|; a block whose indices wrap (..., 359, 0,
		// ...) gets the whole half circle, which is merely not tight
		for (int b = 0; b < 12; ++b) {
			int lo = 1 << 30, hi = -1;
			for (int k = 32 * b; k < 32 * b + 32 && k < p.n_angles; ++k) {
				lo = std::min (lo, p.angles_host[k]);
				hi = std::max (hi, p.angles_host[k]);
			}
			const double psi = hi < 0 ? 0.0 : -0.5 * (lo + hi) * M_PI / 360.0;
			const double del = hi < 0 ? 0.0 : std::min (0.5 * M_PI, 0.5 * (hi - lo) * M_PI / 360.0 + 2e-3);
			sect[b][0]       = (float)std::cos (psi);
			sect[b][1]       = (float)std::sin (psi);
			sect[b][2]       = del >= 0.5 * M_PI ? 0.f : (float)std::cos (del);
			sect[b][3]       = del >= 0.5 * M_PI ? 1.f : (float)std::sin (del);
		}
	}
	// persistent grid: exactly the workgroups the chip keeps resident, so the grid-stride loop ends without a ragged
	// last round
	static int slots[3][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
	if (slots[pr][nblk] == 0) {
		int dev = 0, cus = 256, occ = 2;
		(void)hipGetDevice (&dev);
		(void)hipDeviceGetAttribute (&cus, hipDeviceAttributeMultiprocessorCount, dev);
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor (&occ, (const void*)fn, 256, 0) != hipSuccess || occ < 1) {
			occ = 2;
		}
		slots[pr][nblk] = cus * occ;
	}
	const int64_t span  = kStage;
	// (lists: the number of steps is on the device; t_end - t_begin is the host's upper bound of the entries -- every frame)
	const int64_t tiles = (p.t_end - p.t_begin + span - 1) / span;
	if (p.part == nullptr) {
		return hipErrorInvalidValue;
	}
	// channels beyond what the scratch table holds at one workgroup each go in further launches (the table is sized for
	// every resident-workgroup slot: sweep_scratch_floats ())
	const int64_t per_wg   = 4 * ((int64_t)p.n_angles + 1);
	const int64_t chn_most = std::min<int64_t> ((int64_t)(p.part_floats / (size_t)per_wg), 65535);
	if (chn_most < 1) {
		return hipErrorInvalidValue;
	}
	for (int c0 = 0; c0 < n_chn; c0 += (int)chn_most) {
		const int     nc = (int)std::min<int64_t> (chn_most, n_chn - c0);
		// (lists: a workgroup learns its blocks' bounds from its first steps -- give each at least four when there are few)
		const int64_t want = compact ? std::max<int64_t> (1, tiles / 4) : tiles;
		const int64_t gx = std::min<int64_t> (std::min<int64_t> (want, std::max (1, slots[pr][nblk] / nc)), chn_most / nc);
		SweepParams   q  = p;
		q.chn0           = p.chn0 + c0;
		if (compact) {
			memcpy (q.sect, sect, sizeof (sect));
		}
		const dim3 grid ((unsigned)gx, (unsigned)nc), block (256);
		hipLaunchKernelGGL (fn, grid, block, 0, s, q);
		hipError_t err = hipGetLastError ();
		if (err != hipSuccess) {
			return err;
		}
		const dim3 mgrid ((unsigned)((p.n_angles + 1 + 63) / 64), (unsigned)nc, (unsigned)std::min<int64_t> (64, gx)); // (slices of workgroups)
		hipLaunchKernelGGL (sweep_merge_kernel, mgrid, dim3 (64), 0, s, q.part, (int)gx, q.n_angles, q.angles, q.peaks, q.chn0, q.raw_slot, q.prune ? q.thr : nullptr, q.ctr,
		                    compact ? q.thr_blk : nullptr);
		err = hipGetLastError ();
		if (err != hipSuccess) {
			return err;
		}
	}
	return hipSuccess;
}

// The step table of a sweep over lists (one workgroup per channel): list l has cnt[l] entries from entry l * cap on; its steps of
// up to L entries follow those of the lists before it.  L = 1024 (a full stage) when there is plenty to do; a channel with few
// entries -- a shard of a sharded sweep behind the merged seeds of all shards: 1-2 M entries -- is cut into steps of 512, 256 or
// 128 entries instead, until there are about `want_steps` of them (eight per workgroup): the evaluation's workgroups claim FOUR steps when they start,
// so 1300 full steps kept 325 of a channel's 512 workgroups busy with four steps each and the others with none (round 5:
// sweep_eval_kernel 80 us of a 340 us shard pass).
__global__ __launch_bounds__ (256) void sweep_plan_kernel (const unsigned* __restrict__ cnt, int n_lists, int64_t cap, uint2* __restrict__ desc,
                                                             int64_t desc_stride, unsigned* __restrict__ n_steps, unsigned long long* total, unsigned want_steps)
{
	// blockIdx.x = channel; blockIdx.y = which part of every thread's lists this workgroup writes the steps of (each workgroup
	// repeats the scan over the list lengths -- a few thousand numbers --, the writes are what takes the time: 54 -> 12 us at 140 K steps)
	__shared__ unsigned part[256];
	const int       c   = blockIdx.x, t = threadIdx.x;
	const unsigned* cn  = cnt + (int64_t)c * n_lists;
	uint2*          dsc = desc + (int64_t)c * desc_stride;
	const int       per = (n_lists + 255) / 256, l0 = t * per, l1 = l0 + per < n_lists ? l0 + per : n_lists;
	unsigned        ents = 0; // (a channel of one call lists < 2^31 entries: prot_sweep_dev splits longer calls)
	for (int l = l0; l < l1; ++l) {
		ents += cn[l];
	}
	if (total != nullptr && ents != 0 && blockIdx.y == 0) {
		atomicAdd (total, (unsigned long long)ents);
	}
	auto scan = [&] (unsigned v) { // inclusive scan over the workgroup; returns this thread's value, part[255] = the sum
		part[t] = v;
		__syncthreads ();
		for (int d = 1; d < 256; d <<= 1) {
			const unsigned u = t >= d ? part[t - d] : 0u;
			__syncthreads ();
			part[t] += u;
			__syncthreads ();
		}
		return part[t];
	};
	(void)scan (ents);
	const unsigned tot = part[255];
	__syncthreads ();
	// steps of L = 1024 >> k entries: at most n_lists + tot / L of them; the table holds desc_stride - 8
	const unsigned limit = (unsigned)min ((int64_t)want_steps, desc_stride - 8);
	int            k     = 0;
	while (k < 3 && (unsigned)n_lists + (tot >> (9 - k)) <= limit) {
		++k;
	}
	const unsigned L = (unsigned)kStage >> k;
	unsigned       mine = 0;
	for (int l = l0; l < l1; ++l) {
		mine += (cn[l] + L - 1) / L;
	}
	unsigned  at = scan (mine) - mine;
	const int K = (int)gridDim.y, sub = (per + K - 1) / K, w0 = l0 + (int)blockIdx.y * sub, w1 = w0 + sub;
	for (int l = l0; l < l1; ++l) {
		const unsigned n = cn[l];
		if (l >= w0 && l < w1) {
			for (unsigned i = 0; i * L < n; ++i) {
				dsc[at + i] = make_uint2 ((unsigned)(((int64_t)c * n_lists + l) * cap + (int64_t)i * L), n - i * L < L ? n - i * L : L);
			}
		}
		at += (n + L - 1) / L;
	}
	if (t == 255 && blockIdx.y == 0) {
		n_steps[c] = part[255];
	}
}

hipError_t launch_sweep_plan (const unsigned* cnt, int n_lists, int64_t cap, uint2* desc, int64_t desc_stride, unsigned* n_steps, int n_chn,
                              unsigned long long* total, hipStream_t s)
{
	static int cus = 0;
	if (cus == 0) {
		int dev = 0;
		(void)hipGetDevice (&dev);
		if (hipDeviceGetAttribute (&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
			cus = 256;
		}
	}
	// (sweep_eval_kernel runs four workgroups per CU, shared among the channels: about eight steps for each)
#ifndef PROT_PLAN_WANT_MUL
#define PROT_PLAN_WANT_MUL 8u // (measured per shard of the harmonic hour, eval kernel us at blksiz 2048 / 16384: no cut 81 / 91, x4 76 / 90, x16 86 / 86, x64 142 / 132 --
#endif                        // every step costs 3-4 us of its own, too few leave workgroups without work: tools/ab_build.py sw:<tag>:-DPROT_PLAN_WANT_MUL=<n>u)
	const unsigned want = PROT_PLAN_WANT_MUL * (unsigned)std::max (1, 4 * cus / std::max (1, n_chn));
	const int      per  = (n_lists + 255) / 256;
	hipLaunchKernelGGL (sweep_plan_kernel, dim3 ((unsigned)n_chn, (unsigned)std::min (8, std::max (1, per))), dim3 (256), 0, s, cnt, n_lists, cap, desc, desc_stride,
	                    n_steps, total, want);
	return hipGetLastError ();
}

// The SEEDING pass's evaluation (round 5).  The seeding pass lists every frame of a sparse sample of blocks (no bound yet); until
// round 5 those lists went through the same plan / matrix-core / merge kernels as the main pass's, 55-60 us of dependent round
// trips for a few hundred thousand frames -- a fixed cost every shard of a sharded sweep pays again.  All a seed has to be is a
// LOWER bound of the final peaks: plain vector arithmetic, one angle per thread, every list entry from LDS, the maxima scaled
// down by 1e-4 (whatever rounding this arithmetic has against the matrix core's, the frame that holds an angle's true maximum
// exceeds the scaled value, is listed by the main pass and puts the exact value into the table: the tables stay bit-identical to
// the unpruned sweep's) and folded into the table by atomics (look first): a few thousand on ninety cache lines.
// Geometry: one workgroup per list and per 256 angles, one angle per thread; the list's entries pass through LDS in chunks of 1024
// (coalesced loads, broadcast reads of two entries each); the per-list maxima go to a scratch table, a second small kernel folds
// them into the peak table (as atomics they would be a few hundred thousand on ninety cache lines: the 100-200 us that made
// sweep_kernel give up atomics in round 3).
__global__ __launch_bounds__ (256) void sweep_seed_eval_kernel (const float2* __restrict__ surv, const unsigned* __restrict__ cnt, int n_lists, int64_t cap,
                                                                  const int* __restrict__ angles, int n_angles, const float* __restrict__ lut,
                                                                  float* __restrict__ part)
{
	__shared__ __attribute__ ((aligned (16))) float2 st[1024];
	const int   t   = threadIdx.x;
	const int   l   = (int)blockIdx.x >> 2, qt = (int)blockIdx.x & 3, cl = blockIdx.z; // (a quarter of list l: four workgroups per list)
	const int   idx = (int)blockIdx.y * 256 + t; // place in the angle list
	const bool  on  = idx < n_angles;
	const int   a   = on ? angles[idx] : 0;
	const float sa = lut[2 * a], ca = lut[2 * a + 1]; // lut = {sin, cos}
	float       m0 = 0.f, m1 = 0.f;
	const unsigned      n_all = cnt[(int64_t)cl * n_lists + l];
	const unsigned      qn    = ((n_all + 7u) / 8u) * 2u;        // entries per quarter (even: the LDS reads take pairs)
	const unsigned      q0    = qn * (unsigned)qt < n_all ? qn * (unsigned)qt : n_all;
	const unsigned      n     = n_all - q0 < qn ? n_all - q0 : qn;
	const float2* const list  = surv + ((int64_t)cl * n_lists + l) * cap + q0;
	for (unsigned base = 0; base < n; base += 1024u) {
		const unsigned m = n - base < 1024u ? n - base : 1024u;
		__syncthreads ();
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const unsigned i = (unsigned)t + 256u * (unsigned)k;
			st[i]            = i < m ? list[base + i] : make_float2 (0.f, 0.f); // (padding entries are zeros: they raise nothing)
		}
		__syncthreads ();
		const float4* const s4 = reinterpret_cast<const float4*> (st);
		const unsigned      m2 = (m + 1u) >> 1;
#pragma unroll 8
		for (unsigned i = 0; i < m2; ++i) {
			const float4 e = s4[i];
			m0             = fmaxf (m0, fabsf (fmaf (sa, e.y, ca * e.x)));
			m1             = fmaxf (m1, fabsf (fmaf (sa, e.w, ca * e.z)));
		}
	}
	if (on) {
		part[((int64_t)cl * 4 * n_lists + blockIdx.x) * n_angles + idx] = fmaxf (m0, m1);
	}
}

__global__ __launch_bounds__ (256) void sweep_seed_merge_kernel (const float* __restrict__ part, int n_rows, const int* __restrict__ angles, int n_angles,
                                                                   float* __restrict__ peaks, int chn0)
{
	// blockIdx.z = slice of the rows (one in sixteen): eight loads in flight per thread, then one atomic per angle and slice
	const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x, cl = blockIdx.y;
	if (idx >= n_angles) {
		return;
	}
	const float* const col = part + (int64_t)cl * n_rows * n_angles + idx;
	float m[8] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
	const int step = (int)gridDim.z;
	for (int r = (int)blockIdx.z; r < n_rows; r += 8 * step) {
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const int rr = r + k * step;
			m[k]         = fmaxf (m[k], rr < n_rows ? col[(int64_t)rr * n_angles] : 0.f);
		}
	}
	const float mx = fmaxf (fmaxf (fmaxf (m[0], m[1]), fmaxf (m[2], m[3])), fmaxf (fmaxf (m[4], m[5]), fmaxf (m[6], m[7])));
	if (mx > 0.f) {
		const unsigned  v    = __float_as_uint (mx * 0.9999f);
		unsigned* const slot = reinterpret_cast<unsigned*> (peaks + (int64_t)(chn0 + cl) * kMaxSample + angles[idx]);
		if (v > __hip_atomic_load (slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { // (non-negative floats order as their bit patterns)
			atomicMax (slot, v);
		}
	}
}

hipError_t launch_sweep_seed_eval (const float2* surv, const unsigned* cnt, int n_lists, int64_t cap, const int* angles, int n_angles, const float* lut,
                                   float* peaks, int chn0, int n_chn, float* part, size_t part_floats, hipStream_t s)
{
	if (n_lists <= 0 || n_angles <= 0 || n_chn <= 0) {
		return hipSuccess;
	}
	if (4 * (size_t)n_lists * (size_t)n_angles * (size_t)n_chn > part_floats || 4 * (int64_t)n_lists > 65535 * 16 || part == nullptr) {
		return hipErrorInvalidValue; // (the caller takes the matrix-core route)
	}
	const int  halves = (n_angles + 255) / 256;
	hipLaunchKernelGGL (sweep_seed_eval_kernel, dim3 ((unsigned)(4 * n_lists), (unsigned)halves, (unsigned)n_chn), dim3 (256), 0, s, surv, cnt, n_lists, cap, angles,
	                    n_angles, lut, part);
	hipError_t err = hipGetLastError ();
	if (err != hipSuccess) {
		return err;
	}
	hipLaunchKernelGGL (sweep_seed_merge_kernel, dim3 ((unsigned)halves, (unsigned)n_chn, 16), dim3 (256), 0, s, part, 4 * n_lists, angles, n_angles, peaks, chn0);
	return hipGetLastError ();
}

// floats of scratch a sweep launch may need on this device (every resident-workgroup slot x 4 waves x 360 angles)
size_t sweep_scratch_floats ()
{
	int dev = 0, cus = 256;
	(void)hipGetDevice (&dev);
	(void)hipDeviceGetAttribute (&cus, hipDeviceAttributeMultiprocessorCount, dev);
	return (size_t)cus * 8 * 4 * (kMaxSample + 1);
}

} // namespace prot
