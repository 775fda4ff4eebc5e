// SPDX-License-Identifier: GPL-2.0-or-later
// The SHORT-CHAIN overlap-save block (round 6): the 4096-point transform pair of olasave_kernel
// (prot_olasave.hip: src/phaserotate.c:540-760 restated as a polyphase overlap-save block) on a 512-THREAD workgroup -- 8 points
// per thread, four radix-8 passes each way -- for launches whose workgroups are alone on their CUs: a stateful launch of a few
// hundred kilo-frames (BASELINE configs[2]: 128 blocks on 256 CUs) or a run() call that takes the launch route is ONE block per
// workgroup, and that block is a dependent chain with nothing to overlap it.  With 256 threads x 16 points the chain is ~860
// arithmetic + ~200 LDS instructions per wave, one wave per SIMD, 3.4 us; here it is half as long and two waves share every SIMD.
// The maths, thread by thread and LDS slot by LDS slot, is tools/model/olasave512_model.py (checked against the direct
// convolution); the constants are the second half of the engine's table (prot_engine.hip save_tables: kSv5*).
// The rounding differs from the 256-thread block's (other factorisation): within the 2e-6 gate of the oracle like every other
// form of the filter (tests: test_gpu_short_chain.py), not bit-identical to it -- which kernel a launch takes depends on the launch's
// shape only, never on the data.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "prot_dev.h"
#include "prot_fft.h"
#include "prot_kernels.h"

#ifdef PROT_S5_CLOCK // diagnostic build (tools/ab_build.py s5:clk:-DPROT_S5_CLOCK; tools/cfg3_timeline.py): 100 MHz ticks per workgroup
__device__ unsigned long long g_s5_wg[1024 * 8]; // kernel entry, rows + constants arrived, forward done, inverse done, exit
extern "C" int prot_debug_s5_wg (unsigned long long* out, int n_wg)
{
	return hipMemcpyFromSymbol (out, HIP_SYMBOL (g_s5_wg), sizeof (unsigned long long) * 8 * (size_t)(n_wg < 1024 ? n_wg : 1024)) == hipSuccess ? 0 : -1;
}
#define PROT_S5_T(i)                                                                \
	if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 1024) {                  \
		g_s5_wg[8 * blockIdx.x + i] = __builtin_amdgcn_s_memrealtime ();             \
	}
#else
#define PROT_S5_T(i)
#endif

namespace prot {

namespace {

typedef volatile __attribute__ ((address_space (3))) f2 LdsC5;
constexpr int kS5Chunk = 576; // complex slots of a wave's own exchange area: 8 x 8 x 8 with every eighth row padded by one (9 x 64)

// X[k] = sum_n v[n] W8^{nk} (DIR > 0) or its conjugate kernel (DIR < 0), natural order in and out: one radix-2 layer, two DFT-4
template <int DIR>
PROT_DEV void dft8 (f2 (&v)[8])
{
	f2 a0 = v[0] + v[4], b0 = v[0] - v[4];
	f2 a1 = v[1] + v[5], b1 = v[1] - v[5];
	f2 a2 = v[2] + v[6], b2 = v[2] - v[6];
	f2 a3 = v[3] + v[7], b3 = v[3] - v[7];
	b1 = pkt_mul_w16<pk32, DIR, 2> (b1); // W8
	b3 = pkt_mul_w16<pk32, DIR, 6> (b3); // W8^3
	pkt_radix4<pk32, DIR, false> (a0, a1, a2, a3);
	pkt_radix4<pk32, DIR, true> (b0, b1, b2, b3); // (b2 arrives un-rotated: its W8^2 = -+j is folded into the butterfly)
	v[0] = a0, v[2] = a1, v[4] = a2, v[6] = a3;
	v[1] = b0, v[3] = b1, v[5] = b2, v[7] = b3;
}

PROT_DEV void twiddle7 (f2 (&v)[8], const f2 (&w)[7])
{
#pragma unroll
	for (int j = 1; j < 8; ++j) {
		v[j] = pk32::cmul<1> (v[j], w[j - 1]);
	}
}

} // namespace

// One workgroup = 512 threads = eight waves.  Forward: thread T holds z[512 a + T]; pass over a; exchange 1 (all waves: slot
// (8 k1 + b) 64 + t', t' = T & 63, b = T >> 6) -> thread (k1, t') holds b = 0..7; pass over b; exchange 2 (inside wave k1: slot
// (9 c + k2) 8 + d) -> lane (k2, d) holds c; pass over c; exchange 3 (slot (8 k2 + d) 9 + k3) -> lane (k2, k3) holds d; pass over d.
// The inverse runs the same decimation from the k side and mirrors the three exchanges.  Two workgroup barriers per block.
template <int HS, int MODE>
__global__ __launch_bounds__ (512) void olasave512_kernel (const OlaParams p)
{
	constexpr int BLK = 512 * HS; // output frames per block
	constexpr int NT  = 16 - HS;  // context rows
	constexpr int R0  = NT / 2;   // first valid output row = first dry row
	static_assert (MODE == MODE_ROT || MODE == MODE_HILBERT, "steady rotation or the Hilbert branch alone");
	__shared__ f2 ex1[4096];
	__shared__ f2 ex2[8 * kS5Chunk];
	PROT_S5_T (0)
	kernarg_prefetch ();
	const int c = (int)blockIdx.y;
	if (p.hist_extra > 0 && (int)blockIdx.x >= (int)gridDim.x - p.hist_extra) { // a workgroup of the next history
		hist_write_part (p, c, (int)blockIdx.x - ((int)gridDim.x - p.hist_extra), p.hist_extra);
		return;
	}
	int64_t r0, rlen;
	run_of_workgroup (p, r0, rlen);
	if (r0 >= p.n_out) {
		return;
	}
	const int64_t r1 = (r0 + rlen < p.n_out) ? r0 + rlen : p.n_out;
	int T = (int)threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane (T >> 6), lane = T & 63;

	const float* __restrict__ in   = p.in + (int64_t)c * p.in_stride;
	const float* __restrict__ hist = p.hist ? p.hist + (int64_t)c * p.hist_len : nullptr;
	float* __restrict__ out        = p.out + (int64_t)c * p.out_stride;
	const int64_t n_in    = p.n_in;
	const int     Dd      = p.dry_delay; // even (host guarantees)
	const bool    in_al8  = (reinterpret_cast<uintptr_t> (in) & 7) == 0;
	const bool    out_al8 = (reinterpret_cast<uintptr_t> (out) & 7) == 0;
	float ca = 0.f, g = p.sv_gain;
	if (MODE == MODE_ROT) {
		ca = p.rot ? p.rot[2 * c] : p.rot_v[2 * c];
		g  = (p.rot ? p.rot[2 * c + 1] : p.rot_v[2 * c + 1]) * p.sv_gain;
	}
	ca = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (ca)));
	g  = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (g)));

	// the engine's constants (behind the 256-thread block's fifteen rows): asked for now, used pass by pass
	const f2* __restrict__ tab = reinterpret_cast<const f2*> (p.sv) + kSv5Offset;
	f2 wfA[7], wfB[7], wfC[7], wiA[7], wiB[7], wiC[7];
	const int tpp = 64 * (lane & 7) + 8 * (lane >> 3) + wave; // the inverse's first twiddle index: t'' = 64 k3 + 8 k2 + k1
#ifndef PROT_S5_ALL_TWIDDLES
	// (w, w^2, w^4 of every table from memory, w^3 = w w^2, w^5 = w w^4, w^6 = w^2 w^4, w^7 = w^3 w^4 by one or two products: forty-three
	// 8-byte loads per thread were 176 KB per workgroup through one CU's cache -- more than a microsecond of the block)
#pragma unroll
	for (int j = 0; j < 7; ++j) {
		if (j == 0 || j == 1 || j == 3) {
			wfA[j] = tab[kSv5fA + 512 * j + T];
			wfB[j] = tab[kSv5fB + 64 * j + lane];
			wfC[j] = tab[kSv5fC + 8 * j + (lane & 7)];
			wiA[j] = tab[kSv5iA + 512 * j + tpp];
			wiB[j] = tab[kSv5iB + 64 * j + 8 * (lane >> 3) + wave];
			wiC[j] = tab[kSv5iC + 8 * j + wave];
		}
	}
	auto powers = [] (f2 (&w)[7]) {
		w[2] = pk32::cmul<1> (w[0], w[1]);
		w[4] = pk32::cmul<1> (w[0], w[3]);
		w[5] = pk32::cmul<1> (w[1], w[3]);
		w[6] = pk32::cmul<1> (w[2], w[3]);
	};
#else
#pragma unroll
	for (int j = 0; j < 7; ++j) {
		wfA[j] = tab[kSv5fA + 512 * j + T];
		wfB[j] = tab[kSv5fB + 64 * j + lane];
		wfC[j] = tab[kSv5fC + 8 * j + (lane & 7)];
		wiA[j] = tab[kSv5iA + 512 * j + tpp];
		wiB[j] = tab[kSv5iB + 64 * j + 8 * (lane >> 3) + wave];
		wiC[j] = tab[kSv5iC + 8 * j + wave];
	}
	auto powers = [] (f2 (&)[7]) {};
#endif
	const f2 rho = tab[kSv5Rho + T]; // |B| of this thread's k4 = 0 and k4 = 7 bins (1 outside the rows next to DC)

	LdsC5* const x1w = (LdsC5*)ex1 + (T >> 6) * 64 + lane;        // exchange 1 writer (b = wave): + 512 k1
	LdsC5* const x1r = (LdsC5*)ex1 + (T >> 6) * 512 + lane;       // exchange 1 reader (k1 = wave): + 64 b
	LdsC5* const ch  = (LdsC5*)ex2 + (T >> 6) * kS5Chunk;
	LdsC5* const x2w = ch + 72 * (lane >> 3) + (lane & 7);         // lane (c, d):  + 8 k2
	LdsC5* const x2r = ch + 8 * (lane >> 3) + (lane & 7);          // lane (k2, d): + 72 c
	LdsC5* const x3w = ch + 9 * lane;                              // lane (k2, d): + k3
	LdsC5* const x3r = ch + 72 * (lane >> 3) + (lane & 7);         // lane (k2, k3): + 9 d

	for (int64_t s = r0; s < r1; s += BLK) {
		const int64_t a0 = s - Dd - 512 * R0; // frame of (row 0, thread 0).q: z[j] = x[a0 + 2 j - 1] + i x[a0 + 2 j], dry pair (x[a0 + 2 j], x[a0 + 2 j + 1])
		f2 v[8], dry[8];
		if (in_al8 && a0 >= 1 && a0 + 8192 <= n_in) { // the whole block inside the input: (q, c) pairs are aligned, the frame in front a word
			const float* q = in + a0 + 2 * T;
#pragma unroll
			for (int a = 0; a < 8; ++a) {
				const float pm = q[1024 * a - 1];
				dry[a]         = *reinterpret_cast<const f2*> (q + 1024 * a);
				v[a]           = mkf2 (pm, dry[a].x);
			}
		} else {
			// history or silence in reach: every frame by an unconditional load from an address chosen without a branch
			const float* const safe = p.sv;
			auto ldq = [&] (int64_t i) -> float {
				const int64_t h    = i + p.hist_len;
				const bool    ok_i = i >= 0 && i < n_in;
				const bool    ok_h = i < 0 && hist != nullptr && h >= 0;
				const float*  a    = ok_i ? in + i : (ok_h ? hist + h : safe);
				const float   x    = *a;
				return (ok_i || ok_h) ? x : 0.f;
			};
#pragma unroll
			for (int a = 0; a < 8; ++a) {
				const int64_t f  = a0 + 1024 * a + 2 * (int64_t)T;
				const float   pm = ldq (f - 1);
				dry[a]           = mkf2 (ldq (f), ldq (f + 1));
				v[a]             = mkf2 (pm, dry[a].x);
			}
		}
		PROT_S5_T (1)
		// ---- forward, half-shifted
		dft8<1> (v);
		if (s == r0) { // (the constants' products: once per workgroup, where the first pass's arithmetic hides them)
			powers (wfA), powers (wfB), powers (wfC), powers (wiA), powers (wiB), powers (wiC);
		}
		twiddle7 (v, wfA);
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			x1w[512 * k] = v[k];
		}
		__syncthreads ();
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			v[k] = x1r[64 * k];
		}
		dft8<1> (v);
		twiddle7 (v, wfB);
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			x2w[8 * k] = v[k];
		}
		wave_sync ();
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			v[k] = x2r[72 * k];
		}
		dft8<1> (v);
		twiddle7 (v, wfC);
		wave_sync ();
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			x3w[k] = v[k];
		}
		wave_sync ();
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			v[k] = x3r[9 * k];
		}
		dft8<1> (v);
		// ---- W16^{-k4}, then (-i) B[k] = i rho[k]: rho = 1 but for k4 = 0 (k3 < 4) and k4 = 7 (k3 >= 4), this thread's two factors
		{
			constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, R = 0.70710678118654752440f;
			const f2 w16[8] = { mkf2 (1.f, 0.f), mkf2 (C1, S1), mkf2 (R, R), mkf2 (S1, C1), mkf2 (0.f, 1.f), mkf2 (-S1, C1), mkf2 (-R, R), mkf2 (-C1, S1) };
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				// i W16^{-k} = i (cos + i sin)(pi k / 8) = (-sin, cos)
				f2 w = mkf2 (-w16[k].y, w16[k].x);
				if (k == 0) {
					w = w * mkf2 (rho.x, rho.x);
				}
				if (k == 7) {
					w = w * mkf2 (rho.y, rho.y);
				}
				v[k] = pk32::cmul<1> (v[k], w);
			}
		}
		PROT_S5_T (2)
		// ---- inverse: the same decimation from the k side
		dft8<-1> (v);
		twiddle7 (v, wiA);
		wave_sync ();
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			x3r[9 * k] = v[k];
		}
		wave_sync ();
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			v[k] = x3w[k];
		}
		dft8<-1> (v);
		twiddle7 (v, wiB);
		wave_sync ();
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			x2r[72 * k] = v[k];
		}
		wave_sync ();
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			v[k] = x2w[8 * k];
		}
		dft8<-1> (v);
		twiddle7 (v, wiC);
		// (exchange 1 backwards: wave k1 writes slots [512 k1, 512 k1 + 512), which nobody but itself has read in this block)
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			x1r[64 * k] = v[k];
		}
		__syncthreads ();
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			v[k] = x1w[512 * k];
		}
		dft8<-1> (v);
		PROT_S5_T (3)
		// ---- thread T holds y[512 a + T]: local row 2 a + (T >> 8), pair T & 255; rows R0 .. R0 + HS - 1 are the block's output
		const int hi = T >> 8, tp = T & 255;
#pragma unroll
		for (int a = 0; a < 8; ++a) {
			const int k = 2 * a + hi - R0;
			if (k < 0 || k >= HS) {
				continue;
			}
			const int64_t pos = s + 512 * (int64_t)k + 2 * tp;
			f2            o   = v[a] * mkf2 (g, g);
			if (MODE == MODE_ROT) {
				o = o + dry[a] * mkf2 (ca, ca);
			}
			if (out_al8 && pos + 1 < r1) {
				*reinterpret_cast<f2*> (out + pos) = o;
			} else {
				if (pos < r1) {
					out[pos] = o.x;
				}
				if (pos + 1 < r1) {
					out[pos + 1] = o.y;
				}
			}
		}
		__syncthreads (); // (the next block's first exchange writes where this one's last was read)
	}
	PROT_S5_T (4)
}

template <int HS>
static hipError_t launch_s5_m (const OlaParams& p, dim3 grid, hipStream_t s)
{
	if (p.hilbert_only) {
		hipLaunchKernelGGL ((olasave512_kernel<HS, MODE_HILBERT>), grid, dim3 (512), 0, s, p);
	} else {
		hipLaunchKernelGGL ((olasave512_kernel<HS, MODE_ROT>), grid, dim3 (512), 0, s, p);
	}
	return hipGetLastError ();
}

// does launch_ola_save512 take this launch?  One transform block per workgroup at most `max_wg` workgroups in all (the caller: fewer
// than the device has CUs), steady rotation or the Hilbert branch, nothing fused (sweep lists, pipelined history hand-over)
bool ola_save512_takes (const OlaParams& p, int hop_rows, int n_rho_rows, int n_chn, int max_wg)
{
	if ((hop_rows != 8 && hop_rows != 10 && hop_rows != 12) || n_rho_rows != 1 || (p.dry_delay & 1) || p.accumulate || p.sv == nullptr || p.n_out <= 0 ||
	    2 * (p.dry_delay - p.conv_delay) != 512 * (16 - hop_rows) || p.surv != nullptr || p.hs_flag != nullptr || p.rot_s != nullptr || p.run_stride != 0 ||
	    p.hilbert_only) { // (the sweep's Hilbert passes: a sharded or chunked sweep is bit-identical to the single pass, so its block may not depend on a launch's length)
		return false;
	}
	const int64_t blocks = (p.n_out + 512 * hop_rows - 1) / (512 * hop_rows);
	// (a single block per channel is a run() call on its launch route: it keeps the block of the run() server -- the 256-thread one --
	// so that a stream of run() calls does not depend, to the bit, on which of them the server took)
	return blocks >= 2 && blocks * n_chn <= max_wg;
}

hipError_t launch_ola_save512 (const OlaParams& p, int hop_rows, int n_chn, hipStream_t s)
{
	OlaParams q  = p;
	q.run_frames = 512 * hop_rows; // one block per workgroup
	q.rank_cu    = 0;
	const int64_t runs = (p.n_out + q.run_frames - 1) / q.run_frames;
	dim3          grid ((unsigned)runs, (unsigned)n_chn);
	if (p.hist_out != nullptr) { // the next history by four more workgroups per channel, beside the blocks
		q.hist_extra = 4;
		grid.x += 4;
	}
	switch (hop_rows) {
		case 8: return launch_s5_m<8> (q, grid, s);
		case 10: return launch_s5_m<10> (q, grid, s);
		case 12: return launch_s5_m<12> (q, grid, s);
		default: return hipErrorInvalidValue;
	}
}

} // namespace prot
