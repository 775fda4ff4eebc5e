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
#include <cstring>
#include <type_traits>

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
	// (pk32a: every product an instruction of its own -- the launched kernel and the run() server must round alike, so nothing may be
	// left for the compiler to contract one way here and another way there)
	b1 = pkt_mul_w16<pk32a, DIR, 2> (b1); // W8
	b3 = pkt_mul_w16<pk32a, DIR, 6> (b3); // W8^3
	pkt_radix4<pk32a, DIR, false> (a0, a1, a2, a3);
	pkt_radix4<pk32a, DIR, true> (b0, b1, b2, b3); // (b2 arrives un-rotated: its W8^2 = -+j is folded into the butterfly)
	v[0] = a0, v[2] = a1, v[4] = a2, v[6] = a3;
	v[1] = b0, v[3] = b1, v[5] = b2, v[7] = b3;
}

// the block's last step, spelled out instruction by instruction (same bits in the launched kernel and in the run() server):
// steady rotation ca dry + g y; per-frame rotation cs.x dry + cs.y (gain y) with {cos, sin} of the pair's two frames
PROT_DEV f2 s5_mix (f2 y, float g, f2 dry, float ca)
{
	f2 t, o;
	asm ("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(dry), "v"(f2{ ca, ca }));
	asm ("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(o) : "v"(y), "v"(f2{ g, g }), "v"(t));
	return o;
}
PROT_DEV f2 s5_mix_ramp (f2 y, float gain, f2 dry, f2 cs0, f2 cs1)
{
	f2 h, t, o;
	asm ("v_pk_mul_f32 %0, %1, %2" : "=v"(h) : "v"(y), "v"(f2{ gain, gain }));
	asm ("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(dry), "v"(f2{ cs0.x, cs1.x }));
	asm ("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(o) : "v"(h), "v"(f2{ cs0.y, cs1.y }), "v"(t));
	return o;
}

PROT_DEV void twiddle7 (f2 (&v)[8], const f2 (&w)[7])
{
#pragma unroll
	for (int j = 1; j < 8; ++j) {
		v[j] = pk32::cmul<1> (v[j], w[j - 1]);
	}
}

// the constants a thread multiplies by, pass by pass (prot_kernels.h kSv5*): ONLY w of every table comes from memory, w^2 = w w,
// w^4 = w^2 w^2, w^3 = w w^2, w^5 = w w^4, w^6 = w^2 w^4, w^7 = w^3 w^4 are products.  A constant's load costs a lone block 0.07 us
// (BASELINE configs[2], one box: all 43 per thread loaded 9.6 us per launch, w w^2 w^4 7.3, w alone 6.4) and a product's rounding
// costs accuracy: against the oracle on 200 000 frames max 3.6e-7 / rms 7.9e-8 with every constant loaded (the 256-thread block:
// 3.6e-7 / 8.0e-8), 4.2e-7 / 8.2e-8 with three, 6.0e-7 / 1.04e-7 with one -- the gate is 2e-6 (profiles/r06_experiments.txt;
// -DPROT_S5_W124 / -DPROT_S5_ALL_TWIDDLES build the other two)
struct S5Tw {
	f2 fA[7], fB[7], fC[7], iA[7], iB[7], iC[7];
	f2 rho; // |B| of this thread's k4 = 0 and k4 = 7 bins over the mean (1 outside the rows next to DC)
};

PROT_DEV void s5_load_tw (S5Tw& w, const f2* __restrict__ tab, const int T)
{
	const int wave = __builtin_amdgcn_readfirstlane (T >> 6), lane = T & 63;
	const int tpp  = 64 * (lane & 7) + 8 * (lane >> 3) + wave; // the inverse's first twiddle index: t'' = 64 k3 + 8 k2 + k1
#pragma unroll
	for (int j = 0; j < 7; ++j) {
#if defined(PROT_S5_W124)
		if (j == 0 || j == 1 || j == 3)
#elif !defined(PROT_S5_ALL_TWIDDLES)
		if (j == 0)
#endif
		{
			w.fA[j] = tab[kSv5fA + 512 * j + T];
			w.fB[j] = tab[kSv5fB + 64 * j + lane];
			w.fC[j] = tab[kSv5fC + 8 * j + (lane & 7)];
			w.iA[j] = tab[kSv5iA + 512 * j + tpp];
			w.iB[j] = tab[kSv5iB + 64 * j + 8 * (lane >> 3) + wave];
			w.iC[j] = tab[kSv5iC + 8 * j + wave];
		}
	}
	w.rho = tab[kSv5Rho + T];
}

PROT_DEV void s5_tw_powers (S5Tw& w)
{
#ifndef PROT_S5_ALL_TWIDDLES
	auto powers = [] (f2 (&x)[7]) {
#ifndef PROT_S5_W124
		x[1] = pk32::cmul<1> (x[0], x[0]);
		x[3] = pk32::cmul<1> (x[1], x[1]);
#endif
		x[2] = pk32::cmul<1> (x[0], x[1]);
		x[4] = pk32::cmul<1> (x[0], x[3]);
		x[5] = pk32::cmul<1> (x[1], x[3]);
		x[6] = pk32::cmul<1> (x[2], x[3]);
	};
	powers (w.fA), powers (w.fB), powers (w.fC), powers (w.iA), powers (w.iB), powers (w.iC);
#endif
}

// where a thread writes and reads in the three exchanges (forward; the inverse swaps the roles)
struct S5Lds {
	LdsC5 *x1w, *x1r, *x2w, *x2r, *x3w, *x3r;
};

PROT_DEV S5Lds s5_lds (f2* const ex1, f2* const ex2, const int T)
{
	const int    lane = T & 63;
	LdsC5* const ch   = (LdsC5*)ex2 + (T >> 6) * kS5Chunk;
	S5Lds        l;
	l.x1w = (LdsC5*)ex1 + (T >> 6) * 64 + lane;  // exchange 1 writer (b = wave): + 512 k1
	l.x1r = (LdsC5*)ex1 + (T >> 6) * 512 + lane; // exchange 1 reader (k1 = wave): + 64 b
	l.x2w = ch + 72 * (lane >> 3) + (lane & 7);  // lane (c, d):  + 8 k2
	l.x2r = ch + 8 * (lane >> 3) + (lane & 7);   // lane (k2, d): + 72 c
	l.x3w = ch + 9 * lane;                       // lane (k2, d): + k3
	l.x3r = ch + 72 * (lane >> 3) + (lane & 7);  // lane (k2, k3): + 9 d
	return l;
}

// v[a] = z[512 a + T] -> v[a] = 4096 y[512 a + T], y = IFFT ((-i) B FFT_{n - 1/2} (z)) with |B| = rho (the bulk: the mean, in the gain).
// Two workgroup barriers; the caller puts one more between two blocks.
PROT_DEV void s5_transform (f2 (&v)[8], const S5Tw& w, const S5Lds& l)
{
	// ---- forward, half-shifted
	dft8<1> (v);
	twiddle7 (v, w.fA);
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		l.x1w[512 * k] = v[k];
	}
	__syncthreads ();
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		v[k] = l.x1r[64 * k];
	}
	dft8<1> (v);
	twiddle7 (v, w.fB);
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		l.x2w[8 * k] = v[k];
	}
	wave_sync ();
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		v[k] = l.x2r[72 * k];
	}
	dft8<1> (v);
	twiddle7 (v, w.fC);
	wave_sync ();
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		l.x3w[k] = v[k];
	}
	wave_sync ();
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		v[k] = l.x3r[9 * k];
	}
	dft8<1> (v);
	// ---- W16^{-k4}, then (-i) B[k] = i rho[k]: rho = 1 but for k4 = 0 (k3 < 4) and k4 = 7 (k3 >= 4), this thread's two factors
	{
		constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, R = 0.70710678118654752440f;
		const f2 w16[8] = { mkf2 (1.f, 0.f), mkf2 (C1, S1), mkf2 (R, R), mkf2 (S1, C1), mkf2 (0.f, 1.f), mkf2 (-S1, C1), mkf2 (-R, R), mkf2 (-C1, S1) };
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			// i W16^{-k} = i (cos + i sin)(pi k / 8) = (-sin, cos)
			f2 x = mkf2 (-w16[k].y, w16[k].x);
			if (k == 0) {
				x = x * mkf2 (w.rho.x, w.rho.x);
			}
			if (k == 7) {
				x = x * mkf2 (w.rho.y, w.rho.y);
			}
			v[k] = pk32::cmul<1> (v[k], x);
		}
	}
	// ---- inverse: the same decimation from the k side
	dft8<-1> (v);
	twiddle7 (v, w.iA);
	wave_sync ();
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		l.x3r[9 * k] = v[k];
	}
	wave_sync ();
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		v[k] = l.x3w[k];
	}
	dft8<-1> (v);
	twiddle7 (v, w.iB);
	wave_sync ();
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		l.x2r[72 * k] = v[k];
	}
	wave_sync ();
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		v[k] = l.x2w[8 * k];
	}
	dft8<-1> (v);
	twiddle7 (v, w.iC);
	// (exchange 1 backwards: wave k1 writes slots [512 k1, 512 k1 + 512), which nobody but itself has read in this block)
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		l.x1r[64 * k] = v[k];
	}
	__syncthreads ();
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		v[k] = l.x1w[512 * k];
	}
	dft8<-1> (v);
}

} // namespace

// One workgroup = 512 threads = eight waves.  Forward: thread T holds z[512 a + T]; pass over a; exchange 1 (all waves: slot
// (8 k1 + b) 64 + t', t' = T & 63, b = T >> 6) -> thread (k1, t') holds b = 0..7; pass over b; exchange 2 (inside wave k1: slot
// (9 c + k2) 8 + d) -> lane (k2, d) holds c; pass over c; exchange 3 (slot (8 k2 + d) 9 + k3) -> lane (k2, k3) holds d; pass over d.
// The inverse runs the same decimation from the k side and mirrors the three exchanges.  Two workgroup barriers per block.
// What a block needs before it can ask for its rows and constants travels in the kernel's first fourteen argument dwords, which the
// hardware PRELOADS into scalar registers at wave launch (-mllvm -amdgpu-kernarg-preload-count=14 for this unit: build.py) -- the
// first scalar load of a kernel's arguments is a miss that every load of the block waits behind (0.25 us of a 7.3 us launch in an A/B on
// one box: profiles/r06_experiments.txt); everything else (rotation, output pointer, gain) is read from `p` while the rows are on their way.  (On firmware without the preload the compiler's
// prologue loads the same registers: slower, not wrong.)  dd_hx = dry delay | history workgroups << 20; runs = workgroups with a block.
template <int HS, int MODE, bool PIPED = false>
__global__ __launch_bounds__ (512) void olasave512_kernel (const float* const in0, const int64_t in_stride, const float* const hist0, const float* const sv, const int n_in32,
                                                           const int n_out32, const int dd_hx, const int hist_len, const int runs, const OlaParams p)
{
	constexpr int BLK = 512 * HS; // output frames per block
	constexpr int NT  = 16 - HS;  // context rows
	constexpr int R0  = NT / 2;   // first valid output row = first dry row
	static_assert (MODE == MODE_ROT || MODE == MODE_RAMP || MODE == MODE_HILBERT, "steady or per-frame rotation, or the Hilbert branch alone");
	__shared__ f2 ex1[4096];
	__shared__ f2 ex2[8 * kS5Chunk];
	PROT_S5_T (0)
	const int c  = (int)blockIdx.y;
	const int  hx = (dd_hx >> 20) & 0x3ff, Dd = dd_hx & 0xfffff; // (Dd even: host guarantees)
	constexpr bool pipelined = PIPED; // (p.hs_flag != nullptr: an instantiation of its own -- 0.23 us of a 7.1 us launch as a run-time branch)
	if (hx > 0 && (int)blockIdx.x >= runs) { // a workgroup of the next history
		if (!pipelined) {
			hist_write_part (p, c, (int)blockIdx.x - runs, hx);
			return;
		}
		// pipelined stateful launches (PROT_CFG_PIPELINE: OlaParams::hs_*, prot_olasave.hip has the story): the launch behind this one,
		// on another stream and possibly running already, needs nothing of this one but the history -- written through, awaited,
		// counted; the last of the history workgroups publishes this launch's generation
		hist_write_part<true> (p, c, (int)blockIdx.x - runs, hx);
		asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads ();
		if (threadIdx.x == 0) {
			const unsigned total = (unsigned)hx * gridDim.y;
			if (__hip_atomic_fetch_add (p.hs_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1u) {
				__hip_atomic_store (p.hs_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store (p.hs_flag + (p.hs_post & 7u), p.hs_post, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		return;
	}
	int bx = (int)blockIdx.x; // (run_of_workgroup's rotation inside groups of eight: XCD k does not own the blocks k, k + 8, ...)
	if ((bx | 7) < runs) {
		bx = (bx & ~7) | ((bx + (bx >> 3)) & 7);
	}
	const int64_t n_out = n_out32, n_in = n_in32;
	const int64_t r0    = (int64_t)bx * BLK;
	if (r0 >= n_out) {
		return;
	}
	const int64_t r1 = r0 + BLK < n_out ? r0 + BLK : n_out;
	const int T = (int)threadIdx.x;

	const float* __restrict__ in   = in0 + (int64_t)c * in_stride;
	const float* __restrict__ hist = hist0 ? hist0 + (int64_t)c * hist_len : nullptr;
	const bool    in_al8  = (reinterpret_cast<uintptr_t> (in) & 7) == 0;
	bool          piped   = false;
	if (r0 - Dd - 512 * R0 < 1) { // the block reaches into the history
		piped = pipelined;
		if (piped && p.hs_wait != 0u) {
			// ... which the launch before this one is writing (or has written) on another stream.  Bounded (prot_olasave.hip: the launch
			// waited for cannot be behind this one in a queue; a wait that never ends is reported in the status word, PROT_EHIP)
			if (threadIdx.x == 0) {
				unsigned spins = 0;
				while (__hip_atomic_load (p.hs_flag + (p.hs_wait & 7u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.hs_wait) {
					__builtin_amdgcn_s_sleep (8);
					if (++spins > kPipeSpinMax) {
						__hip_atomic_store (p.hs_flag + kPipeStatusWord, p.hs_wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
						break;
					}
				}
			}
			__syncthreads ();
			__builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "agent");
		}
	}

	// the engine's constants (behind the 256-thread block's fifteen rows): asked for now, used pass by pass
	S5Tw w;
	s5_load_tw (w, reinterpret_cast<const f2*> (sv) + kSv5Offset, T);
	const S5Lds l = s5_lds (ex1, ex2, T);

	{ // (one block per workgroup)
		const int64_t s = r0;
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
			// history or silence in reach: every frame by an unconditional load from an address chosen without a branch.  (A pipelined
			// launch's history was written by another launch while this one was already running: device-scope loads -- a plain one could
			// hit a stale line of this XCD's L2.  Two copies of the loop under one wave-uniform branch: a choice per load would put
			// control flow between the loads)
			const float* const safe = sv;
			auto stage = [&] (auto device_scope) {
				auto ldq = [&] (int64_t i) -> float {
					const int64_t h    = i + hist_len;
					const bool    ok_i = i >= 0 && i < n_in;
					const bool    ok_h = i < 0 && hist != nullptr && h >= 0;
					const float*  a    = ok_i ? in + i : (ok_h ? hist + h : safe);
					float         x;
					if constexpr (decltype (device_scope)::value) {
						x = __hip_atomic_load (const_cast<float*> (a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					} else {
						x = *a;
					}
					return (ok_i || ok_h) ? x : 0.f;
				};
#pragma unroll
				for (int a = 0; a < 8; ++a) {
					const int64_t f  = a0 + 1024 * a + 2 * (int64_t)T;
					const float   pm = ldq (f - 1);
					dry[a]           = mkf2 (ldq (f), ldq (f + 1));
					v[a]             = mkf2 (pm, dry[a].x);
				}
			};
			if constexpr (PIPED) {
				if (piped) {
					stage (std::true_type ());
				} else {
					stage (std::false_type ());
				}
			} else {
				stage (std::false_type ());
			}
		}
		PROT_S5_T (1)
		s5_tw_powers (w); // (the constants' products: behind the wait for the block's rows)
		s5_transform (v, w, l);
		// ---- the rest of the arguments: asked for by the compiler wherever it likes, waited for here at the earliest
		float* __restrict__ out = p.out + (int64_t)c * p.out_stride;
		const bool out_al8      = (reinterpret_cast<uintptr_t> (out) & 7) == 0;
		float      ca = 0.f, g = p.sv_gain;
		if (MODE == MODE_ROT) {
			ca = p.rot ? p.rot[2 * c] : p.rot_v[2 * c];
			g  = (p.rot ? p.rot[2 * c + 1] : p.rot_v[2 * c + 1]) * p.sv_gain;
		}
		ca = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (ca)));
		g  = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (g)));
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
			if (pos >= r1) {
				continue;
			}
			f2 o;
			if (MODE == MODE_RAMP) {
				const f2* const rs = reinterpret_cast<const f2*> (p.rot_s) + ((int64_t)c * n_out + pos);
				o                  = s5_mix_ramp (v[a], g, dry[a], rs[0], pos + 1 < r1 ? rs[1] : mkf2 (0.f, 0.f));
			} else {
				o = s5_mix (v[a], g, dry[a], ca); // (MODE_HILBERT: ca = 0)
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
	}
	PROT_S5_T (4)
}

// ---- the run() server (round 4; round 6: on the 512-thread block) ------------------------------------------------------------------
// An LV2 host calls run() (src/phaserotate.c:774-852) with ~1024 frames at a time, and a kernel launch plus a stream synchronisation
// cost more than the block itself.  The server is a RESIDENT kernel -- one 512-thread workgroup per channel of every LV2 engine of the
// process that asks (SvServerSlots), started by the first run() and left running: the host stores the call's parameters (the OlaParams a
// launch would have carried) and its input into the mailbox (SvServerBox, prot_kernels.h: device memory through the PCIe BAR where there
// is a large one, pinned host-coherent memory otherwise) and rings a doorbell word; channel 0's workgroup polls the doorbell, relays
// the decision to the engine's other channels through a device word (ONE decision for all: serve call s, or leave), every workgroup does
// its channel's block, writes its output through to the host (system-scope stores, each acknowledged) and stores the call's number into
// its done word, which the host polls.  No launch, no stream synchronisation, no copy command per call.
// What a workgroup keeps from call to call (round 6): the frames in front of the call in LDS (sh_stream = [kSrvHist frames | the
// call's input | silence]: a block's rows are unconditional LDS reads -- staging was 2.8 us of a 13.4 us call), the constants in registers.
// Leaving: on the host's stop command (any other use of the engine stops the server first: prot_lv2.hip) and BY ITSELF after 0.25 s
// without a doorbell or, between two calls, after 2 s of residence (a resident kernel blocks whatever shares its hardware queue and
// every device-wide synchronisation of the process: INTEGRATION.md, "run() latency and jitter"); `alive` is its last store.
#ifndef PROT_SRV_POLL_SLEEP
#define PROT_SRV_POLL_SLEEP 4 // (x 64 cycles between two looks at the doorbell; 1 and 0 measured the same to 0.1 us per call)
#endif
constexpr uint64_t kServerIdleTicks = 25000000ull;  // 100 MHz ticks
constexpr uint64_t kServerLifeTicks = 200000000ull;

#ifdef PROT_SERVER_STAMPS
} // namespace prot
__device__ unsigned g_srv_t[16];
extern "C" int prot_debug_srv_t (unsigned* out)
{
	return hipMemcpyFromSymbol (out, HIP_SYMBOL (g_srv_t), sizeof (g_srv_t)) == hipSuccess ? 0 : -1;
}
namespace prot {
#define PROT_SRV_T(i)                                                            \
	if (threadIdx.x == 0 && chn == 0) {                                           \
		g_srv_t[i] = (unsigned)__builtin_amdgcn_s_memrealtime ();                 \
	}
#else
#define PROT_SRV_T(i)
#endif

namespace {

typedef __attribute__ ((address_space (3))) float LdsF5;

// the next history of a served call in device memory, hist_out[i] = [old history | input](i + n_in): read by a launch behind the
// kernel's end or by this workgroup after a restart -- nobody waits for it (behind the done word); eight loads in flight per thread
PROT_DEV void server5_next_history (const OlaParams& p, const int c, const float* const sh_in)
{
	if (p.hist_out == nullptr) {
		return;
	}
	const float* __restrict__ hist = p.hist ? p.hist + (int64_t)c * p.hist_len : nullptr;
	float* __restrict__ ho         = p.hist_out + (int64_t)c * p.hist_len;
	const int64_t n_in   = p.n_in;
	const int64_t keep64 = (int64_t)p.hist_len - n_in;
	const int     keep   = keep64 > 0 ? (int)keep64 : 0; // entries that come from the old history
	for (int base = (int)threadIdx.x; base < p.hist_len; base += 512 * 8) {
		float v[8];
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			const int i = base + 512 * q < p.hist_len ? base + 512 * q : p.hist_len - 1;
			if (i < keep) {
				v[q] = hist != nullptr ? *(PROT_GLOBAL const float*)(hist + n_in + i) : 0.f;
			} else {
				v[q] = *(LdsF5*)(sh_in + (i - keep64));
			}
		}
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			if (base + 512 * q < p.hist_len) {
				*(PROT_GLOBAL float*)(ho + base + 512 * q) = v[q];
			}
		}
	}
}

template <int HS>
PROT_DEV void server5_loop (const SvServerBox* const box, SvServerAck* const ack, unsigned* const relay, const unsigned last_seq, const int chn, const int n_chn,
                            const int no_life_cap, f2* const ex1, f2* const ex2, unsigned* const sh_w, float* const sh_stream, unsigned* const sh_p)
{
	constexpr int NT = 16 - HS, R0 = NT / 2;
	const float* const port_in = reinterpret_cast<const float*> (reinterpret_cast<const char*> (box) + kServerInOffset);
	float* const       sh_hist = sh_stream;
	float* const       sh_in   = sh_stream + kSrvHist;
	const int          tid     = threadIdx.x;
	unsigned           seen    = last_seq;
	const float*       hist_tag = nullptr; // the history in device memory whose last kSrvHist frames sh_hist holds (nullptr: none yet)
	const float*       sv_tag   = nullptr; // the constants table w was loaded from
	S5Tw               w;
	const S5Lds        l = s5_lds (ex1, ex2, tid);
	// (pad0[c]: how far channel c's workgroup has come -- 1 resident, 2 doorbell seen, 3 parameters read, 4 block done, 5 done
	// word stored, 9 left; what the host prints when a call is not served)
	auto stage = [&] (unsigned v) {
		int tid_s = tid;
		asm volatile ("" : "+v"(tid_s));
		if (tid_s == 0 && chn < 8) {
			__hip_atomic_store (&ack->pad0[chn], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef PROT_SERVER_STAMPS
			if (chn == 0 && v >= 2u && v <= 5u) { // 100 MHz ticks at the stages of channel 0's workgroup
				__hip_atomic_store (&ack->pad0[6 + v], (unsigned)__builtin_amdgcn_s_memrealtime (), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			}
#endif
		}
	};
	for (int i = tid; i < kSrvIn; i += 512) {
		*(LdsF5*)(sh_in + i) = 0.f;
	}
	stage (1u);
	const uint64_t t_born = __builtin_amdgcn_s_memrealtime ();
	for (;;) {
		// (the thread id made opaque at each use, the decision made wave-uniform by readfirstlane: with `tid == 0` guarding both
		// the end of one pass -- the done word -- and the start of the next -- the poll --, the compiler rotated the loop into
		// "thread 0: store, poll; everybody else: straight back to the barrier", and the other threads served the same call
		// again and again while thread 0 waited, masked off, to store its done word: found on the hardware with the stage words)
		int tid_poll = tid;
		asm volatile ("" : "+v"(tid_poll));
		if (tid_poll == 0) {
			unsigned word;
			unsigned frames = 0u;
			if (chn == 0) {
				unsigned       s = seen, cmd = 0u;
				const uint64_t t_0 = __builtin_amdgcn_s_memrealtime ();
				for (;;) {
					const uint64_t bell = __hip_atomic_load (const_cast<uint64_t*> (&box->bell), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					s                   = (unsigned)bell;
					if (s != seen) {
						cmd    = (unsigned)(bell >> 63);
						frames = (unsigned)(bell >> 32) & 0x7fffffffu;
						break;
					}
					const uint64_t now = __builtin_amdgcn_s_memrealtime ();
					if (now - t_0 > kServerIdleTicks || (!no_life_cap && now - t_born > kServerLifeTicks)) {
						cmd = 1u;
						break;
					}
					__builtin_amdgcn_s_sleep (PROT_SRV_POLL_SLEEP);
				}
				word = 2u * s + cmd;
				if (n_chn > 1) {
					// (decision and frame count in ONE 8-byte store: nothing else has to be ordered with it -- what the other channels read
					// next is the host's mailbox, past the caches -- so no release, which on this path was a wait for every earlier store)
					__hip_atomic_store (reinterpret_cast<uint64_t*> (relay), (uint64_t)word | ((uint64_t)frames << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			} else {
				uint64_t both;
				while ((unsigned)(both = __hip_atomic_load (reinterpret_cast<uint64_t*> (relay), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 2u * seen) {
					__builtin_amdgcn_s_sleep (1);
				}
				word   = (unsigned)both;
				frames = (unsigned)(both >> 32);
			}
			sh_w[0] = word;
			sh_w[1] = frames;
		}
		__syncthreads ();
		const unsigned word = (unsigned)__builtin_amdgcn_readfirstlane ((int)sh_w[0]);
		if (word & 1u) {
			break;
		}
		stage (2u);
		// The host's stores (parameters, input frames, per-frame rotation) happened before its doorbell store, and what is read of
		// them below is read past the caches (system-scope loads) after thread 0 has seen the doorbell.  An AGENT-scope acquire: it
		// empties this CU's vector cache (which may hold lines of the mailbox's rotation area or of a history buffer as they were some
		// calls ago); a system-scope one would invalidate the XCD's L2 with them.
		__builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "agent");
		const int frames = __builtin_amdgcn_readfirstlane ((int)sh_w[1]) < 512 * HS ? __builtin_amdgcn_readfirstlane ((int)sh_w[1]) : 512 * HS;
		for (int i = tid; i < (int)(sizeof (OlaParams) / 4); i += 512) {
			sh_p[i] = __hip_atomic_load (reinterpret_cast<unsigned*> (const_cast<OlaParams*> (&box->p)) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		{ // (the port block: [channel][frames], prot_lv2.hip lv2_run_chunk; eight bytes per lane when the channel's frames start on an 8-byte boundary)
			const float* const src = port_in + (int64_t)chn * frames;
			if (((chn * frames) & 1) == 0) {
				uint64_t* const       dst8 = reinterpret_cast<uint64_t*> (sh_in);
				const uint64_t* const src8 = reinterpret_cast<const uint64_t*> (src);
				for (int i = tid; i < frames / 2; i += 512) {
					dst8[i] = __hip_atomic_load (const_cast<uint64_t*> (src8) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				}
				if ((frames & 1) && tid == 0) {
					sh_in[frames - 1] = __hip_atomic_load (const_cast<float*> (src) + frames - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				}
			} else {
				for (int i = tid; i < frames; i += 512) {
					sh_in[i] = __hip_atomic_load (const_cast<float*> (src) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				}
			}
		}
		__syncthreads ();
		const OlaParams& p = *reinterpret_cast<const OlaParams*> (sh_p); // (p.in is the mailbox: the block reads sh_in)
		{
			// the frames in front of this call: held in LDS since the call before it (whose next history in device memory is this call's
			// history: the same address), fetched otherwise
			const float* const hg = uniform_ptr_generic (p.hist ? p.hist + (int64_t)chn * p.hist_len : nullptr);
			if (hg != hist_tag) {
				for (int base = tid; base < kSrvHist; base += 512 * 4) {
					float v[4];
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						const int h = base + 512 * q - (kSrvHist - p.hist_len); // index into the engine's history (shorter or longer than the copy)
						v[q]        = (hg != nullptr && h >= 0 && base + 512 * q < kSrvHist) ? *(PROT_GLOBAL const float*)(hg + h) : 0.f;
					}
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						if (base + 512 * q < kSrvHist) {
							*(LdsF5*)(sh_hist + base + 512 * q) = v[q];
						}
					}
				}
				hist_tag = hg;
				__syncthreads ();
			}
			const float* const svg = uniform_ptr_generic (p.sv);
			if (svg != sv_tag) { // the engine's constants: once per residence
				s5_load_tw (w, reinterpret_cast<const f2*> (svg) + kSv5Offset, tid);
				s5_tw_powers (w);
				sv_tag = svg;
			}
		}
		stage (3u);
		PROT_SRV_T (5)
		PROT_SRV_T (0)
		{
			// ---- the call's block: output frames [0, n_out), frame f of the virtual stream at sh_in[f]
			const int   Dd   = p.dry_delay;
			const int   a0   = -Dd - 512 * R0; // (one block per call: s = 0)
			const int   r1   = (int)p.n_out;
			const bool  ramp = p.rot_s != nullptr;
			float       ca = 0.f, g = p.sv_gain;
			if (!ramp) {
				ca = p.rot ? *(PROT_GLOBAL const float*)(p.rot + 2 * chn) : p.rot_v[2 * chn];
				g  = (p.rot ? *(PROT_GLOBAL const float*)(p.rot + 2 * chn + 1) : p.rot_v[2 * chn + 1]) * p.sv_gain;
			}
			ca = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (ca)));
			g  = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (g)));
			int T = tid;
			asm volatile ("" : "+v"(T)); // (opaque per call: per-lane addresses are otherwise kept alive across the poll loop)
			T &= 511;
			f2 v[8], dry[8];
			const float* const q = sh_in + a0 + 2 * T;
#pragma unroll
			for (int a = 0; a < 8; ++a) {
				const float pm = *(LdsF5*)(q + 1024 * a - 1);
				dry[a]         = *(__attribute__ ((address_space (3))) const f2*)(q + 1024 * a);
				v[a]           = mkf2 (pm, dry[a].x);
			}
			PROT_SRV_T (1)
			s5_transform (v, w, l);
			PROT_SRV_T (2)
			float* __restrict__ out = p.out + (int64_t)chn * p.out_stride;
			const bool out_al8      = (reinterpret_cast<uintptr_t> (out) & 7) == 0;
			const int  hi = T >> 8, tp = T & 255;
#pragma unroll
			for (int a = 0; a < 8; ++a) {
				const int k = 2 * a + hi - R0;
				if (k < 0 || k >= HS) {
					continue;
				}
				const int pos = 512 * k + 2 * tp;
				if (pos >= r1) {
					continue;
				}
				f2 o;
				if (ramp) { // per-frame rotation (an angle on its way to its target): {cos, sin} per frame from the mailbox
					const f2 cs0 = *((PROT_GLOBAL const f2*)p.rot_s + ((int64_t)chn * p.n_out + pos));
					const f2 cs1 = pos + 1 < r1 ? *((PROT_GLOBAL const f2*)p.rot_s + ((int64_t)chn * p.n_out + pos + 1)) : mkf2 (0.f, 0.f);
					o            = s5_mix_ramp (v[a], g, dry[a], cs0, cs1);
				} else {
					o = s5_mix (v[a], g, dry[a], ca);
				}
				// (the outputs cross the host link as system-scope stores -- written through, acknowledged by the memory side, one 8-byte store
				// per lane, whole lines per wave -- so that publishing them needs no release fence: at system scope that is a write-back of
				// the XCD's L2, 0.3-0.8 us of every call in an A/B on one box)
				if (out_al8 && pos + 1 < r1) {
					__hip_atomic_store (reinterpret_cast<uint64_t*> (out + pos), __builtin_bit_cast (uint64_t, o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				} else {
					__hip_atomic_store (out + pos, o.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					if (pos + 1 < r1) {
						__hip_atomic_store (out + pos + 1, o.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					}
				}
			}
			PROT_SRV_T (3)
			PROT_SRV_T (4)
		}
		stage (4u);
		asm volatile ("s_waitcnt vmcnt(0)" ::: "memory"); // every output store of this wave acknowledged; the barrier: of every wave
		__syncthreads ();
		int tid_done = tid;
		asm volatile ("" : "+v"(tid_done));
		if (tid_done == 0) {
			__hip_atomic_store (&ack->done[chn], word >> 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		stage (5u);
		// behind the done word: the next history in device memory, the LDS copy moved on, silence behind the next call's input
		server5_next_history (p, chn, sh_in);
		if (p.hist_out != nullptr) {
			static_assert (kSrvHist % 512 == 0, "whole turns of the workgroup");
			constexpr int PER = kSrvHist / 512;
			const int     n_in = (int)p.n_in, n = n_in < kSrvHist ? n_in : kSrvHist;
			float         x[PER];
#pragma unroll
			for (int q = 0; q < PER; ++q) { // new [i] = old [i + n] while that exists, the input's tail behind it
				const int i = tid + 512 * q;
				x[q]        = i + n < kSrvHist ? *(LdsF5*)(sh_hist + i + n) : *(LdsF5*)(sh_in + (n_in - kSrvHist + i));
			}
			__syncthreads ();
#pragma unroll
			for (int q = 0; q < PER; ++q) {
				*(LdsF5*)(sh_hist + tid + 512 * q) = x[q];
			}
			hist_tag = uniform_ptr_generic (p.hist_out + (int64_t)chn * p.hist_len);
		}
		__syncthreads ();
		for (int i = tid; i < kSrvIn; i += 512) {
			*(LdsF5*)(sh_in + i) = 0.f;
		}
		seen = word >> 1;
	}
	stage (9u);
	if (tid == 0 && chn == 0) {
		__hip_atomic_store (&ack->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

} // namespace

__global__ __launch_bounds__ (512, 1) void olasave512_server_kernel (const SvServerSlots slots)
{
	__shared__ f2       ex1[4096];
	__shared__ f2       ex2[8 * kS5Chunk];
	__shared__ unsigned sh_w[2];
	// [the kSrvHist frames in front of the call | the call's input | silence]: one array, a block's rows are unconditional reads of it
	__shared__ __attribute__ ((aligned (16))) float sh_stream[kSrvHist + kSrvIn];
	__shared__ __attribute__ ((aligned (16))) unsigned sh_p[(sizeof (OlaParams) + 3) / 4]; // the call's parameters (read in place)
	const SvServerSlot& sl = slots.s[blockIdx.y];
	if (sl.hop_rows == 8) {
		server5_loop<8> (sl.box, sl.ack, sl.relay, sl.last_seq, sl.chn, sl.n_chn, sl.no_life_cap, ex1, ex2, sh_w, sh_stream, sh_p);
	} else {
		server5_loop<10> (sl.box, sl.ack, sl.relay, sl.last_seq, sl.chn, sl.n_chn, sl.no_life_cap, ex1, ex2, sh_w, sh_stream, sh_p);
	}
}

hipError_t launch_ola_save_server (const SvServerSlots& slots, hipStream_t s)
{
	if (slots.n < 1 || slots.n > kServerMaxChn) {
		return hipErrorInvalidValue;
	}
	for (int i = 0; i < slots.n; ++i) {
		const SvServerSlot& sl = slots.s[i];
		if ((sl.hop_rows != 8 && sl.hop_rows != 10) || sl.box == nullptr || sl.ack == nullptr || sl.relay == nullptr) {
			return hipErrorInvalidValue;
		}
		if (sl.chn == 0) {
			hipError_t err = hipMemsetD32Async (reinterpret_cast<hipDeviceptr_t> (sl.relay), (int)(2u * sl.last_seq), 1, s);
			if (err != hipSuccess) {
				return err;
			}
		}
	}
	hipLaunchKernelGGL (olasave512_server_kernel, dim3 (1, (unsigned)slots.n), dim3 (512), 0, s, slots);
	return hipGetLastError ();
}

// a call the server is to take: one block per channel, rotation steady or per frame -- the parameters go into the box instead of into a launch
hipError_t post_ola_save_server (const OlaParams& p, int hop_rows, int n_rho_rows, int n_chn, SvServerBox* box)
{
	if ((p.dry_delay & 1) || (p.run_frames % (512 * hop_rows)) != 0 || p.accumulate || p.sv == nullptr || n_rho_rows != 1 ||
	    2 * (p.dry_delay - p.conv_delay) != 512 * (16 - hop_rows) || p.n_out <= 0 || p.run_stride != 0 || p.surv != nullptr || p.hilbert_only ||
	    p.hs_flag != nullptr || p.n_out > 512 * hop_rows || n_chn > kServerMaxChn || (hop_rows != 8 && hop_rows != 10) ||
	    p.hist == nullptr || p.dry_delay + 512 * ((16 - hop_rows) / 2) + 2 > kSrvHist || 8192 - p.dry_delay - 512 * ((16 - hop_rows) / 2) > kSrvIn ||
	    p.n_in > 512 * hop_rows) {
		return hipErrorInvalidValue;
	}
	memcpy (&box->p, &p, sizeof (OlaParams));
	return hipSuccess;
}

template <int HS>
static hipError_t launch_s5_m (const OlaParams& p, dim3 grid, hipStream_t s)
{
	const int runs  = (int)grid.x - p.hist_extra;
	const int dd_hx = p.dry_delay | (p.hist_extra << 20);
#define PROT_S5_LAUNCH(M) hipLaunchKernelGGL ((olasave512_kernel<HS, M>), grid, dim3 (512), 0, s, p.in, p.in_stride, p.hist, p.sv, (int)p.n_in, (int)p.n_out, dd_hx, p.hist_len, runs, p)
	if (p.hs_flag != nullptr) { // (pipelined launches: steady rotation only -- prot_apply_dev's route)
		if (p.hilbert_only || p.rot_s != nullptr) {
			return hipErrorInvalidValue;
		}
		hipLaunchKernelGGL ((olasave512_kernel<HS, MODE_ROT, true>), grid, dim3 (512), 0, s, p.in, p.in_stride, p.hist, p.sv, (int)p.n_in, (int)p.n_out, dd_hx, p.hist_len, runs, p);
		return hipGetLastError ();
	}
	if (p.hilbert_only) {
		PROT_S5_LAUNCH (MODE_HILBERT);
	} else if (p.rot_s != nullptr) {
		PROT_S5_LAUNCH (MODE_RAMP);
	} else {
		PROT_S5_LAUNCH (MODE_ROT);
	}
#undef PROT_S5_LAUNCH
	return hipGetLastError ();
}

// does launch_ola_save512 take this launch?  One transform block per workgroup at most `max_wg` workgroups in all (the caller: fewer
// than the device has CUs), steady rotation or the Hilbert branch, nothing fused (sweep lists, pipelined history hand-over)
bool ola_save512_takes (const OlaParams& p, int hop_rows, int n_rho_rows, int n_chn, int max_wg)
{
	if ((hop_rows != 8 && hop_rows != 10 && hop_rows != 12) || n_rho_rows != 1 || (p.dry_delay & 1) || p.accumulate || p.sv == nullptr || p.n_out <= 0 ||
	    2 * (p.dry_delay - p.conv_delay) != 512 * (16 - hop_rows) || p.surv != nullptr || p.run_stride != 0 || (p.hs_flag != nullptr && (p.hist_out == nullptr || p.rot_s != nullptr)) ||
	    p.n_in >= ((int64_t)1 << 31) || p.dry_delay >= (1 << 20) ||
	    p.hilbert_only) { // (the sweep's Hilbert passes: a sharded or chunked sweep is bit-identical to the single pass, so its block may not depend on a launch's length)
		return false;
	}
	const int64_t blocks = (p.n_out + 512 * hop_rows - 1) / (512 * hop_rows);
	// (a single block per channel is a run() call on its launch route: the same block as the run() server's, so that a stream of
	// run() calls does not depend, to the bit, on which of them the server took)
	return blocks * n_chn <= max_wg;
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
