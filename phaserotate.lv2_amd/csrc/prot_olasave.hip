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
//     of B when the engine is created (prot_engine.hip) and keeps olapoly_kernel for any FIR that fails them.
//  3. The rotation constants arrive as kernel arguments (no copy in front of the launch).
//
// The transform itself is olapoly_kernel's: 16 x 16 x 16, three radix-16 passes in registers (prot_fft.h), one
// workgroup-wide and one wave-local LDS exchange per direction (prot_dev.h), packed-fp32 butterflies, split outer twiddles.
#include "prot_dev.h"

#include <cstring>

namespace prot {

typedef volatile __attribute__ ((address_space (3))) f2 LdsC;
typedef volatile __attribute__ ((address_space (3))) float LdsF;

#ifdef PROT_SV_INC // (tuning builds: a variant of the generated code, tools/ab_build.py sv:...)
#include PROT_SV_INC
#else
#include "prot_olasave_asm.inc"
#endif
#ifdef PROT_SV_CLOCK // diagnostic build (tools/ab_build.py sv:clk:-DPROT_SV_CLOCK): shader cycles and 100 MHz ticks spent in the generated loop
__device__ unsigned long long g_sv_clock[4];
__device__ unsigned long long g_sv_wg[4096 * 4]; // per workgroup: kernel entry, first generated-loop entry, last exit, kernel exit (100 MHz ticks)
extern "C" int prot_debug_sv_clock (unsigned long long* out)
{
	return hipMemcpyFromSymbol (out, HIP_SYMBOL (g_sv_clock), sizeof (g_sv_clock)) == hipSuccess ? 0 : -1;
}
extern "C" int prot_debug_sv_wg (unsigned long long* out, int n_wg)
{
	return hipMemcpyFromSymbol (out, HIP_SYMBOL (g_sv_wg), sizeof (unsigned long long) * 4 * (size_t)(n_wg < 4096 ? n_wg : 4096)) == hipSuccess ? 0 : -1;
}
#endif
#define PROT_SV_CAT_(k, a, b, c) PROT_SV_##k##_##a##_##b##_##c
#define PROT_SV_CLOB_(a) PROT_SV_ASM_CLOBBERS_##a
// one generated block of code: KIND_ = ASM (a run of nf whole blocks streamed from / to global memory) or CORE (one block staged
// in LDS); the transform's mode AM_ (0 mix with the dry rows, 2 Hilbert branch only)
#define PROT_SV_CALL(KIND_, HS_, AM_, in_, out_, nf_, ca_, sa_)                                                                        \
	if (HS == HS_ && AM == AM_) {                                                                                                       \
		asm volatile (PROT_SV_CAT_ (KIND_, HS_, AM_, 1)                                                                                 \
		              :                                                                                                                 \
		              : "s"(in_), "s"(out_), "s"(nf_), "s"(lds_off), "s"(a_sv), "s"(ca_), "s"(sa_), "v"(t0)                                \
		              : PROT_SV_CLOB_ (HS_));                                                                                           \
	}
// MODE_SWEEP: the generated loop with the sweep's test in its epilogue (no output rows: survivors appended at cursor_, the
// running max |x| in raw_); cursor_ and raw_ are read and written
#define PROT_SV_SWEEP_CALL(HS_, in_, cursor_, raw_, nf_, thr_, gain_)                                                                  \
	if (HS == HS_) {                                                                                                                    \
		asm volatile (PROT_SV_SWEEP_##HS_                                                                                               \
		              : "=s"(cursor_), "=v"(raw_)                                                                                       \
		              : "s"(in_), "s"(nf_), "s"(lds_off), "s"(a_sv), "s"(thr_), "s"(gain_), "v"(t0), "s"(cursor_), "v"(raw_)             \
		              : PROT_SV_SWEEP_CLOBBERS_##HS_);                                                                                  \
	}
#define PROT_SV_ALL(KIND_, in_, out_, nf_, ca_, sa_)                                                                                   \
	PROT_SV_CALL (KIND_, 8, 0, in_, out_, nf_, ca_, sa_)                                                                                \
	PROT_SV_CALL (KIND_, 8, 2, in_, out_, nf_, ca_, sa_)                                                                                \
	PROT_SV_CALL (KIND_, 10, 0, in_, out_, nf_, ca_, sa_)                                                                               \
	PROT_SV_CALL (KIND_, 10, 2, in_, out_, nf_, ca_, sa_)                                                                               \
	PROT_SV_CALL (KIND_, 12, 0, in_, out_, nf_, ca_, sa_)                                                                               \
	PROT_SV_CALL (KIND_, 12, 2, in_, out_, nf_, ca_, sa_)

// HS new rows (512 HS frames) per block; MODE_ROT / MODE_RAMP / MODE_HILBERT.
// ALL arithmetic of the block is generated code (prot_olasave_asm.inc, tools/gen/gen_olasave.py: 120 hand-allocated VGPRs);
// this function moves data.  A workgroup walks its run block by block:
//  * consecutive blocks that lie wholly inside the input array (8-byte aligned streams, steady rotation) are one call of the
//    generated LOOP: the next block's rows prefetched, the context rows carried in registers, stores straight from registers;
//  * every other block -- history or silence in reach, a partial last block, odd alignment, per-frame rotation -- is staged
//    here into the LDS exchange area (16 rows of (q, c) pairs + the frames in front of each wave's first pair), transformed by
//    the generated CORE (the same instructions on the same values: the two paths agree bit for bit, and a sharded or chunked
//    sweep does not depend on where its chunks end), and stored from LDS with bounds.
template <int HS, int MODE>
PROT_DEV void olasave_body (const OlaParams& p, f2* const smem, float* const sw_lo_s)
{
	constexpr int BLK = 512 * HS; // output frames per block
	constexpr int NT  = 16 - HS;  // context rows
	constexpr int R0  = NT / 2;   // first valid output row = first dry row
	constexpr int AM  = MODE == MODE_ROT ? 0 : 2; // the generated code's mode (per-frame rotation, sweep: Hilbert branch, mixed / tested here)
	static_assert (HS == 8 || HS == 10 || HS == 12, "rows per block (HS = 14 needs 124 VGPRs before any temporary: olapoly_kernel)");
	static_assert (MODE != MODE_SWEEP || HS != 10, "the fused sweep exists for the CLI engines (blksiz 2048, 4096)");

	const int t0 = threadIdx.x;
	const int c  = (int)blockIdx.y;
#ifdef PROT_SV_CLOCK
	const uint64_t wg_t0 = __builtin_amdgcn_s_memrealtime ();
	uint64_t       wg_a0 = 0, wg_a1 = 0;
#endif
	// Pipelined stateful launches: the next launch (on the engine's other stream, possibly running already) needs nothing
	// of this one but the history, i.e. the tail of this launch's INPUT -- written first, counted, and published by the
	// last workgroup to get there.  (The host takes this route only when n_in >= hist_len: the slice is then a plain copy
	// of input frames and does not read the previous history.)
	const bool piped = MODE != MODE_SWEEP && p.hs_flag != nullptr;
	if constexpr (MODE != MODE_SWEEP) {
		if (p.hist_extra > 0 && (int)blockIdx.x >= (int)gridDim.x - p.hist_extra) { // a workgroup of the next history (short launches)
			hist_write_part (p, c, (int)blockIdx.x - ((int)gridDim.x - p.hist_extra), p.hist_extra);
			return;
		}
	}
	// MODE_SWEEP: the bound of this launch -- the smallest of the table's entries for the call's angles as they stand now (they only
	// grow: a lower bound of every final peak), squared, less prot_sweep.hip's margin; every wave appends to a list of its own
	float            sw_raw = 0.f;
	// (the bound is parked in global memory -- behind the list lengths, one copy per wave -- and re-read where it is used: a
	// register copy would live across the generated code.  An LDS word behind the exchange area was tried first and read back
	// other values on the hardware from the second staged block on, for a reason that was not found: tools/attic/sweep_shard_repro.py shows the symptom when the word is moved back)
	auto sw_thr_slot = [&] () -> float* {
		const int w = __builtin_amdgcn_readfirstlane (threadIdx.x >> 6);
		return reinterpret_cast<float*> (p.surv_cnt) + (int64_t)gridDim.y * gridDim.x * 4 + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + w;
	};
	auto sw_thr = [&] () -> float {
#if defined(PROT_SW_THR_LDS) && PROT_SW_THR_LDS == 1
		// (diagnostic build, DESIGN.md 4.4b: the round-4 form whose reads came back wrong on the hardware.  ROOT CAUSE, found in the
		// ISA in round 5 (tools/lint_exec_copies.py): the generic pointer makes this a flat load whose 64-bit per-lane address the
		// compiler keeps in VGPRs; the generated code owns v0..v119, so the address is copied into one of v120..v127 in front of
		// every asm block and back behind it -- and the copy in front of the staged block's asm was placed INSIDE the divergent
		// region that gathers the block's front frames (EXEC = lanes 8..11 of each wave): the other sixty lanes of the copy never
		// received the address, lane 0 -- the one v_readfirstlane reads -- came back with whatever that register held before, and
		// the load went to LDS word 0 or anywhere else.  Not an overlapping LDS write: a per-lane value kept alive across the
		// generated code, the one thing this file forbids itself -- here by the compiler's hand.)
		return __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (((volatile float*)sw_lo_s)[0])));
#elif defined(PROT_SW_THR_LDS)
		// (the same word in LDS, read through an address that is made at the point of use from a scalar: nothing per-lane lives
		// across the generated code -- tables bit-identical on the hardware, tools/attic/sweep_shard_repro.py)
		unsigned a = (unsigned)(uintptr_t)(__attribute__ ((address_space (3))) float*)sw_lo_s;
		asm volatile ("" : "+s"(a));
		float v;
		asm volatile ("v_mov_b32 %0, %1\n\tds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "s"(a) : "memory");
		return __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (v)));
#else
		return __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (__hip_atomic_load (sw_thr_slot (), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))));
#endif
	};
	unsigned         sw_cnt  = 0u; // entries this wave has appended (wave-uniform; the list's address is recomputed where it is needed:
	                               // nothing but this count, the bound and the running maximum lives across the generated code)
	// (wave reductions with the lane id made opaque at each use: with __shfl_xor the compiler keeps the six permutation indices of
	// the reduction at the top of the kernel alive for the one at its end -- across the generated code, i.e. in scratch)
	auto sw_wave_reduce = [] (float v, bool take_max) -> float {
		int lane;
		asm volatile ("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
		for (int m = 1; m < 64; m <<= 1) {
			const float o = __int_as_float (__builtin_amdgcn_ds_bpermute ((lane ^ m) << 2, __float_as_int (v)));
			v             = take_max ? fmaxf (v, o) : fminf (v, o);
		}
		return v;
	};
	auto sw_list = [&] () -> float2* {
		const int w = __builtin_amdgcn_readfirstlane (threadIdx.x >> 6);
		return uniform_ptr_generic (p.surv + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) * p.surv_cap);
	};
	if (MODE == MODE_SWEEP) {
		float lo = 1.0e18f;
		if (!p.sw_thr_zero) {
			for (int i = t0; i < p.sw_n_angles; i += 256) {
				lo = fminf (lo, __builtin_nontemporal_load (&p.sw_peaks[(int64_t)c * kMaxSample + p.sw_angles[i]]));
			}
		} else {
			lo = 0.f;
		}
		lo = sw_wave_reduce (lo, false);
		if ((t0 & 63) == 0) {
			sw_lo_s[t0 >> 6] = lo;
		}
		__syncthreads ();
		lo = fminf (fminf (sw_lo_s[0], sw_lo_s[1]), fminf (sw_lo_s[2], sw_lo_s[3]));
		__syncthreads ();
		if ((t0 & 63) == 0) {
			__hip_atomic_store (sw_thr_slot (), lo * lo * 0.9999f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PROT_SW_THR_LDS
		if (t0 == 0) {
			sw_lo_s[0] = lo * lo * 0.9999f;
		}
		__syncthreads ();
#endif
	}
	if (piped) {
		// my slice of the next history: device-scope (write-through) stores, so that publishing it needs no release FENCE --
		// at device scope that is a write-back of the XCD's whole L2, and with one per workgroup two overlapping launches
		// took 28 us each instead of 11.7 (measured).  The stores are awaited (s_waitcnt: a write-through store is
		// acknowledged by the memory side), then counted with a device-scope atomic; the last workgroup to count publishes.
		{
			const int per = (p.hist_len + (int)gridDim.x - 1) / (int)gridDim.x;
			const int i0  = (int)blockIdx.x * per;
			const int i1  = i0 + per < p.hist_len ? i0 + per : p.hist_len;
			const float* __restrict__ src = p.in + (int64_t)c * p.in_stride + (p.n_in - p.hist_len);
			float* __restrict__ ho        = p.hist_out + (int64_t)c * p.hist_len;
			for (int i = i0 + t0; i < i1; i += 256) {
				__hip_atomic_store (&ho[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		// (a workgroup-scope release is no wait at all when the waves of a workgroup share their CU's cache: the count
		// below would overtake the stores of the other waves)
		asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads ();
		if (t0 == 0) {
			const unsigned total = gridDim.x * gridDim.y;
			if (__hip_atomic_fetch_add (p.hs_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1u) {
				__hip_atomic_store (p.hs_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store (p.hs_flag + (p.hs_post & 7u), p.hs_post, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	}
	int64_t   r0, rlen;
	run_of_workgroup (p, r0, rlen);
	if (r0 >= p.n_out) {
		if (!piped && p.hist_extra == 0) {
			hist_write_slice (p, c, threadIdx.x);
		}
		if (MODE == MODE_SWEEP && (t0 & 63) == 0) {
			p.surv_cnt[((int64_t)c * gridDim.x + blockIdx.x) * 4 + (t0 >> 6)] = 0u;
		}
		return;
	}
	const int64_t r1 = (r0 + rlen < p.n_out) ? r0 + rlen : p.n_out;
	if (piped && p.hs_wait != 0u && r0 - p.dry_delay - 512 * R0 < 1) {
		// this run starts inside the history the previous launch is writing (or has written) on the other stream
		if (t0 == 0) {
			// bounded: the launch waited for cannot be behind this one in a queue (host rule: pipelined launches take at most
			// 1/6 of the resident-workgroup slots), so the wait ends within a launch's duration -- if it ever does not (a
			// runtime that serialises the streams differently, a lost launch) the kernel gives up after ~2 s instead of
			// hanging the device, and says so in the status word the host checks at the next synchronisation (PROT_EHIP)
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

	const float* __restrict__ in   = p.in + (int64_t)c * p.in_stride;
	const float* __restrict__ hist = p.hist ? p.hist + (int64_t)c * p.hist_len : nullptr;
	const int hist_len             = p.hist_len;
	float* __restrict__ out        = p.out + (int64_t)c * p.out_stride;
	const int64_t n_in = p.n_in;
	const int     Dd   = p.dry_delay; // even (host guarantees)
	const bool in_al8  = (reinterpret_cast<uintptr_t> (in) & 7) == 0;
	const bool out_al8 = (reinterpret_cast<uintptr_t> (out) & 7) == 0;

	if (MODE == MODE_SWEEP && p.sw_raw_slot >= 0 && p.sw_raw_tail && blockIdx.x == 0) {
		// the call's last dry_delay input frames: the rows reach them only as the next chunk's history (see prot_sweep.hip)
		const float* xin = p.in + (int64_t)c * p.in_stride;
		for (int64_t i = p.n_in - p.dry_delay + t0; i < p.n_in; i += 256) {
			sw_raw = fmaxf (sw_raw, i >= 0 ? fabsf (xin[i]) : 0.f);
		}
	}
	float ca = 0.f, sa = p.sv_gain;
	if (MODE == MODE_ROT) {
		ca = p.rot ? p.rot[2 * c] : p.rot_v[2 * c];
		sa = (p.rot ? p.rot[2 * c + 1] : p.rot_v[2 * c + 1]) * p.sv_gain;
	}
	ca = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (ca)));
	sa = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (sa)));
	const unsigned lds_off = (unsigned)(uintptr_t)(__attribute__ ((address_space (3))) f2*)smem;
	const float*   a_sv    = uniform_ptr_generic (p.sv);

	// the (x[a], x[a+1]) pair of thread t in the 512-frame row that starts at frame ar (a = ar + 2 t): a row lies wholly in the
	// input array or wholly in the history (both start on row boundaries) -- aligned 8-byte loads; anything else (silence in
	// reach, odd alignment) frame by frame through the virtual stream [history | input | silence]
	// (a pipelined launch reads a history that another launch, possibly on another XCD, wrote while this one was already
	// running: frame by frame with device-scope loads -- the aligned 8-byte route could hit a stale line of this XCD's L2,
	// found as launch 5 of a series reading what launch 1 had read from the same buffer of the ring)
	const bool hist_al8 = !piped && hist != nullptr && (reinterpret_cast<uintptr_t> (hist) & 7) == 0 && (hist_len & 1) == 0;
	auto in_f = [] (const float* q) -> float { return *q; };
	auto in_c = [] (const f2* q) -> f2 { return *q; };
	auto gl_c = [] (const f2* q) -> f2 { return *q; };
	auto ldv = [&] (const float* in_, int64_t n_in_, const float* hist_, int hist_len_, int64_t i) -> float {
		if (i >= 0) {
			return i < n_in_ ? in_f (in_ + i) : 0.f;
		}
		const int64_t h = i + hist_len_;
		if (hist_ == nullptr || h < 0) {
			return 0.f;
		}
		return piped ? __hip_atomic_load (const_cast<float*> (hist_) + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : in_f (hist_ + h);
	};
	auto row_pair = [&] (int64_t ar, int t) -> f2 {
		if (in_al8 && ar >= 0 && ar + 512 <= n_in) {
			return in_c (reinterpret_cast<const f2*> (in + ar) + t);
		}
		if (hist_al8 && ar + 512 <= 0 && ar + hist_len >= 0) {
			return in_c (reinterpret_cast<const f2*> (hist + (ar + hist_len)) + t);
		}
		if (ar >= n_in) { // a row of the silence behind the input (wave-uniform): seven of a 1024-frame run() call's sixteen rows, which
			return mkf2 (0.f, 0.f); // otherwise take the frame-by-frame route below, two bounds-checked loads with a branch each per lane
		}
		const int64_t a = ar + 2 * (int64_t)t;
		return mkf2 (ldv (in, n_in, hist, hist_len, a), ldv (in, n_in, hist, hist_len, a + 1));
	};

	for (int64_t s = r0; s < r1;) {
		const int64_t a0 = s - Dd - 512 * R0; // frame of (row 0, thread 0).q: the dry frame of output row r is input row r
		if (MODE != MODE_RAMP && in_al8 && (out_al8 || MODE == MODE_SWEEP) && a0 >= 1 && !(p.sv_flags & 1) &&
		    (MODE != MODE_SWEEP || (s >= p.sw_t_begin && s >= p.sw_dry_from))) {
			// blocks from here on that read only frames inside the array and are emitted whole
			const int64_t n1 = (r1 - s) / BLK;
			const int64_t n2 = n_in - 8192 - a0 >= 0 ? (n_in - 8192 - a0) / BLK + 1 : 0;
			int           nf = (int)(n1 < n2 ? n1 : n2);
			if (nf > 0) {
				nf                 = __builtin_amdgcn_readfirstlane (nf);
				const float* a_in  = uniform_ptr_generic (in + a0);
				float*       a_out = uniform_ptr_generic (out + s);
#ifdef PROT_SV_CLOCK
				const uint64_t c0 = __builtin_readcyclecounter (), q0 = __builtin_amdgcn_s_memrealtime ();
#endif
				if constexpr (MODE == MODE_SWEEP) {
					float2* const base  = sw_list ();
					float2*       cur   = uniform_ptr_generic (base + sw_cnt);
					const float   thr_u = sw_thr ();
					PROT_SV_SWEEP_CALL (8, a_in, cur, sw_raw, nf, thr_u, sa)
					PROT_SV_SWEEP_CALL (12, a_in, cur, sw_raw, nf, thr_u, sa)
					sw_cnt = (unsigned)__builtin_amdgcn_readfirstlane ((int)(cur - sw_list ()));
				} else {
					PROT_SV_ALL (ASM, a_in, a_out, nf, ca, sa)
				}
#ifdef PROT_SV_CLOCK
#ifndef PROT_SV_CLOCK_NOSUM // (the sums are four atomics at the end of every run: microseconds on a short launch -- tools/cfg3_timeline.py builds without)
				if (t0 == 0) {
					atomicAdd (&g_sv_clock[0], (unsigned long long)(__builtin_readcyclecounter () - c0));
					atomicAdd (&g_sv_clock[1], (unsigned long long)(__builtin_amdgcn_s_memrealtime () - q0));
					atomicAdd (&g_sv_clock[2], (unsigned long long)nf);
					atomicAdd (&g_sv_clock[3], 1ull);
				}
#endif
				wg_a0 = wg_a0 ? wg_a0 : q0;
				wg_a1 = __builtin_amdgcn_s_memrealtime ();
#endif
				s += (int64_t)nf * BLK;
				continue;
			}
		}
		// ---- one block through LDS
		int t = t0;
		asm volatile ("" : "+v"(t)); // (opaque per block: per-lane address terms are otherwise hoisted out of the loop)
		t &= 255;
		LdsC* const st   = (LdsC*)smem;               // rows: slot 256 row + t
		LdsF* const gcar = (LdsF*)(smem + 4096);      // gathered frames in the generated code's lane layout:
		LdsF* const gnx  = (LdsF*)(smem + 4096 + 128); //   lanes HS-NT..HS-1 <-> rows 0..NT-1, lanes 0..HS-1 <-> rows NT..15
		__syncthreads (); // (every wave has left the generated loop, if that is where it comes from)
		{
#ifdef PROT_NO_STAGE6
			if (false) {
#else
			if (MODE != MODE_SWEEP && !piped) {
#endif // (the sweep's instantiations have no registers for it: they spill as it is)
				// (round 6: a staged block is the WHOLE chain of its workgroup on a short launch -- cfg3's first and last block of each
				// channel took 9.2-9.7 us beside 6.1 for the blocks of the generated loop, with the sixteen rows fetched one after the
				// other behind wave-uniform branches.  Every frame by one unconditional load from an address chosen without a branch:
				// thirty-four loads in flight, one round trip)
				const float* const safe = a_sv; // (an address that can always be read)
				auto ldq = [&] (int64_t i) -> float {
					const int64_t h    = i + hist_len;
					const bool    ok_i = i >= 0 && i < n_in;
					const bool    ok_h = i < 0 && hist != nullptr && h >= 0;
					const float*  a    = ok_i ? in + i : (ok_h ? hist + h : safe);
					const float   x    = *a;
					return (ok_i || ok_h) ? x : 0.f;
				};
				f2 v[16];
#pragma unroll
				for (int k = 0; k < 16; ++k) {
					const int64_t a = a0 + 512 * (int64_t)k + 2 * (int64_t)t;
					v[k]            = mkf2 (ldq (a), ldq (a + 1));
				}
				const int   l = t & 63, w = t >> 6;
				const int   rc = l - (HS - NT), rn = NT + l; // the row this lane gathers for, as a carried / as a new row
				const bool  okc = rc >= 0 && rc < NT, okn = l < HS;
				const float gcv = ldq (a0 + 512 * (int64_t)(okc ? rc : 0) + 128 * w - 1);
				const float gnv = ldq (a0 + 512 * (int64_t)(okn ? rn : NT) + 128 * w - 1);
#pragma unroll
				for (int k = 0; k < 16; ++k) {
					st[256 * k + t] = v[k];
				}
				gcar[t] = okc ? gcv : 0.f;
				gnx[t]  = okn ? gnv : 0.f;
			} else {
#pragma unroll 2
				for (int k = 0; k < 16; ++k) {
					st[256 * k + t] = row_pair (a0 + 512 * (int64_t)k, t);
				}
				const int l = t & 63, w = t >> 6;
				const int rc = l - (HS - NT), rn = NT + l; // the row this lane gathers for, as a carried / as a new row
				gcar[t] = (rc >= 0 && rc < NT) ? ldv (in, n_in, hist, hist_len, a0 + 512 * (int64_t)rc + 128 * w - 1) : 0.f;
				gnx[t]  = (l < HS) ? ldv (in, n_in, hist, hist_len, a0 + 512 * (int64_t)rn + 128 * w - 1) : 0.f;
			}
		}
		__syncthreads ();
		{
			const float  cca = (MODE == MODE_RAMP || MODE == MODE_SWEEP) ? 0.f : ca, csa = MODE == MODE_RAMP ? __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (p.sv_gain))) : sa;
			const float* a_in  = a_sv; // (unused by the core)
			float*       a_out = nullptr;
			const int    one   = 1;
			PROT_SV_ALL (CORE, a_in, a_out, one, cca, csa)
		}
		__syncthreads ();
		// The thread id opaque AGAIN: nothing per-lane may be kept alive across the generated code (which owns v0..v119) except
		// the thread id itself -- the compiler otherwise carries hoisted per-lane addresses across it through scratch, and a
		// spill placed inside one of the divergent branches above comes back with the inactive lanes' slots stale
		// (seen: the per-frame rotation of lanes >= 2 read at index 0)
		t = t0;
		asm volatile ("" : "+v"(t));
		t &= 255;
		// rows 0..HS-1 of the exchange area now hold the output pairs of frames s + 512 k + 2 t (+1)
		if constexpr (MODE == MODE_SWEEP) {
			// the sweep's test on an edge block: the same order as the generated loop (row, first / second frame of the pairs,
			// lane), with what only edge blocks meet -- frames in front of sw_t_begin or beyond the call are not tested, the dry
			// term is 0 while t < sw_dry_from (Q1; the raw peak takes the frame all the same)
			float2* const list    = sw_list ();
			const float   sw_thr2 = sw_thr ();
			for (int k = 0; k < HS; ++k) {
				const f2      hp   = st[256 * k + t];
				const int64_t pos0 = s + 512 * (int64_t)k + 2 * (int64_t)t;
				const f2      dp   = row_pair (pos0 - 2 * (int64_t)t - Dd, t);
#pragma unroll
				for (int e = 0; e < 2; ++e) {
					const int64_t pos = pos0 + e;
					const bool    in_ = pos >= p.sw_t_begin && pos < r1;
					const float   raw = e ? dp.y : dp.x, h = e ? hp.y : hp.x;
					const float   d   = pos >= p.sw_dry_from ? raw : 0.f;
					if (in_) {
						sw_raw = fmaxf (sw_raw, fabsf (raw));
					}
					const bool     live = in_ && d * d + h * h > sw_thr2;
					const uint64_t mask = __builtin_amdgcn_ballot_w64 (live);
					if (live) {
						list[sw_cnt + __builtin_amdgcn_mbcnt_hi ((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo ((unsigned)mask, 0u))] = make_float2 (d, h);
					}
					sw_cnt += (unsigned)__builtin_popcountll (mask);
				}
			}
		} else if (MODE != MODE_RAMP && s + BLK <= r1 && out_al8) {
#pragma unroll
			for (int m = 0; m < HS / 2; ++m) {
				PROT_GLOBAL f2* __restrict__ ob = reinterpret_cast<PROT_GLOBAL f2*> (uniform_ptr (out + s + 1024 * m));
#pragma unroll
				for (int e = 0; e < 2; ++e) {
					ob[(unsigned)t + 256u * e] = st[256 * (2 * m + e) + t];
				}
			}
		} else {
#pragma unroll 2
			for (int k = 0; k < HS; ++k) {
				f2            o    = st[256 * k + t];
				const int64_t pos0 = s + 512 * (int64_t)k + 2 * (int64_t)t;
				if (MODE == MODE_RAMP && pos0 < r1) {
					// per-frame rotation: the dry pair of output row k is the (q, c) pair of input row R0 + k
					const f2 d   = row_pair (pos0 - 2 * (int64_t)t - Dd, t);
					const f2 cs0 = gl_c (reinterpret_cast<const f2*> (p.rot_s) + ((int64_t)c * p.n_out + pos0));
					const f2 cs1 = pos0 + 1 < r1 ? gl_c (reinterpret_cast<const f2*> (p.rot_s) + ((int64_t)c * p.n_out + pos0 + 1)) : mkf2 (0.f, 0.f);
					o            = mkf2 (cs0.x * d.x + cs0.y * o.x, cs1.x * d.y + cs1.y * o.y);
				}
				if (pos0 < r1) {
					out[pos0] = o.x;
				}
				if (pos0 + 1 < r1) {
					out[pos0 + 1] = o.y;
				}
			}
		}
		__syncthreads (); // (the output rows have been read: the exchange area is free for whatever comes next)
		s += BLK;
	}
#ifdef PROT_SV_CLOCK
	if (t0 == 0 && blockIdx.x < 4096) {
		g_sv_wg[4 * blockIdx.x]     = wg_t0;
		g_sv_wg[4 * blockIdx.x + 1] = wg_a0;
		g_sv_wg[4 * blockIdx.x + 2] = wg_a1;
		g_sv_wg[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime ();
	}
#endif
	if (!piped) {
		if (p.hist_extra == 0) {
			hist_write_slice (p, c, threadIdx.x);
		}
	}
	if constexpr (MODE == MODE_SWEEP) {
		if ((t0 & 63) == 0) {
			p.surv_cnt[((int64_t)c * gridDim.x + blockIdx.x) * 4 + (t0 >> 6)] = sw_cnt;
		}
		if (p.sw_raw_slot >= 0) {
			sw_raw = sw_wave_reduce (sw_raw, true);
			if ((t0 & 63) == 0 && sw_raw > 0.f) { // (non-negative floats order as their bit patterns; look before the atomic)
				unsigned* const slot = reinterpret_cast<unsigned*> (p.sw_peaks + (int64_t)c * kMaxSample + p.sw_raw_slot);
				if (__float_as_uint (sw_raw) > __hip_atomic_load (slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
					atomicMax (slot, __float_as_uint (sw_raw));
				}
			}
		}
	}
}

template <int HS, int MODE>
__global__ __launch_bounds__ (256, 4) void olasave_kernel (const OlaParams p)
{
#ifndef PROT_NO_KPREFETCH
	kernarg_prefetch ();
#endif
	__shared__ f2    smem[kLdsExchange + 512]; // the exchange area + the two W256 tables (written by the generated prologue)
	__shared__ float sw_lo_s[4];
	olasave_body<HS, MODE> (p, smem, sw_lo_s);
}

constexpr int kHistExtra = 4, kHistExtraMaxRuns = 200;

template <int HS>
static hipError_t launch_sv_m (const OlaParams& p, dim3 grid, hipStream_t s)
{
	const dim3 block (256);
	if (p.surv != nullptr) {
		if constexpr (HS == 8 || HS == 12) {
			hipLaunchKernelGGL ((olasave_kernel<HS, MODE_SWEEP>), grid, block, 0, s, p);
		} else {
			return hipErrorInvalidValue;
		}
	} else if (p.hilbert_only) {
		hipLaunchKernelGGL ((olasave_kernel<HS, MODE_HILBERT>), grid, block, 0, s, p);
	} else if (p.rot_s != nullptr) {
		hipLaunchKernelGGL ((olasave_kernel<HS, MODE_RAMP>), grid, block, 0, s, p);
	} else {
		hipLaunchKernelGGL ((olasave_kernel<HS, MODE_ROT>), grid, block, 0, s, p);
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
	const int64_t unit = p.run_stride > 0 ? p.run_stride : p.run_frames;
	const int64_t runs = (p.n_out + unit - 1) / unit;
	dim3          grid ((unsigned)runs, (unsigned)n_chn);
	OlaParams     q = p;
	// a short launch (fewer workgroups than CUs: every block is a lone dependent chain) that also writes the next history: four more
	// workgroups per channel do that, beside the blocks instead of behind each of them (cfg3: 2.5 us of an 11 us launch)
#ifndef PROT_NO_HISTEXTRA
	if (p.hist_out != nullptr && p.hs_flag == nullptr && p.surv == nullptr && p.run_stride == 0 && !p.hilbert_only && runs * n_chn <= kHistExtraMaxRuns)
#else
	if (false)
#endif
	{
		q.hist_extra = kHistExtra;
		grid.x += kHistExtra;
	}
	switch (hop_rows * 4 + n_rho_rows) {
		case 8 * 4 + 1: return launch_sv_m<8> (q, grid, s);
		case 10 * 4 + 1: return launch_sv_m<10> (q, grid, s);
		case 12 * 4 + 1: return launch_sv_m<12> (q, grid, s);
		default: return hipErrorInvalidValue;
	}
}

} // namespace prot
