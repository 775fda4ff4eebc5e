// prot_dev.h -- device helpers shared by the OLA kernels (prot_kernels.hip, prot_olasave.hip): the virtual input stream
// [history | input | silence], SGPR-pinned stream pointers, the fused history write, the LDS exchange geometry.
#pragma once
#include "prot_fft.h"
#include "prot_kernels.h"

namespace prot {

enum { MODE_ROT = 0, MODE_RAMP = 1, MODE_HILBERT = 2, MODE_SWEEP = 3 };

// ------------------------------------------------------------------ helpers
// virtual stream [hist | in | silence] at index i (i may be negative)
template <class S>
PROT_DEV float ldv (const S* __restrict__ in, int64_t n_in, const S* __restrict__ hist, int hist_len, int64_t i)
{
	if (i >= 0) {
		return i < n_in ? (float)in[i] : 0.f;
	}
	const int64_t h = i + hist_len;
	return (hist != nullptr && h >= 0) ? (float)hist[h] : 0.f;
}

// pin a wave-uniform pointer in an SGPR pair: the loads/stores that use it take the
// "saddr + 32-bit lane offset + immediate" form (half the address traffic of 64-bit per-lane
// addresses and no VALU pointer arithmetic); opaque to loop strength reduction, which would
// otherwise turn the bases back into per-lane pointer induction variables
#define PROT_GLOBAL __attribute__ ((address_space (1)))
template <class P>
PROT_DEV PROT_GLOBAL P* uniform_ptr (P* q)
{
	const uint64_t b  = reinterpret_cast<uint64_t> (q);
	const uint32_t lo = __builtin_amdgcn_readfirstlane ((uint32_t)b);
	const uint32_t hi = __builtin_amdgcn_readfirstlane ((uint32_t)(b >> 32));
	return (PROT_GLOBAL P*)(((uint64_t)hi << 32) | lo);
}

// the same for a plain (generic address space) pointer: an SGPR pair for inline-asm "s" operands
template <class P>
PROT_DEV P* uniform_ptr_generic (P* q)
{
	const uint64_t b  = reinterpret_cast<uint64_t> (q);
	const uint32_t lo = __builtin_amdgcn_readfirstlane ((uint32_t)b);
	const uint32_t hi = __builtin_amdgcn_readfirstlane ((uint32_t)(b >> 32));
	return (P*)(((uint64_t)hi << 32) | lo);
}

// Fused history advance: workgroup b of a channel copies slice b of the next call's history (the last
// hist_len frames of [hist | in | silence]) -- saves the separate history kernel and its launch gap on
// small stateful launches (LV2 run(), batched buffers).  Called by every workgroup exactly once: on its early
// return, or at the very END of the kernel (after the main loop, when the loop's registers are dead -- called
// up front it perturbed the allocation of kernels that sit at the 256-VGPR limit into spilling inside the loop).
template <class S>
PROT_DEV void hist_write_slice (const OlaParamsT<S>& p, int c, int t)
{
	if (p.hist_out == nullptr) {
		return;
	}
	const int per = (p.hist_len + (int)gridDim.x - 1) / (int)gridDim.x;
	const int i0  = (int)blockIdx.x * per;
	const int i1  = i0 + per < p.hist_len ? i0 + per : p.hist_len;
	const S* __restrict__ in   = p.in + (int64_t)c * p.in_stride;
	const S* __restrict__ hist = p.hist ? p.hist + (int64_t)c * p.hist_len : nullptr;
	S* __restrict__ ho         = p.hist_out + (int64_t)c * p.hist_len;
	for (int i = i0 + t; i < i1; i += 256) {
		ho[i] = (S)ldv (in, p.n_in, hist, p.hist_len, p.n_in - p.hist_len + i);
	}
}

// Part `part` of `nparts` of a channel's next history, hist_out[i] = [hist | in | silence](n_in - hist_len + i), eight loads in flight per
// thread and no branch around a load (short launches, round 6: the workgroups that do this do nothing else -- OlaParams::hist_extra)
// (DEVICE_SCOPE: write-through stores -- a pipelined launch's history is read by the next launch while this one still runs)
template <bool DEVICE_SCOPE = false>
PROT_DEV void hist_write_part (const OlaParamsT<float>& p, const int c, const int part, const int nparts)
{
	if (p.hist_out == nullptr) {
		return;
	}
	const int per = (((p.hist_len + nparts - 1) / nparts) + 3) & ~3;
	const int i0  = part * per;
	const int i1  = i0 + per < p.hist_len ? i0 + per : p.hist_len;
	const float* __restrict__ in   = p.in + (int64_t)c * p.in_stride;
	const float* __restrict__ hist = p.hist ? p.hist + (int64_t)c * p.hist_len : nullptr;
	float* __restrict__ ho         = p.hist_out + (int64_t)c * p.hist_len;
	const int nthr = (int)blockDim.x;
	if (p.n_in <= 0) {
		for (int i = i0 + (int)threadIdx.x; i < i1; i += (int)blockDim.x) {
			ho[i] = ldv (in, p.n_in, hist, p.hist_len, p.n_in - p.hist_len + i);
		}
		return;
	}
	for (int base = i0 + (int)threadIdx.x; base < i1; base += nthr * 8) {
		float v[8];
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			const int     i  = base + nthr * q < i1 ? base + nthr * q : i1 - 1;
			const int64_t f  = p.n_in - p.hist_len + i; // frame of the virtual stream (< n_in)
			const bool    hs = f < 0 && hist != nullptr && f + p.hist_len >= 0;
			const float*  a  = hs ? hist + (f + p.hist_len) : in + (f >= 0 ? f : 0);
			const float   x  = *a;
			v[q]             = (f >= 0 || hs) ? x : 0.f;
		}
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			if (base + nthr * q < i1) {
				if constexpr (DEVICE_SCOPE) {
					__hip_atomic_store (&ho[base + nthr * q], v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				} else {
					ho[base + nthr * q] = v[q];
				}
			}
		}
	}
}

// Every 64-byte line of the kernel arguments asked for at once: OlaParams is seven lines and the compiler loads its fields where
// they are used, each first touch of a line a miss that the next branch waits for -- 1.8 us from the kernel's entry to its first
// block on a short launch (tools/cfg3_timeline.py).  The loaded words are thrown away; the loads behind them hit the scalar cache.
PROT_DEV void kernarg_prefetch ()
{
	const uint64_t ka = (uint64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr ();
	unsigned d0, d1, d2, d3, d4, d5, d6;
	static_assert (sizeof (OlaParamsT<float>) <= 7 * 64 && sizeof (OlaParamsT<float>) > 6 * 64, "one load per line of the arguments");
	asm volatile ("s_load_dword %0, %7, 0x0\n\ts_load_dword %1, %7, 0x40\n\ts_load_dword %2, %7, 0x80\n\ts_load_dword %3, %7, 0xc0\n\t"
	              "s_load_dword %4, %7, 0x100\n\ts_load_dword %5, %7, 0x140\n\ts_load_dword %6, %7, 0x180\n\t"
	              "s_waitcnt lgkmcnt(0)" // (the words arrive in registers the compiler believes written HERE: without the wait it reuses them at once
	                                     // and the late words land on live values -- seen as a crash)
	              : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6) // (early clobber: a word that arrives while the
	                // later loads are still being issued -- a busy scalar cache on a big launch -- must not land on the address: seen as an aperture violation)
	              : "s"(ka));
}

// The run [r0, r0 + rlen) of output frames this workgroup emits.  One-round launches (every workgroup resident at once,
// p.rank_cu > 0) give the runs unequal lengths by DISPATCH RANK: the hardware arbitrates oldest-first, so of the workgroups
// of a CU the one dispatched first runs ahead and the last one finishes alone; workgroups are dealt out in id order, one
// per CU per round, so id / CUs is the rank (if a dispatcher ever does it differently the lengths are merely not ideal --
// the runs still tile the stream).  Which run: workgroups are dealt round-robin over the 8 XCDs, so XCD k would own the
// runs k, k + 8, ... -- rotating the position inside each group of eight consecutive runs by the group's number gives every
// XCD all eight residues in turn (-2 %, measured on olapoly_kernel).
template <class S>
PROT_DEV void run_of_workgroup (const OlaParamsT<S>& p, int64_t& r0, int64_t& rlen)
{
	int       bx = (int)blockIdx.x;
	const int gx = (int)gridDim.x - p.hist_extra; // workgroups with a run (per channel)
	if ((bx | 7) < gx) {
		bx = (bx & ~7) | ((bx + (bx >> 3)) & 7);
	}
	rlen = p.run_frames;
	if (p.run_stride > 0) { // a sparse sample of the stream (the sweep's bound-seeding pass): run g starts at g * run_stride
		r0 = (int64_t)blockIdx.x * p.run_stride;
		return;
	}
	if (p.rank_cu > 0) {
		const int id = (int)blockIdx.y * gx + bx, id0 = (int)blockIdx.y * gx;
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
}

// ------------------------------------------------------------ LDS exchange geometry of the 4096-point transform
// 16 x 16 x 16: thread t keeps 16 complex values; the exchange between pass 1 and 2 (and its inverse mirror) goes through
// slots k * kL1 + t (workgroup-wide, one barrier), the one between pass 2 and 3 is a 16 x 16 transposition inside each
// 16-lane group, in the group's own chunk [th * kL1, th * kL1 + 272) with row stride kG (wave-local: DS instructions of one
// wave execute in order, no s_barrier).  Strides chosen conflict-free for ds_write_b64 / ds_read_b64.
constexpr int kL1 = 272;
constexpr int kG  = 17;
constexpr int kLdsExchange = 16 * kL1; // complex slots

// exchange accesses are volatile, address-space-3 pointers: the backend keeps one ds_read_b64 / ds_write_b64 per element in
// source order instead of fusing neighbours into ds_read2_b64 / ds_write2_b64 (8 and 13 LDS cycles against 2 x 2 and 2 x 6)
#define PROT_LDS_PTR(T) volatile __attribute__ ((address_space (3))) typename T::cplx

PROT_DEV void wave_sync ()
{
	__builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier ();
	__builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
}

} // namespace prot
