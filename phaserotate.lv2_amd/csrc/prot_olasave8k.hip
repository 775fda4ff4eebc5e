// prot_olasave8k.hip -- olasave8k_kernel: the overlap-save, table-free polyphase block of prot_olasave.hip on an 8192-POINT
// transform pair (round 4).  512 threads (8 waves) per workgroup, 16 points per thread, two workgroups per CU (<= 128
// VGPRs, 78 KiB of LDS): the same 16 waves per CU as the 4096-point kernel, but a block is 16 rows of 512 complex samples
// of which NT = firlen / 1024 are context -- 13 of 16 rows new for the LV2 48 kHz filter (10 of 16 on 4096 points), 12 of
// 16 at 96 kHz (8 of 16), 14 of 16 for CLI blksiz 2048 (12 of 16): fewer VALU issue cycles and LDS instructions per frame,
// the resource that bounds this path.
//
// What it replaces: the block body of process_channel (src/phaserotate.c:629-720) and PhaseRotateProc::hilbert / rotate
// (cli/phase-rotate.cc:181-232), exactly as olasave_kernel does.  The transform is 16 x 2 x 16 x 16: the radix-2 layer over
// the top bit of the thread index is taken by the READER of the first LDS exchange (forward) and of the last one (inverse);
// tools/model/olasave_model8k.py is the numpy model, tools/gen/gen_olasave8k.py the generator of ALL the arithmetic
// (prot_olasave8k_asm.inc: a streamed run of blocks, and one LDS-staged block for edges); this file moves data.
//
// Blocks with an odd number of context rows (LV2 48 kHz: 3) are kept circularly rotated by half a row, so that the valid
// outputs are whole rows for every thread: position p of the block holds local sample (p + 256) mod 8192.
#include "prot_dev.h"

namespace prot {

typedef volatile __attribute__ ((address_space (3))) f2 LdsC8;
typedef volatile __attribute__ ((address_space (3))) float LdsF8;

#ifdef PROT_SV8_INC // (tuning builds: a variant of the generated code, tools/ab_build.py sv8:...)
#include PROT_SV8_INC
#else
#include "prot_olasave8k_asm.inc"
#endif

constexpr int kSv8LdsBytes = 16 * 560 * 8 + 8192; // exchange area (16 rows of 560 slots) + the two 512-entry twiddle tables

#define PROT_SV8_CAT_(k, a, b) PROT_SV8_##k##_##a##_##b
#define PROT_SV8_CLOB_(a) PROT_SV8_ASM_CLOBBERS_##a
#if defined(PROT_SV8_STAMPS) && !defined(PROT_SV8_TIMELINE)
#define PROT_SV8_TIMELINE
#endif
#ifdef PROT_SV8_TIMELINE
// diagnostic builds (tools/sv8_stamps.py; -DPROT_SV8_TIMELINE alone: the per-workgroup timeline around the SHIPPED loop, nothing inside it; python3 tools/gen/gen_olasave8k.py --variant stamps --opts stamps, then
// tools/ab_build.py sv8:stamps:-DPROT_SV8_STAMPS): the generated loop stamps the shader clock at 15 points of a block (around
// each of its four barriers and its waits) and sums the intervals per wave in the lanes of one more operand; per-workgroup
// entry / end times beside it.  Nothing of this exists in the shipped library.
__device__ unsigned g_sv8_acc[2048 * 8 * 16];         // per workgroup and wave: [0] blocks, [1..14] cycles per interval (stamped loop), [15] 1
extern "C" int prot_debug_sv8_stamps (unsigned long long* out, int n, int reset)
{
	// out[1..14] = cycles per interval summed over the waves of the launches since the last reset is NOT kept any more: the slots hold
	// the LAST launch (every wave overwrites its own); out[16] = wave-blocks, out[17] = waves of that launch
	static unsigned h[2048 * 8 * 16];
	if (reset) {
		for (auto& x : h) {
			x = 0u;
		}
		return (int)hipMemcpyToSymbol (HIP_SYMBOL (g_sv8_acc), h, sizeof (h));
	}
	hipError_t e = hipMemcpyFromSymbol (h, HIP_SYMBOL (g_sv8_acc), sizeof (h));
	unsigned long long sum[32] = {};
	for (int i = 0; i < 2048 * 8; ++i) {
		if (h[16 * i + 15] == 0u || h[16 * i] == 0u) {
			continue;
		}
		for (int l = 1; l < 15; ++l) {
			sum[l] += h[16 * i + l];
		}
		sum[16] += h[16 * i];
		sum[17] += 1;
	}
	for (int i = 0; i < n && i < 32; ++i) {
		out[i] = sum[i];
	}
	return (int)e;
}
__device__ unsigned g_sv8_prog[64 * 2048];           // "tick" variant of the loop: per workgroup, lane k = 100 MHz time at which the block with k blocks left ended
extern "C" int prot_debug_sv8_prog (unsigned* out, int n_wg)
{
	return (int)hipMemcpyFromSymbol (out, HIP_SYMBOL (g_sv8_prog), sizeof (unsigned) * 64 * (n_wg < 2048 ? n_wg : 2048));
}
__device__ unsigned long long g_sv8_wg[4 * 2048];    // per workgroup: entry, loop start, end (s_memrealtime, 100 MHz), XCC_ID << 32 | HW_ID
extern "C" int prot_debug_sv8_wg (unsigned long long* out, int n_wg)
{
	return (int)hipMemcpyFromSymbol (out, HIP_SYMBOL (g_sv8_wg), sizeof (unsigned long long) * 4 * (n_wg < 2048 ? n_wg : 2048));
}
#endif
#ifdef PROT_SV8_STAMPS
#define PROT_SV8_CALL(KIND_, HS_, AM_, in_, out_, nf_, ca_, sa_)                                                                       \
	if (HS == HS_ && AM == AM_) {                                                                                                       \
		asm volatile (PROT_SV8_CAT_ (KIND_, HS_, AM_)                                                                                   \
		              : "+v"(sv8_acc)                                                                                                   \
		              : "s"(in_), "s"(out_), "s"(nf_), "s"(lds_off), "s"(a_sv), "s"(ca_), "s"(sa_), "v"(t0)                                \
		              : PROT_SV8_CLOB_ (HS_));                                                                                          \
	}
#else
#define PROT_SV8_CALL(KIND_, HS_, AM_, in_, out_, nf_, ca_, sa_)                                                                       \
	if (HS == HS_ && AM == AM_) {                                                                                                       \
		asm volatile (PROT_SV8_CAT_ (KIND_, HS_, AM_)                                                                                   \
		              :                                                                                                                 \
		              : "s"(in_), "s"(out_), "s"(nf_), "s"(lds_off), "s"(a_sv), "s"(ca_), "s"(sa_), "v"(t0)                                \
		              : PROT_SV8_CLOB_ (HS_));                                                                                          \
	}
#endif
#define PROT_SV8_ALL(KIND_, in_, out_, nf_, ca_, sa_)                                                                                  \
	PROT_SV8_CALL (KIND_, 8, 0, in_, out_, nf_, ca_, sa_)                                                                               \
	PROT_SV8_CALL (KIND_, 8, 2, in_, out_, nf_, ca_, sa_)                                                                               \
	PROT_SV8_CALL (KIND_, 12, 0, in_, out_, nf_, ca_, sa_)                                                                              \
	PROT_SV8_CALL (KIND_, 12, 2, in_, out_, nf_, ca_, sa_)                                                                              \
	PROT_SV8_CALL (KIND_, 13, 0, in_, out_, nf_, ca_, sa_)                                                                              \
	PROT_SV8_CALL (KIND_, 13, 2, in_, out_, nf_, ca_, sa_)

// MODE_SWEEP: the generated loop with the sweep's test in its epilogue (no output rows: survivors appended at cursor_, the running
// max |x| in raw_); cursor_ and raw_ are read and written (prot_olasave.hip's PROT_SV_SWEEP_CALL)
#define PROT_SV8_SWEEP_CALL(HS_, in_, cursor_, raw_, nf_, thr_, gain_)                                                                 \
	if (HS == HS_) {                                                                                                                    \
		asm volatile (PROT_SV8_SWEEP_##HS_                                                                                              \
		              : "=s"(cursor_), "=v"(raw_)                                                                                       \
		              : "s"(in_), "s"(nf_), "s"(lds_off), "s"(a_sv), "s"(thr_), "s"(gain_), "v"(t0), "s"(cursor_), "v"(raw_)             \
		              : PROT_SV8_SWEEP_CLOBBERS_##HS_);                                                                                 \
	}

// HS new rows (1024 HS frames) per block; MODE_ROT / MODE_RAMP / MODE_HILBERT / MODE_SWEEP (HS = 8: the fused sweep of the CLI's
// blksiz 8192 engine -- the Hilbert branch tested against the call's bound, survivors listed per wave: prot_olasave.hip).  Same division of labour as olasave_kernel:
// consecutive whole in-range blocks are ONE call of the generated loop; every other block is staged here, position by
// position, into the LDS exchange area, transformed by the generated CORE (the same instructions on the same values: the
// two paths agree bit for bit) and stored from LDS with bounds.
template <int HS, int MODE>
__global__ __launch_bounds__ (512, 4) void olasave8k_kernel (const OlaParams p)
{
	constexpr int BLK = 1024 * HS;           // output frames per block
	constexpr int NT  = 16 - HS;             // context rows (of 512 complex samples = 1024 frames)
	constexpr int R0  = NT / 2;              // first valid output row = first dry row
	constexpr int ROT = (NT & 1) ? 256 : 0;  // complex samples the block is rotated by
	constexpr int AM  = MODE == MODE_ROT ? 0 : 2;
	static_assert (HS == 8 || HS == 12 || HS == 13, "rows per block");
	static_assert (MODE != MODE_SWEEP || HS == 8 || HS == 12, "the fused sweep exists for the CLI engines on these blocks (blksiz 8192, 4096)");

	extern __shared__ f2 smem8[]; // kSv8LdsBytes: the exchange area + the two tables (written by the generated prologue)

	const int t0 = threadIdx.x;
	const int c  = blockIdx.y;
#ifdef PROT_SV8_TIMELINE
	unsigned sv8_acc = 0u, sv8_blocks = 0u;
	(void)sv8_acc;
	if (t0 == 0 && blockIdx.y == 0 && blockIdx.x < 2048) { // (stored at once: nothing of the timeline lives across the generated code)
		g_sv8_wg[4 * blockIdx.x]     = __builtin_amdgcn_s_memrealtime ();
		g_sv8_wg[4 * blockIdx.x + 1] = 0ull;
	}
#endif
	// MODE_SWEEP state (as in prot_olasave.hip: the bound parked per wave in global memory behind the list lengths and re-read where
	// it is used, nothing per lane but the running maximum and nothing uniform but the count lives across the generated code)
	__shared__ float sw_lo_s[8];
	float            sw_raw = 0.f;
	unsigned         sw_cnt = 0u;
	auto sw_thr_slot = [&] () -> float* {
		const int w = __builtin_amdgcn_readfirstlane (threadIdx.x >> 6);
		return reinterpret_cast<float*> (p.surv_cnt) + (int64_t)gridDim.y * gridDim.x * 8 + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + w;
	};
	auto sw_thr = [&] () -> float {
		return __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (__hip_atomic_load (sw_thr_slot (), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))));
	};
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
		return uniform_ptr_generic (p.surv + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + w) * p.surv_cap);
	};
	if (MODE == MODE_SWEEP) {
		float lo = 1.0e18f;
		if (!p.sw_thr_zero) {
			for (int i = t0; i < p.sw_n_angles; i += 512) {
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
		lo = fminf (fminf (fminf (sw_lo_s[0], sw_lo_s[1]), fminf (sw_lo_s[2], sw_lo_s[3])), fminf (fminf (sw_lo_s[4], sw_lo_s[5]), fminf (sw_lo_s[6], sw_lo_s[7])));
		__syncthreads ();
		if ((t0 & 63) == 0) {
			__hip_atomic_store (sw_thr_slot (), lo * lo * 0.9999f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
	}
	int64_t   r0, rlen;
	run_of_workgroup (p, r0, rlen);
	if (r0 >= p.n_out) {
		if (t0 < 256) {
			hist_write_slice (p, blockIdx.y, t0);
		}
		if (MODE == MODE_SWEEP && (t0 & 63) == 0) {
			p.surv_cnt[((int64_t)c * gridDim.x + blockIdx.x) * 8 + (t0 >> 6)] = 0u;
		}
		return;
	}
#ifdef PROT_SV8_CYCLIC
	// TIMING EXPERIMENT (tools/ab_build.py sv8:cyc:-DPROT_SV8_CYCLIC with the generator's "cyclic" loop; results wrong): workgroup
	// b takes blocks 2 + b, 2 + b + 512, ... -- the resident workgroups read and write ONE dense window that moves through the
	// stream -- instead of a run of its own; the same number of blocks for every workgroup, the remainder of the stream untouched
	const int64_t cyc_per = (p.n_out / BLK - 8) / 512;
	r0                    = (2 + (int64_t)blockIdx.x) * BLK;
	const int64_t r1      = r0 + cyc_per * BLK;
#else
	const int64_t r1 = (r0 + rlen < p.n_out) ? r0 + rlen : p.n_out;
#endif

	const float* __restrict__ in   = p.in + (int64_t)c * p.in_stride;
	const float* __restrict__ hist = p.hist ? p.hist + (int64_t)c * p.hist_len : nullptr;
	float* __restrict__ out        = p.out + (int64_t)c * p.out_stride;
	const int64_t n_in = p.n_in;
	const int     Dd   = p.dry_delay; // even (host guarantees)
	const bool in_al8  = (reinterpret_cast<uintptr_t> (in) & 7) == 0;
	const bool out_al8 = (reinterpret_cast<uintptr_t> (out) & 7) == 0;

	if (MODE == MODE_SWEEP && p.sw_raw_slot >= 0 && p.sw_raw_tail && blockIdx.x == 0) {
		// the call's last dry_delay input frames: the rows reach them only as the next chunk's history (see prot_sweep.hip)
		for (int64_t i = n_in - Dd + t0; i < n_in; i += 512) {
			sw_raw = fmaxf (sw_raw, i >= 0 ? fabsf (in[i]) : 0.f);
		}
	}
	float ca = 0.f, sa = p.sv_gain;
	if (MODE == MODE_ROT) {
		ca = p.rot ? p.rot[2 * c] : p.rot_v[2 * c];
		sa = (p.rot ? p.rot[2 * c + 1] : p.rot_v[2 * c + 1]) * p.sv_gain;
	}
	ca = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (ca)));
	sa = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (sa)));
	const unsigned lds_off = (unsigned)(uintptr_t)(__attribute__ ((address_space (3))) f2*)smem8;
	const float*   a_sv    = uniform_ptr_generic (p.sv);

	const bool hist_al8 = hist != nullptr && (reinterpret_cast<uintptr_t> (hist) & 7) == 0 && (p.hist_len & 1) == 0;
	// the (x[a], x[a+1]) pair of lane t (0..255) in the 512-frame half row that starts at frame ar: a half row lies wholly in
	// the input array or wholly in the history (both start on half-row boundaries) -- aligned 8-byte loads; anything else
	// (silence in reach, odd alignment) frame by frame through the virtual stream [history | input | silence]
	auto half_pair = [&] (int64_t ar, int t) -> f2 {
		if (in_al8 && ar >= 0 && ar + 512 <= n_in) {
			return reinterpret_cast<const f2*> (in + ar)[t];
		}
		if (hist_al8 && ar + 512 <= 0 && ar + p.hist_len >= 0) {
			return reinterpret_cast<const f2*> (hist + (ar + p.hist_len))[t];
		}
		const int64_t a = ar + 2 * (int64_t)t;
		return mkf2 (ldv (in, n_in, hist, p.hist_len, a), ldv (in, n_in, hist, p.hist_len, a + 1));
	};

	for (int64_t s = r0; s < r1;) {
		// frame of local sample 0's aligned pair: the dry frame of output position p is input position p, and output row R0
		// starts at local sample 512 R0 + ROT
		const int64_t a0 = s - Dd - 1024 * R0 - 2 * ROT;
		if (MODE != MODE_RAMP && in_al8 && (out_al8 || MODE == MODE_SWEEP) && a0 >= 1 && !(p.sv_flags & 1) &&
		    (MODE != MODE_SWEEP || (s >= p.sw_t_begin && s >= p.sw_dry_from))) {
			// blocks from here on that read only frames inside the array (local samples 0..8191 and the frame in front) and
			// are emitted whole
			const int64_t n1 = (r1 - s) / BLK;
			const int64_t n2 = n_in - 16384 - a0 >= 0 ? (n_in - 16384 - a0) / BLK + 1 : 0;
			int           nf = (int)(n1 < n2 ? n1 : n2);
			if (nf > 0) {
				nf                 = __builtin_amdgcn_readfirstlane (nf);
				const float* a_in  = uniform_ptr_generic (in + a0);
				if constexpr (MODE == MODE_SWEEP) {
					float2* const base  = sw_list ();
					float2*       cur   = uniform_ptr_generic (base + sw_cnt);
					const float   thr_u = sw_thr ();
					PROT_SV8_SWEEP_CALL (8, a_in, cur, sw_raw, nf, thr_u, sa)
					PROT_SV8_SWEEP_CALL (12, a_in, cur, sw_raw, nf, thr_u, sa)
					sw_cnt = (unsigned)__builtin_amdgcn_readfirstlane ((int)(cur - sw_list ()));
				} else {
					float* a_out = uniform_ptr_generic (out + s);
#ifdef PROT_SV8_TIMELINE
					if (t0 == 0 && blockIdx.y == 0 && blockIdx.x < 2048 && sv8_blocks == 0u) {
						g_sv8_wg[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime ();
					}
#endif
					PROT_SV8_ALL (ASM, a_in, a_out, nf, ca, sa)
#ifdef PROT_SV8_TIMELINE
					sv8_blocks += (unsigned)nf;
#endif
				}
				s += (int64_t)nf * BLK;
				continue;
			}
		}
		// ---- one block through LDS
		int t = t0;
		asm volatile ("" : "+v"(t)); // (opaque per block: per-lane address terms are otherwise hoisted out of the loop)
		t &= 511;
		LdsC8* const st = (LdsC8*)smem8;            // positions: slot 512 row + T
		LdsF8* const gc = (LdsF8*)(smem8 + 8192);   // byte 65536: the frame in front of each wave's first sample of row (lane), lanes 0..15
		__syncthreads (); // (every wave has left the generated loop, if that is where it comes from)
		{
			const int h = t >> 8, tl = t & 255;
#pragma unroll 2
			for (int k = 0; k < 16; ++k) {
				const int nloc = (512 * k + 256 * h + ROT) & 8191; // first local sample of this half row
				st[512 * k + t] = half_pair (a0 + 2 * (int64_t)nloc, tl);
			}
			const int l = t & 63, w = t >> 6;
			const int nl = (512 * (l & 15) + 64 * w + ROT) & 8191;
			gc[t]        = l < 16 ? ldv (in, n_in, hist, p.hist_len, a0 + 2 * (int64_t)nl - 1) : 0.f;
		}
		__syncthreads ();
		{
			const float  cca = (MODE == MODE_RAMP || MODE == MODE_SWEEP) ? 0.f : ca, csa = MODE == MODE_RAMP ? __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (p.sv_gain))) : sa;
			const float* a_in  = a_sv; // (unused by the core)
			float*       a_out = nullptr;
			const int    one   = 1;
			PROT_SV8_ALL (CORE, a_in, a_out, one, cca, csa)
		}
		__syncthreads ();
		// (the thread id opaque again: nothing per-lane may live across the generated code, see prot_olasave.hip)
		t = t0;
		asm volatile ("" : "+v"(t));
		t &= 511;
		// rows 0..HS-1 of the staging area now hold the output pairs of frames s + 1024 k + 2 t (+1)
		if constexpr (MODE == MODE_SWEEP) {
			// the sweep's test on an edge block (prot_olasave.hip): frames in front of sw_t_begin or beyond the call are not tested, the
			// dry term is 0 while t < sw_dry_from (Q1; the raw peak takes the frame all the same)
			float2* const list    = sw_list ();
			const float   sw_thr2 = sw_thr ();
			for (int k = 0; k < HS; ++k) {
				const f2      hp   = st[512 * k + t];
				const int64_t pos0 = s + 1024 * (int64_t)k + 2 * (int64_t)t;
				const f2      dp   = half_pair (pos0 - 2 * (int64_t)(t & 255) - Dd, t & 255);
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
			for (int k = 0; k < HS; ++k) {
				PROT_GLOBAL f2* __restrict__ ob = reinterpret_cast<PROT_GLOBAL f2*> (uniform_ptr (out + s + 1024 * k));
				ob[(unsigned)t]                 = st[512 * k + t];
			}
		} else {
#pragma unroll 2
			for (int k = 0; k < HS; ++k) {
				f2            o    = st[512 * k + t];
				const int64_t pos0 = s + 1024 * (int64_t)k + 2 * (int64_t)t;
				if (MODE == MODE_RAMP && pos0 < r1) {
					// per-frame rotation: the dry pair of an output pair is the input pair Dd frames earlier
					const f2 d   = half_pair (pos0 - 2 * (int64_t)(t & 255) - Dd, t & 255);
					const f2 cs0 = reinterpret_cast<const f2*> (p.rot_s)[(int64_t)c * p.n_out + pos0];
					const f2 cs1 = pos0 + 1 < r1 ? reinterpret_cast<const f2*> (p.rot_s)[(int64_t)c * p.n_out + pos0 + 1] : mkf2 (0.f, 0.f);
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
	if (t0 < 256) {
		hist_write_slice (p, blockIdx.y, t0);
	}
#ifdef PROT_SV8_TIMELINE
	// (plain stores into per-workgroup slots, summed by the host: atomics on a few shared words -- 57 K of them per launch -- took
	// longer than the kernel's own tail and were mistaken for it at first)
	if (blockIdx.y == 0 && blockIdx.x < 2048) {
		const int l = t0 & 63, w8 = t0 >> 6;
		if (l < 16) {
			g_sv8_acc[(blockIdx.x * 8 + w8) * 16 + l] = l == 15 ? 1u : l == 0 && sv8_acc == 0u ? sv8_blocks : sv8_acc; // (lane 0: the stamped loop counts its blocks)
		}
		if (t0 < 64) {
			g_sv8_prog[64 * blockIdx.x + t0] = sv8_acc;
		}
		if (t0 == 0) {
			unsigned hw, xcc;
			asm volatile ("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
			unsigned long long* w = g_sv8_wg + 4 * blockIdx.x;
			w[2] = __builtin_amdgcn_s_memrealtime (), w[3] = ((unsigned long long)xcc << 32) | hw;
		}
	}
#endif
	if constexpr (MODE == MODE_SWEEP) {
		if ((t0 & 63) == 0) {
			p.surv_cnt[((int64_t)c * gridDim.x + blockIdx.x) * 8 + (t0 >> 6)] = sw_cnt;
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

// ---------------------------------------------------------------------------------------------------- dynamic blocks (round 5)
// The static run plan leaves the CUs half empty for the last fifth of a launch: while both workgroups of a CU are resident the
// one dispatched first takes 7.5 us per block and the other 17.4 (the hardware serves the oldest wave first), the first ends its
// run at 198 us and the second works on alone until 252 (tools/sv8_progress.py, profiles/r05_sv8_progress_static.json) -- 40
// blocks/us over the launch against 48.5 in its steady state, whatever the run lengths.  Here a workgroup CLAIMS its next block
// when it needs it: tools/gen/gen_olasave8k.py, Gen8kDyn (the claim protocol, the mailbox, why a scalar atomic).
#define PROT_SV8_DYN_CAT_(a, b) PROT_SV8_DYN_##a##_##b
#ifdef PROT_SV8_STAMPS
#define PROT_SV8_DYN_OUT_ "+v"(sv8_acc)
#else
#define PROT_SV8_DYN_OUT_
#endif
#define PROT_SV8_DYN_CALL(HS_, AM_)                                                                                                     \
	if (HS == HS_ && AM == AM_) {                                                                                                       \
		asm volatile (PROT_SV8_DYN_CAT_ (HS_, AM_)                                                                                      \
		              : PROT_SV8_DYN_OUT_                                                                                               \
		              : "s"(a_in), "s"(a_out), "s"(first), "s"(lds_off), "s"(a_sv), "s"(ca), "s"(sa), "v"(t0), "s"(a_ctr), "s"(xr), "s"(tries),  \
		                "s"(per), "s"(nb)                                                                                               \
		              : PROT_SV8_CLOB_ (HS_));                                                                                          \
	}

template <int HS, int MODE>
__global__ __launch_bounds__ (512, 4) void olasave8k_dyn_kernel (const OlaParams p)
{
	constexpr int BLK = 1024 * HS;
	constexpr int NT  = 16 - HS;
	constexpr int R0  = NT / 2;
	constexpr int ROT = (NT & 1) ? 256 : 0;
	constexpr int AM  = MODE == MODE_ROT ? 0 : 2;
	static_assert (MODE == MODE_ROT || MODE == MODE_HILBERT, "steady rotation or the Hilbert branch alone");

	extern __shared__ f2 smem8[];
	__shared__ int       sh_claim[4];
	const int            t0 = threadIdx.x;
	const int            c  = blockIdx.y;
#ifdef PROT_SV8_TIMELINE
	unsigned sv8_acc = 0u;
	(void)sv8_acc;
	if (t0 == 0 && blockIdx.y == 0 && blockIdx.x < 2048) {
		g_sv8_wg[4 * blockIdx.x]     = __builtin_amdgcn_s_memrealtime ();
		g_sv8_wg[4 * blockIdx.x + 1] = 0ull;
	}
#endif
	const float* __restrict__ in   = p.in + (int64_t)c * p.in_stride;
	const float* __restrict__ hist = p.hist ? p.hist + (int64_t)c * p.hist_len : nullptr;
	float* __restrict__ out        = p.out + (int64_t)c * p.out_stride;
	const int64_t n_in = p.n_in;
	const int     Dd   = p.dry_delay;
	const bool in_al8  = (reinterpret_cast<uintptr_t> (in) & 7) == 0;
	const bool out_al8 = (reinterpret_cast<uintptr_t> (out) & 7) == 0;
	float ca = 0.f, sa = p.sv_gain;
	if (MODE == MODE_ROT) {
		ca = p.rot ? p.rot[2 * c] : p.rot_v[2 * c];
		sa = (p.rot ? p.rot[2 * c + 1] : p.rot_v[2 * c + 1]) * p.sv_gain;
	}
	ca = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (ca)));
	sa = __uint_as_float (__builtin_amdgcn_readfirstlane (__float_as_uint (sa)));
	const unsigned lds_off = (unsigned)(uintptr_t)(__attribute__ ((address_space (3))) f2*)smem8;
	const float*   a_sv    = uniform_ptr_generic (p.sv);
	const bool hist_al8 = hist != nullptr && (reinterpret_cast<uintptr_t> (hist) & 7) == 0 && (p.hist_len & 1) == 0;
	auto half_pair = [&] (int64_t ar, int t) -> f2 {
		if (in_al8 && ar >= 0 && ar + 512 <= n_in) {
			return reinterpret_cast<const f2*> (in + ar)[t];
		}
		if (hist_al8 && ar + 512 <= 0 && ar + p.hist_len >= 0) {
			return reinterpret_cast<const f2*> (hist + (ar + p.hist_len))[t];
		}
		const int64_t a = ar + 2 * (int64_t)t;
		return mkf2 (ldv (in, n_in, hist, p.hist_len, a), ldv (in, n_in, hist, p.hist_len, a + 1));
	};

	// ---- the blocks in front of and behind the claimed range: staged through LDS exactly as olasave8k_kernel stages them (the
	// same generated CORE on the same values), one per workgroup
	// (the 64-bit division is done in vector registers: its result is made wave-uniform again, or it lives -- and spills -- per lane)
	const uint64_t n_blocks_v = (uint64_t)(p.n_out + BLK - 1) / (uint64_t)BLK;
	const int64_t  n_blocks   = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane ((int)(n_blocks_v >> 32)) << 32) |
	                                      (uint32_t)__builtin_amdgcn_readfirstlane ((int)(uint32_t)n_blocks_v));
	const int n_edge = (int)(n_blocks - p.dyn_nb); // (a handful: the host takes this route only for streams that are almost all whole blocks)
	for (int eb = (int)blockIdx.x; eb < n_edge; eb += (int)gridDim.x) {
		const int64_t b  = eb < p.dyn_lo ? (int64_t)eb : (int64_t)eb + p.dyn_nb;
		const int64_t s  = b * BLK;
		const int64_t r1 = p.n_out;
		const int64_t a0 = s - Dd - 1024 * R0 - 2 * ROT;
		int t = t0;
		asm volatile ("" : "+v"(t));
		t &= 511;
		LdsC8* const st = (LdsC8*)smem8;
		LdsF8* const gc = (LdsF8*)(smem8 + 8192);
		__syncthreads ();
		{
			const int h = t >> 8, tl = t & 255;
#pragma unroll 2
			for (int k = 0; k < 16; ++k) {
				const int nloc = (512 * k + 256 * h + ROT) & 8191;
				st[512 * k + t] = half_pair (a0 + 2 * (int64_t)nloc, tl);
			}
			const int l = t & 63, w = t >> 6;
			const int nl = (512 * (l & 15) + 64 * w + ROT) & 8191;
			gc[t]        = l < 16 ? ldv (in, n_in, hist, p.hist_len, a0 + 2 * (int64_t)nl - 1) : 0.f;
		}
		__syncthreads ();
		{
			const float* a_in  = a_sv; // (unused by the core)
			float*       a_out = nullptr;
			const int    one   = 1;
			PROT_SV8_ALL (CORE, a_in, a_out, one, ca, sa)
		}
		__syncthreads ();
		t = t0;
		asm volatile ("" : "+v"(t));
		t &= 511;
		if (s + BLK <= r1 && out_al8) {
#pragma unroll
			for (int k = 0; k < HS; ++k) {
				PROT_GLOBAL f2* __restrict__ ob = reinterpret_cast<PROT_GLOBAL f2*> (uniform_ptr (out + s + 1024 * k));
				ob[(unsigned)t]                 = st[512 * k + t];
			}
		} else {
#pragma unroll 2
			for (int k = 0; k < HS; ++k) {
				const f2      o    = st[512 * k + t];
				const int64_t pos0 = s + 1024 * (int64_t)k + 2 * (int64_t)t;
				if (pos0 < r1) {
					out[pos0] = o.x;
				}
				if (pos0 + 1 < r1) {
					out[pos0 + 1] = o.y;
				}
			}
		}
		__syncthreads ();
	}

	// ---- the workgroup's first claim (thread 0; the generated loop makes the later ones, wave 0 with a scalar atomic): its XCD's
	// region first, the next ones when that is empty
	const int nb  = (int)p.dyn_nb;
	const int per = (nb + 7) / 8;
	unsigned* const ctr = p.dyn_ctr + (int64_t)c * kDynWords;
	if (nb > 0) {
		int tid_c = t0;
		asm volatile ("" : "+v"(tid_c));
		if (tid_c == 0) {
			unsigned xcc;
			asm volatile ("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
			int r = (int)(xcc & 7u), left = 8, blk = -1;
			for (;;) {
				const int      start = r * per;
				const int      len   = start >= nb ? 0 : (nb - start < per ? nb - start : per);
				const unsigned tk    = atomicAdd (ctr + 32 * r, 1u);
				if (tk < (unsigned)len) {
					blk = start + (int)tk;
					break;
				}
				if (--left == 0) {
					break;
				}
				r = (r + 1) & 7;
			}
			sh_claim[0] = blk;
			sh_claim[1] = r;
			sh_claim[2] = left;
		}
		__syncthreads ();
		const int first = __builtin_amdgcn_readfirstlane (sh_claim[0]);
		const int xr    = __builtin_amdgcn_readfirstlane (sh_claim[1]);
		const int tries = __builtin_amdgcn_readfirstlane (sh_claim[2]);
		if (first >= 0) {
			// block 0 of the numbering = the channel's first whole in-range block
			const int64_t   s0    = p.dyn_lo * BLK;
			const float*    a_in  = uniform_ptr_generic (in + (s0 - Dd - 1024 * R0 - 2 * ROT));
			float*          a_out = uniform_ptr_generic (out + s0);
			unsigned* const a_ctr = uniform_ptr_generic (ctr + 32 * xr);
#ifdef PROT_SV8_TIMELINE
			if (t0 == 0 && blockIdx.y == 0 && blockIdx.x < 2048) {
				g_sv8_wg[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime ();
			}
#endif
			PROT_SV8_DYN_CALL (8, 0)
			PROT_SV8_DYN_CALL (8, 2)
			PROT_SV8_DYN_CALL (12, 0)
			PROT_SV8_DYN_CALL (12, 2)
			PROT_SV8_DYN_CALL (13, 0)
			PROT_SV8_DYN_CALL (13, 2)
		}
	}
	if (t0 < 256) {
		hist_write_slice (p, blockIdx.y, t0);
	}
#ifdef PROT_SV8_TIMELINE
	if (blockIdx.y == 0 && blockIdx.x < 2048) {
		if ((t0 & 63) < 16) {
			g_sv8_acc[(blockIdx.x * 8 + (t0 >> 6)) * 16 + (t0 & 63)] = (t0 & 63) == 15 ? 1u : sv8_acc;
		}
		if (t0 < 64) {
			g_sv8_prog[64 * blockIdx.x + t0] = sv8_acc;
		}
		if (t0 == 0) {
			unsigned hw, xcc;
			asm volatile ("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
			unsigned long long* w = g_sv8_wg + 4 * blockIdx.x;
			w[2] = __builtin_amdgcn_s_memrealtime (), w[3] = ((unsigned long long)xcc << 32) | hw;
		}
	}
#endif
	// ---- the last workgroup of the channel to leave puts the counters back to zero for the next launch (every workgroup has made
	// all its draws by now, and their results have come back)
	{
		int tid_x = t0;
		asm volatile ("" : "+v"(tid_x));
		if (tid_x == 0) {
			const unsigned gone = atomicAdd (ctr + 256, 1u);
			if (gone == gridDim.x - 1u) {
#pragma unroll
				for (int r = 0; r < 8; ++r) {
					__hip_atomic_store (ctr + 32 * r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				__hip_atomic_store (ctr + 256, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	}
}

template <int HS>
static hipError_t launch_sv8_dyn_m (const OlaParams& p, dim3 grid, hipStream_t s)
{
	const dim3 block (512);
	typedef void (*kern_t) (OlaParams);
	kern_t fn = p.hilbert_only ? olasave8k_dyn_kernel<HS, MODE_HILBERT> : olasave8k_dyn_kernel<HS, MODE_ROT>;
	static bool asked[2][8] = {};
	int         dev = 0;
	(void)hipGetDevice (&dev);
	const int mi = p.hilbert_only ? 1 : 0;
	if (dev < 0 || dev >= 8 || !asked[mi][dev]) {
		hipError_t err = hipFuncSetAttribute ((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, kSv8LdsBytes);
		if (err != hipSuccess) {
			return err;
		}
		if (dev >= 0 && dev < 8) {
			asked[mi][dev] = true;
		}
	}
	hipLaunchKernelGGL (fn, grid, block, kSv8LdsBytes, s, p);
	return hipGetLastError ();
}

hipError_t launch_ola_save8k_dyn (const OlaParams& p, int hop_rows, int n_chn, int grid_x, hipStream_t s)
{
	if ((p.dry_delay & 1) || p.accumulate || p.sv == nullptr || p.hs_flag != nullptr || p.rot_s != nullptr || p.surv != nullptr || p.dyn_ctr == nullptr ||
	    p.dyn_nb < 1 || p.dyn_lo < 0 || grid_x < 1 || 2 * (p.dry_delay - p.conv_delay) != 1024 * (16 - hop_rows) || (p.sv_flags & 1)) {
		return hipErrorInvalidValue;
	}
	dim3 grid ((unsigned)grid_x, (unsigned)n_chn);
	switch (hop_rows) {
		case 8: return launch_sv8_dyn_m<8> (p, grid, s);
		case 12: return launch_sv8_dyn_m<12> (p, grid, s);
		case 13: return launch_sv8_dyn_m<13> (p, grid, s);
		default: return hipErrorInvalidValue;
	}
}

template <int HS>
static hipError_t launch_sv8_m (const OlaParams& p, dim3 grid, hipStream_t s)
{
	const dim3 block (512);
	typedef void (*kern_t) (OlaParams);
	kern_t fn = p.hilbert_only ? olasave8k_kernel<HS, MODE_HILBERT> : p.rot_s != nullptr ? olasave8k_kernel<HS, MODE_RAMP> : olasave8k_kernel<HS, MODE_ROT>;
	if (p.surv != nullptr) { // the fused sweep: the Hilbert pass with the test in its epilogue (lists instead of output rows)
		if constexpr (HS == 8 || HS == 12) {
			fn = olasave8k_kernel<HS, MODE_SWEEP>;
		} else {
			return hipErrorInvalidValue;
		}
	}
	// (more than 64 KiB of LDS per workgroup has to be asked for, once per kernel and device)
	static bool asked[4][8] = {};
	int         dev = 0;
	(void)hipGetDevice (&dev);
	const int mi = p.surv != nullptr ? 3 : p.hilbert_only ? 2 : p.rot_s != nullptr ? 1 : 0;
	if (dev >= 0 && dev < 8 && !asked[mi][dev]) {
		hipError_t err = hipFuncSetAttribute ((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, kSv8LdsBytes);
		if (err != hipSuccess) {
			return err;
		}
		asked[mi][dev] = true;
	} else if (dev >= 8) {
		hipError_t err = hipFuncSetAttribute ((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, kSv8LdsBytes);
		if (err != hipSuccess) {
			return err;
		}
	}
	hipLaunchKernelGGL (fn, grid, block, kSv8LdsBytes, s, p);
	return hipGetLastError ();
}

hipError_t launch_ola_save8k (const OlaParams& p, int hop_rows, int n_chn, hipStream_t s)
{
	if ((p.dry_delay & 1) || (p.run_frames % (1024 * hop_rows)) != 0 || p.accumulate || p.sv == nullptr || p.hs_flag != nullptr ||
	    2 * (p.dry_delay - p.conv_delay) != 1024 * (16 - hop_rows)) {
		return hipErrorInvalidValue;
	}
	if (p.n_out <= 0) {
		return hipSuccess;
	}
	const int64_t unit = p.run_stride > 0 ? p.run_stride : p.run_frames;
	const int64_t runs = (p.n_out + unit - 1) / unit;
	dim3          grid ((unsigned)runs, (unsigned)n_chn);
	switch (hop_rows) {
		case 8: return launch_sv8_m<8> (p, grid, s);
		case 12: return launch_sv8_m<12> (p, grid, s);
		case 13: return launch_sv8_m<13> (p, grid, s);
		default: return hipErrorInvalidValue;
	}
}

} // namespace prot
