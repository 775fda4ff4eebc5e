// prot_ablate.h -- the diagnostic switches of the OLA kernels in prot_kernels.hip, in one place.  None of them is
// defined in the shipped library (phaserotate.lv2_amd/build.py passes no -D): PROT_ABL2 is 0 and every guarded
// statement folds away, Stamps is an empty struct.  Variant libraries are built by tools/ab_build.py
// (`python tools/ab_build.py <tag>:-DPROT_ABL2=5`) into lib/variants/ and loaded with PROT_LIB_PATH; their results are
// WRONG by design (ablations) or slower (stamps), and no test or bench line uses them.
//
// The overlap-save kernel (prot_olasave.hip) has its own counterparts: generator options of tools/gen/gen_olasave.py
// (nolds, nomem, novalu: ablations emitted as separate include files under csrc/variants/) and PROT_SV_CLOCK.
//
// Switches that were tried and withdrawn are no longer in the source; the record of what they measured is DESIGN.md 5:
//   prefetch placement (PROT_PREFETCH_AT / _EARLY / _ELSE: the last one miscompiled under hipcc 7.2), non-temporal
//   loads / stores, LDS stores pinned per butterfly stage (PROT_PIPE_STORES), the spectrum table shrunk to probe four
//   workgroups per CU (PROT_PROBE4), tables fetched before the first block (PROT_TABLES_LATE=0), the per-XCD run
//   rotation switch (now unconditional), non-volatile exchange accesses (PROT_LDS_VOLATILE=0).
#pragma once
#include <stdint.h>

// PROT_ABL2 (timing experiments only, results are wrong): 1 no LDS exchange, 2 no workgroup barriers, 4 no radix-16
// butterflies, 8 no twiddle multiplies, 16 no spectrum multiply, 32 no stream loads, 64 no stores, 128 the stream does
// not advance (each run re-reads and re-writes its first block: the same instructions, served by the caches)
#ifndef PROT_ABL2
#define PROT_ABL2 0
#endif

namespace prot {

// PROT_STAMPS (diagnostic build only, tools/attic/phase_stamps.py): per-phase shader-cycle sums of the polyphase kernel's
// main loop, accumulated per wave in SGPRs and added to g_stamps at the end.  Each stamp is an s_memtime, which
// drains lgkmcnt: the stamped build runs ~10 % slower and nothing of it exists in the shipped library.
#ifdef PROT_STAMPS
__device__ unsigned long long g_stamps[32];
__device__ unsigned long long g_wg_t[4 * 4096]; // per workgroup: entry, loop start, end (s_memrealtime, 100 MHz), HW_ID
struct Stamps {
	uint64_t last, acc[16];
	__device__ __forceinline__ void start ()
	{
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			acc[i] = 0;
		}
		last = __builtin_readcyclecounter ();
	}
	__device__ __forceinline__ void tick (int i)
	{
#if PROT_STAMPS == 1 // (2: the per-workgroup timeline only -- the per-phase stamps slow the kernel by a third)
		const uint64_t now = __builtin_readcyclecounter ();
		acc[i] += now - last;
		last = now;
#endif
	}
};
#else
struct Stamps {
	__device__ __forceinline__ void start () {}
	__device__ __forceinline__ void tick (int) {}
};
#endif

} // namespace prot
