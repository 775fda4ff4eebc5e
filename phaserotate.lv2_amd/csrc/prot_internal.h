// prot_internal.h -- entry points of libprot_hip.so that are NOT part of the drop-in boundary (include/prot_hip.h):
// hooks for the test-suite.  Exported so that a test process can call them; no product code does.
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
// the shared library dlopen()ed in place of the system's RCCL by prot_allreduce_peaks (tests: a counting stand-in);
// only before the first all-reduce of the process, NULL or "" = the default search.  0 or PROT_EINVAL.
int prot_internal_set_rccl_library (const char* path);
// process-wide switches of the test-suite and bench.py (the shipped library reads no environment variables):
enum {
	PROT_TEST_SWEEP_UNPRUNED = 0, // != 0: prot_sweep_dev launches the unpruned sweep kernel (bit-identical tables; timing and tests)
	PROT_TEST_EDGE_PATH_ONLY = 1, // != 0: engines created from now on run every overlap-save block through the staged (edge) path
	PROT_TEST_LV2_STAGED     = 2, // != 0: prot_lv2_run copies its buffers through staging memory whatever their size
	PROT_TEST_ALLREDUCE      = 3, // 0 automatic; 1 merge through the host; 2 the RCCL collective even when engines share a device; 3 ... and for one engine
	PROT_TEST_SWEEP_UNFUSED  = 4, // != 0: prot_sweep_dev keeps the separate Hilbert and sweep passes where it would fuse them (bit-identical tables)
	PROT_TEST_LV2_NO_SERVER  = 5, // != 0: prot_lv2_run launches every call (no resident server kernel)
	PROT_TEST_LV2_NO_BAR     = 6, // != 0: engines created from now on keep the run() server's mailbox in pinned host memory (as on a device without a large BAR)
	PROT_TEST_SEED_MFMA      = 7, // != 0: the seeding pass of a fused sweep is evaluated by the matrix-core kernels of the main pass (rounds 4's route; A/B)
	PROT_TEST_FUSED_SPLIT_LOG2 = 8, // > 0: prot_sweep_dev splits calls of more than 2^value channel-frames in two (default 31: the lists' 32-bit entry offsets)
	PROT_TEST_F16_NO_MFMA    = 9, // != 0: the fp16 sample path keeps its packed-half butterflies on the vector pipe (olapoly_kernel<T16>; A/B and cross-check)
	PROT_TEST_NO_SHORT_CHAIN = 10, // != 0: short launches keep the 256-thread block (olasave_kernel) instead of the 512-thread one (A/B and cross-check)
	PROT_TEST_OPTIONS        = 11
};
int prot_internal_set_test_option (int option, int value); // 0 or PROT_EINVAL
// the lists the last fused sweep pass of the engine left (device pointers, valid until the next sweep): float2 entries (dry, hilbert),
// list l of channel c at entry (c * n_lists + l) * cap, its length in cnt[c * n_lists + l]; out = { surv, cnt, n_lists, cap }
struct prot_engine;
int prot_internal_fused_lists (struct prot_engine* e, unsigned long long out[4]);
// mean microseconds per prot_lv2_run call on `frames` frames per channel, `calls` calls straight from C (what an LV2 host's audio
// thread sees, without a binding in between)
int prot_internal_time_lv2_run (struct prot_engine* e, unsigned frames, unsigned calls, double us[3]); // mean, slowest call, its index
int prot_internal_time_lv2_run_multi (struct prot_engine* const* es, unsigned n, unsigned frames, unsigned calls, double us[3]); // the same, n engines in turn
// the same with `gap_us` microseconds of busy-waiting between two calls (an audio period, scaled down: what a deferred run() needs
// to find its last block served) and each call timed BY ITSELF: us = { mean call, slowest call, its index }
int prot_internal_time_lv2_run_paced (struct prot_engine* e, unsigned frames, unsigned calls, unsigned gap_us, double us[3]);
// the run() server of an LV2 engine: out = { calls it has taken, kernels started for it }
int prot_internal_server_stats (struct prot_engine* e, unsigned long long out[2]);
#ifdef __cplusplus
}
#endif
