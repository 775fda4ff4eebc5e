// prot_engine.h -- what the host-side units of libprot_hip.so share (round 6: prot_capi.hip, 4 200 lines, split into prot_engine.hip
// [errors, FIR design, the per-engine tables, create / destroy / reset / sync, streams], prot_apply.hip [run geometry, the OLA launch,
// prot_apply_dev*], prot_lv2.hip [the resident run() server, prot_lv2_run, the deferred mode], prot_sweep_host.hip [prot_sweep_dev,
// seeding, the CLI's block and file calls], prot_multi.hip [RCCL, several engines], prot_hooks.hip [test hooks]; same code).
// Internal: nothing here crosses the C ABI (include/prot_hip.h).
#pragma once
#include "../../include/prot_hip.h"
#include "prot_internal.h"
#include "prot_kernels.h"

#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <dlfcn.h>
#include <thread>
#include <climits>
#include <map>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace prot_host {

// The shipped library (the one LV2 hosts load) never reads its environment.  Tuning switches -- run geometry, forced
// kernel forms -- exist only in PROT_TUNING builds (tools/ab_build.py); what the test-suite and bench.py need to switch
// (the unpruned sweep, every block through the staged path, the all-reduce route) goes through prot_internal_set_test_option.
#ifdef PROT_TUNING
inline const char* tuning_env (const char* name) { return getenv (name); }
#else
inline const char* tuning_env (const char*) { return nullptr; }
#endif
// ... and that entry point exists only in the PROT_TESTHOOKS build of the library (libprot_hip_hooks.so: the same sources, what
// tests/ and bench.py load); the shipped libprot_hip.so has no test hook, no switch to flip, and exports nothing but the symbols
// include/prot_hip.h declares (version script, phaserotate.lv2_amd/build.py).
#ifdef PROT_TESTHOOKS
extern std::atomic<int> g_test_opt_v[PROT_TEST_OPTIONS];
inline int              test_opt (int o) { return g_test_opt_v[o].load (); }
#else
inline int test_opt (int) { return 0; }
#endif

extern thread_local char g_err[256];
int hip_fail (hipError_t e, const char* what);

#define HIP_TRY(expr)                                  \
	do {                                               \
		hipError_t _e = (expr);                        \
		if (_e != hipSuccess) {                        \
			return hip_fail (_e, #expr);               \
		}                                              \
	} while (0)

// ---- SinCosLut (cli/phase-rotate.cc:41-74)
extern float          g_lut[PROT_MAXSAMPLE * 2]; // {sin, cos}
extern std::once_flag g_lut_once;
void lut_init ();

template <typename T>
struct DevBuf {
	T*     p   = nullptr;
	size_t cap = 0; // elements
	int    ensure (size_t n)
	{
		if (n <= cap) {
			return PROT_OK;
		}
		if (p) {
			(void)hipFree (p);
			p   = nullptr;
			cap = 0;
		}
		hipError_t e = hipMalloc ((void**)&p, n * sizeof (T));
		if (e != hipSuccess) {
			return hip_fail (e, "hipMalloc");
		}
		cap = n;
		return PROT_OK;
	}
	void release ()
	{
		if (p) {
			(void)hipFree (p);
		}
		p   = nullptr;
		cap = 0;
	}
};

template <typename T>
struct PinBuf {
	T*     p   = nullptr;
	size_t cap = 0;
	bool   coherent = false;
	int    ensure (size_t n)
	{
		if (n <= cap) {
			return PROT_OK;
		}
		if (p) {
			(void)hipHostFree (p);
			p   = nullptr;
			cap = 0;
		}
		// (coherent: what a RESIDENT kernel reads and writes while the host does -- the run() server's mailbox and port block)
		hipError_t e = hipHostMalloc ((void**)&p, n * sizeof (T), coherent ? (hipHostMallocCoherent | hipHostMallocMapped) : hipHostMallocDefault);
		if (e != hipSuccess) {
			return hip_fail (e, "hipHostMalloc");
		}
		cap = n;
		return PROT_OK;
	}
	void release ()
	{
		if (p) {
			(void)hipHostFree (p);
		}
		p   = nullptr;
		cap = 0;
	}
};

struct Lv2Chn {            // Channel angle state, src/phaserotate.c:52-55, :61
	float               angle = 0.f, sa = 0.f, ca = 1.f;
	uint32_t            offset = 0;
	std::vector<float>  rotq;  // [parsiz][2] {ca, sa} of the block being delivered
	bool                rotq_steady = true; // every entry of rotq is {ca, sa} (no ramp in it)
};

} // namespace prot_host
using namespace prot_host;

struct prot_engine {
	prot_info   info {};
	int         device  = 0;
	int         hopseg  = 0;
	hipStream_t stream  = nullptr;
	std::vector<float> fir;
	uint32_t    part_len = 0; // taps per FIR partition
	bool        use_p2   = false; // two partitions in one launch (frequency-domain delay line)
	bool        use_poly = false; // polyphase form: partitions of g[s] = fir[2 s + 1], blocks of 4096 frames
	DevBuf<float2> d_Hp;          // 4096-tap engines (direct form): spectrum of the 2048 odd taps -- short LV2 launches, fp16 path

	DevBuf<float2> d_H, d_tw;
	// overlap-save, table-free form of the one-partition polyphase engines (prot_olasave.hip): four workgroups per CU
	bool           use_save = false;
	// long launches of overlap-save LV2 engines: 8192-point blocks (prot_olasave8k.hip)
	bool           use_save8 = false;
	int            sv8_rows  = 0;     // new rows per 8192-point block (12 or 13)
	float          sv8_gain  = 0.f;
	DevBuf<float>  d_sv8;
	// CLI blksiz 16384 (round 5): every fp32 launch on 16384-point blocks (olasave16k_kernel: one table-free partition, one launch)
	bool           use_save16 = false;
	int            sv16_rows  = 0;     // new rows per 16384-point block: 8 (blksiz 16384) or 12 (blksiz 8192)
	float          sv16_gain  = 0.f;
	DevBuf<float>  d_sv16;
	// dynamic blocks (olasave8k_dyn_kernel): kDynSets counter records per channel, one set per launch in turn (a launch leaves its
	// set zeroed; the turn keeps launches that overlap on different streams of the caller's apart)
	DevBuf<uint32_t> d_dyn;
	unsigned         dyn_seq = 0;
	bool             dyn_off = false; // PROT_CFG_STATIC_RUNS
	DevBuf<float>  d_sv;
	float          sv_gain = 0.f;
	int            sv_rows = 0;
	int            sv_flags = 0; // 1 (PROT_SV_NOASM at creation, tests): every block through the C++ form of olasave_kernel
	static constexpr int kDynSets = 8;
	static constexpr int kPipe = 3, kHist = kPipe + 1; // pipelined launches in flight; history ring (one more)
	DevBuf<float>  d_lut, d_rot, d_hist[kHist], d_peaks, d_in, d_out, d_u, d_inter, d_rot_s;
	// whole-file calls: two sets of staging buffers, a copy stream and events, so that chunk k+1 is uploaded and
	// de-interleaved while chunk k is swept (cli/phase-rotate.cc:565-587 reads and analyses block by block)
	DevBuf<float>  d_inter_b, d_in_b;
	// prot_cli_file_resident: the de-interleaved file of the last prot_cli_analyze_file, [n_chn][file_n] planar
	bool           file_resident = false;
	DevBuf<float>  d_file;
	const float*   file_key = nullptr;
	uint64_t       file_n   = 0, file_stride = 0; // (rows 256-byte aligned: the same aligned paths as the staged chunks)
	hipStream_t    copy_stream = nullptr;
	hipEvent_t     ev_copied[2] = {}, ev_free[2] = {};
	std::vector<int> angles_last;          // what d_angles holds, uploaded on angles_stream
	hipStream_t      angles_stream = nullptr;
	DevBuf<uint16_t> d_hist16[2]; // fp16 sample path: its own input history (half precision), double-buffered
	int            hist16_cur = 0;
	DevBuf<int>    d_angles;
	DevBuf<float>  d_thr;   // [n_chn][4] shared pruning bounds of one sweep launch
	DevBuf<unsigned> d_ctr; // [n_chn][32][32] step counters of a sweep launch (zero between launches)
	DevBuf<float>  d_part;  // per-workgroup maxima of one sweep launch (prot_sweep.hip), allocated by the first sweep
	DevBuf<unsigned long long> d_stats; // {tiles evaluated, tiles seen, tiles untested, surviving frames} since the last reset
	// fused sweep (olasave_kernel MODE_SWEEP + sweep_eval_kernel): the survivor lists of one pass, their lengths, the step table
	DevBuf<float2>   d_surv;
	DevBuf<unsigned> d_surv_cnt, d_nsteps;
	DevBuf<uint2>    d_desc;
	uint64_t         fused_frames = 0;          // channel-frames the fused pass has tested since the last reset
	DevBuf<unsigned long long> d_fused_surv;    // ... and how many of them it kept (device counter)
	bool             sweep_seeded = false;      // the table holds a bound for the angle list of angles_last
	bool             sweep_seeded_ext = false;  // ... put there by prot_sweep_seed_dev (and whatever the caller merged into it): the next prot_sweep_dev keeps it even at pos 0
	bool             seed_chunk_ready = false;  // analyze_range: the range's first chunk is on the device already (its seeding phase uploaded it)
	int              last_n_lists = 0;          // geometry of the last fused pass (prot_internal_fused_lists)
	int64_t          last_cap     = 0;
	PinBuf<float>  h_io;
	// the run() server (prot_olasave512.hip): a resident kernel that takes prot_lv2_run's calls through a mailbox
	char*          srv_box = nullptr;      // host -> device half (prot::SvServerBox + input frames + per-frame rotation): device memory
	                                       // the CPU stores into through the BAR (srv_bar), or pinned host-coherent memory
	size_t         srv_box_bytes = 0;
	bool           srv_bar = false;
	PinBuf<prot::SvServerAck> srv_ack;     // device -> host half: pinned host-coherent memory
	DevBuf<unsigned> d_srv_relay;
	hipStream_t    srv_stream  = nullptr;
	hipEvent_t     srv_ev      = nullptr;
	bool           sv8_always   = false; // CLI blksiz 8192: every fp32 launch on 8192-point blocks (one table-free partition)
	bool           srv_eligible = false; // LV2 engine on the overlap-save kernel, <= 32 channels, not pipelined
	bool           srv_running = false;
	uint32_t       srv_seq     = 0;      // number of the last call posted
	uint64_t       srv_calls   = 0;      // calls the server took / kernels started for it (prot_internal_server_stats)
	uint64_t       srv_starts  = 0;
	// DEFERRED run() (prot_lv2_set_deferred; round 6): a call posts its block to the server and returns WITHOUT waiting for it; what
	// it hands back is the output `rt_frames` frames older, out of a host ring that the next call tops up with the block posted by
	// this one.  rt_frames = 0: off (every call waits for its own block).
	uint32_t           rt_frames   = 0;     // the extra latency, = the longest call the mode takes in one piece
	std::vector<float> rt_ring;             // [n_chn][rt_cap]
	uint32_t           rt_cap = 0, rt_rd = 0, rt_fill = 0;
	bool               rt_pending  = false; // a block is with the server (or, rt_pend_ready, was computed by a blocking route)
	bool               rt_pend_ready = false;
	bool               rt_discard  = false; // ... and silence has already gone into the ring in its place: it arrived late
	uint32_t           rt_pend_seq = 0, rt_pend_n = 0, rt_skip_n = 0;
	const float*       rt_pend_out = nullptr;
	uint64_t           rt_late = 0, rt_dropped = 0;
	// small host->device parameter copies (rotation pairs, angle lists) are staged through a ring of
	// pinned slots owned by the engine, each guarded by an event: the caller's array may die as soon
	// as the call returns, and a slot is never rewritten before its previous copy has been consumed
	static constexpr int kSlots = 64, kSlotBytes = 4096;
	PinBuf<char>   h_slots;
	hipEvent_t     slot_ev[kSlots] = {};
	int            slot_next = 0;
	int            hist_cur = 0; // ring: a call reads [cur] and writes [(cur + 1) % kHist]
	// PROT_CFG_PIPELINE: stateful prot_apply_dev calls on the engine's own streams rotate over kPipe of them and hand
	// the history over on the device (prot_olasave.hip); everything else joins them first (join_pipe)
	bool           pipeline   = false;
	hipStream_t    pipe_stream[kPipe] = {};
	hipEvent_t     pipe_ev[kPipe + 1] = {};
	// The stream of the last call, when it was the caller's.  It is used once more, to record ev_last on it, when something has to come
	// BEHIND that call on another stream -- prot_engine_reset, a call on another stream, prot_engine_sync: mark_last -- and forgotten
	// there; until then the caller keeps it alive (the rule is in prot_hip.h).  Recording the event at the end of every
	// call instead, so that no handle need be kept, was measured: +3 us per launch on back-to-back 11 us launches (BASELINE config
	// 3; tools/ubench/event_cost.hip: 2.9-5.3 us per hipEventRecord behind a kernel, whatever the event's flags).
	hipEvent_t     ev_last        = nullptr;
	bool           last_foreign   = false;
	// Work queued on the ENGINE's stream (stream = NULL calls) since the host last waited for it: a call that names a stream of the
	// caller's is put behind it by an event (pick_stream) -- prime_dev (NULL) followed by sweep_dev (S) reads the history the first
	// wrote.  The common same-stream sequence never records anything.
	hipEvent_t     ev_own         = nullptr;
	bool           own_dirty      = false;
	const void*    last_stream_id = nullptr;
	hipEvent_t     ev_reset    = nullptr;
	hipStream_t    reset_stream  = nullptr; // where the last prot_engine_reset put its zeroing, behind ev_reset ...
	bool           reset_pending = false;   // ... until a call has taken a stream (pick_stream orders a DIFFERENT stream behind it)
	DevBuf<uint32_t> d_pipe;      // [0..7] generation g's history is complete when slot g & 7 holds g; [8..11] slice counters;
	                              // [12] status: the generation a kernel gave up waiting for (0: none) -- checked by prot_engine_sync
	bool           pipe_used  = false; // pipelined launches since the status word was last looked at
	uint32_t       pipe_gen   = 0;    // pipelined launches so far (= the value the last one publishes)
	uint32_t       pipe_q     = 0;    // ... since the last join (stream and counter of the next one)
	bool           pipe_dirty = false; // work may be in flight on the pipe streams
	uint64_t       pos      = 0; // frames consumed since reset (CLI stream position)
	int            pairs_per_run_override = 0;
	int            wg_slots = 768; // resident OLA workgroups on the device (CUs x 3)
	int            cus      = 256; // compute units of the device (read once, at create)
	uint32_t       lv2_max_block = 0; // frames per prot_lv2_run chunk the buffers are reserved for
	bool           counted = false;   // in g_engines (communicator cache lifetime)

	// LV2 block-rate angle state (src/phaserotate.c:563-571, :673-717)
	float   interp_th = 0.f, interp_nm = 0.f;
	std::vector<Lv2Chn> lv2;

	std::vector<float> h_peaks; // host mirror for prot_cli_peak
	bool               h_peaks_valid = false;
};

namespace prot_host {

// core of prot_apply_dev / the Hilbert pass of the sweep
// what the fused Hilbert pass of a sweep adds to an overlap-save launch (OlaParams::surv ...)
struct FusedSweep {
	float*     peaks;      // table [n_chn][360] (bound, raw peak)
	const int* angles;
	int        n_angles, raw_slot, raw_tail;
	bool       sample;     // the seeding pass: every `stride_blocks`-th block, no bound, nothing of the engine's state moves
	int        stride_blocks;
	int64_t    t_begin, dry_from;
	bool       plan_only;  // compute the geometry (n_lists, cap) and return without launching
	// out: the geometry of the lists
	int        n_lists;
	int64_t    cap;
};

// ---- prot_engine.hip
int      zero_state (prot_engine* e, hipStream_t zs = nullptr);
uint64_t file_align_frames (const prot_engine* e);
uint64_t file_chunk_frames (const prot_engine* e);
bool       mark_last (prot_engine* e);
hipStream_t pick_stream (prot_engine* e, void* stream, bool keep_server = false);
hipError_t sync_own (prot_engine* e);
int        stage_small (prot_engine* e, void* dst_dev, const void* src, size_t bytes, hipStream_t s);
// ---- prot_apply.hip
int  pick_run_frames (const prot_engine* e, uint64_t n_out, int n_chn, int unit_frames = 0, int slots_override = 0);
int  ola_launch (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_in, float* d_out, int64_t out_stride, uint64_t n_out,
                 const float* d_rot, const float* d_rot_s, bool hilbert_only, bool stateless, int chn0, int n_chn, hipStream_t s,
                 bool advance_hist = false, const float* h_rot = nullptr, const uint32_t* pipe_wait_post = nullptr, FusedSweep* fs = nullptr,
                 prot::SvServerBox* srv = nullptr);
int  hist_advance (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_in, int chn0, int n_chn, hipStream_t s);
bool io_overlap_bytes (const void* in, int64_t in_stride, uint64_t n_in, const void* out, int64_t out_stride, uint64_t n_out, int nc, size_t esz);
bool io_overlap (const float* in, int64_t in_stride, uint64_t n_in, const float* out, int64_t out_stride, uint64_t n_out, int nc);
int  join_pipe (prot_engine* e, hipStream_t s);
// ---- prot_lv2.hip: the run() server (one resident kernel per process) and the deferred run()
extern std::mutex                g_srv_mu;
extern std::vector<prot_engine*> g_srv_roster, g_srv_members;
extern hipStream_t               g_srv_stream;
extern int                       g_srv_hold;
int  lv2_reserve (prot_engine* e, uint32_t frames);
void rt_wait_pending (prot_engine* o);
int  server_stop_locked ();
int  server_stop_any ();
bool server_is_member (const prot_engine* e);
bool server_roster_takes (prot_engine* e);
void server_roster_remove (prot_engine* e);
int  server_start_locked (prot_engine* e);
int  server_start_wait_locked (prot_engine* e);
int  rt_prestart (prot_engine* e);
void rt_ring_reset (prot_engine* e); // (deferred run(): the ring back to rt_frames frames of silence)
// Calls that allocate, free or synchronise the whole device HOLD the server off for their whole duration (prot_lv2.hip)
struct ServerHold {
	ServerHold ();
	~ServerHold ();
	ServerHold (const ServerHold&)            = delete;
	ServerHold& operator= (const ServerHold&) = delete;
};
// ---- prot_sweep_host.hip
// phase 0: the whole range (one engine); 1: prime the history, upload the range's first chunk and SEED the bound from it (sharded
// analysis: the tables are merged before the main pass); 2: the main pass behind phase 1
int analyze_range (prot_engine* e, const float* data, uint64_t f0, uint64_t nf, bool is_first, bool is_last, int ang_start,
                   int ang_end, int ang_stride, int chn, int phase = 0, int seed_blocks = 64);
// ---- prot_multi.hip: communicators live as long as engines do
extern std::mutex g_rccl_lock;
extern int        g_engines; // live engines (guarded by g_rccl_lock)
void rccl_drop_comms_locked ();
int  rccl_set_override (const char* path); // (test hook: the library Rccl::load opens)


// the stream of one API call: picked at its start, remembered at its end when it is the caller's (mark_last)
struct CallStream {
	prot_engine* e;
	hipStream_t  s;
	CallStream (prot_engine* e_, void* stream, bool keep_server = false) : e (e_), s (pick_stream (e_, stream, keep_server)) {}
	~CallStream ()
	{
		e->last_foreign   = s != e->stream;
		e->last_stream_id = e->last_foreign ? (const void*)s : nullptr;
		if (!e->last_foreign) {
			e->own_dirty = true;
		}
	}
	CallStream (const CallStream&)            = delete;
	CallStream& operator= (const CallStream&) = delete;
	operator hipStream_t () const { return s; }
};


} // namespace prot_host

