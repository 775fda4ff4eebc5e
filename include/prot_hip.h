/*
 * prot_hip.h -- C ABI of libprot_hip.so: the MI355X (gfx950) implementation
 * of phaserotate.lv2's FFT/Hilbert phase-rotation hot path.
 *
 * Plain C: opaque handle, plain pointers and sizes, int status returns.  No
 * HIP, C++ or torch types cross this boundary (a HIP stream is passed as
 * void*; NULL = the engine's own stream).  Every entry point names the
 * reference interface it replaces (file:line into x42/phaserotate.lv2).
 *
 * There is NO CPU fallback behind this ABI: if no gfx950 device/HIP runtime
 * is usable, prot_engine_create fails with PROT_ENODEV and nothing else works.
 */
#ifndef PROT_HIP_H
#define PROT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PROT_ABI_VERSION 7
#define PROT_SUBSAMPLE 2                      /* cli/phase-rotate.cc:38 */
#define PROT_MAXSAMPLE (180 * PROT_SUBSAMPLE) /* cli/phase-rotate.cc:39 */

enum prot_status {
	PROT_OK           = 0,
	PROT_EINVAL       = 1, /* bad argument */
	PROT_ENODEV       = 2, /* no usable gfx950 device / HIP runtime */
	PROT_ENOMEM       = 3, /* host or device allocation failed */
	PROT_EHIP         = 4, /* a HIP call failed (see prot_last_error) */
	PROT_EUNSUPPORTED = 5  /* valid in the reference, not yet built here */
};

enum prot_kind {
	/* PhaseRotateProc(blksiz): FIR = blksiz taps, fftlen = 2*blksiz,
	 * hop = blksiz, dry delay blksiz/2 (cli/phase-rotate.cc:128-165, :214-225) */
	PROT_KIND_CLI = 0,
	/* instantiate(rate): FIR 3072/4096/8192 taps by rate, latency
	 * parsiz + firlen/2 (src/phaserotate.c:278-297, :374-401).  The device
	 * engine re-blocks the same LTI filter onto one 4096/8192-point OLA FFT
	 * per hop instead of n_segm 512..2048-point partitions. */
	PROT_KIND_LV2 = 1
};

typedef struct {
	int32_t  kind;    /* enum prot_kind */
	double   param;   /* CLI: blksiz (power of two 1024..32768); LV2: sample rate */
	uint32_t n_chn;   /* channel-streams (LV2: 1..2 via the plugin; engine itself: 1..65535) */
	int32_t  device;  /* HIP device ordinal */
	uint32_t fftlen;  /* 0 or 4096: the device transform length */
	uint32_t flags;   /* PROT_CFG_* */
} prot_config;

/* FIRs of 2049..3072 taps (LV2 < 64 kHz): keep ONE zero-padded transform per 1024-frame hop
 * (the literal "one 4096-pt FFT per 1024-frame run()" blocking) instead of two 2048-tap
 * partitions sharing a forward transform.  Same filter, ~1.6x the transforms per frame. */
#define PROT_CFG_SINGLE_FFT_PER_HOP 1u
/* FIRs of more than 2049 taps: keep the direct form (2048-tap partitions of the FIR, pairs sharing
 * a forward transform) instead of the polyphase form (the FIR is zero at every even tap, so the
 * stream read as complex samples x[2j-1] + i x[2j] is convolved with the odd taps only: half the
 * partitions, 4096-frame blocks).  Same filter; kept for cross-checks. */
#define PROT_CFG_DIRECT_FORM 2u
/* Polyphase form also for the 4096-tap FIRs (LV2 96 kHz, CLI blksiz 4096), where the default is the
 * direct form (measured 6-7 % faster there: no longer hop to gain); every other FIR length is polyphase
 * by default (1024..3072 taps: longer hops; 8192 and more: half the launches). */
#define PROT_CFG_POLYPHASE 4u
/* One-partition polyphase engines (LV2 48 / 96 kHz, CLI blksiz 1024..4096): keep the overlap-ADD kernel with its
 * spectrum table (olapoly_kernel, three workgroups per CU) instead of the overlap-SAVE, table-free form
 * (olasave_kernel, four per CU).  Same filter; kept for cross-checks. */
#define PROT_CFG_OVERLAP_ADD 8u
/* Pipelined stateful launches (overlap-save engines only; ignored by the others): consecutive prot_apply_dev calls with
 * stream == NULL, steady rotation and history rotate over three streams of the engine and overlap on the device --
 * launch k + 1 needs nothing of launch k but the history, which the kernel hands over on the device.  For callers that
 * issue many short launches back to back (BASELINE config 3: 256 x 1024-frame buffers per launch).  Same results bit for
 * bit.  Outputs are complete after prot_engine_sync (or any later call on the engine, which is ordered behind them);
 * the caller must not reuse an input or output buffer of a launch before that.  (The shipped library reads no
 * environment variable: this flag is the only switch.) */
#define PROT_CFG_PIPELINE 16u
/* Overlap-save LV2 engines run LONG launches (>= 2^24 channel-frames) on 8192-point blocks (olasave8k_kernel: 512-thread
 * workgroups, two per CU, 12 or 13 of 16 rows of a block new) and everything shorter on 4096-point blocks (olasave_kernel).
 * This flag keeps the 4096-point kernel for every launch.  Same filter, results equal to a rounding error (~1e-7);
 * for cross-checks and timing comparisons.  CLI engines always use ONE block size (sharded sweeps are bit-identical
 * whatever the shard sizes because of it). */
#define PROT_CFG_BLOCK4096 32u
/* Engines on 8192-point blocks: keep the STATIC run plan (every resident workgroup a run of consecutive blocks, lengths by
 * dispatch rank) for long launches, instead of letting the workgroups claim their blocks one at a time as they go
 * (olasave8k_dyn_kernel).  The same blocks through the same generated arithmetic: results equal bit for bit; for cross-checks
 * and timing comparisons. */
#define PROT_CFG_STATIC_RUNS 64u

typedef struct {
	int32_t  kind;
	uint32_t n_chn;
	uint32_t fftlen;      /* device OLA FFT length */
	uint32_t hop;         /* new frames per FFT block (polyphase engines: 4096 = 2048 complex samples) */
	uint32_t firlen;      /* Hilbert FIR taps */
	uint32_t latency;     /* LV2: latency port value; CLI: blksiz/2 (cli/phase-rotate.cc:963) */
	uint32_t dry_delay;   /* frames the dry path is delayed by */
	uint32_t conv_delay;  /* extra delay in front of the FIR */
	uint32_t history;     /* input frames of state kept per channel */
	uint32_t blksiz;      /* CLI: parsiz; LV2: reference parsiz (ramp granularity) */
	uint32_t partitions;  /* FIR partitions of <= fftlen/2 taps convolved per hop (polyphase engines: partitions
	                       * of the odd taps) */
	uint32_t run_align;   /* frames: a stream cut at multiples of this (CLI: of lcm(run_align, blksiz)) and fed to
	                       * separate engines / launches reproduces the single-pass arithmetic BIT FOR BIT -- it is
	                       * the span of one device transform block (pair); shard and chunk boundaries use it */
	uint32_t form;        /* which device kernel convolves: PROT_FORM_* (reporting only) */
	uint32_t wg_per_cu;   /* resident workgroups per CU that kernel is sized for (run lengths are planned with it) */
} prot_info;
#define PROT_FORM_DIRECT 0u        /* partitions of the FIR itself, hop pairs as re/im (ola4096_kernel, ola4096p2_kernel) */
#define PROT_FORM_POLY_OLA 1u      /* polyphase, overlap-add, spectrum table (olapoly_kernel) */
#define PROT_FORM_POLY_SAVE 2u     /* polyphase, overlap-save, table-free, generated loop (olasave_kernel) */
#define PROT_FORM_POLY_SAVE8K 3u   /* the same on 8192-point blocks for long launches (olasave8k_kernel), olasave_kernel for short ones */
#define PROT_FORM_POLY_SAVE16K 4u  /* the same on 16384-point blocks (olasave16k_kernel): CLI blksiz 16384, every fp32 launch */

typedef struct prot_engine prot_engine;

/* ---- lifetime ----------------------------------------------------------- */
/* replaces PhaseRotateProc::PhaseRotateProc + PhaseRotate::PhaseRotate
 * (cli/phase-rotate.cc:128-165, :314-335) and instantiate()'s DSP part
 * (src/phaserotate.c:278-401).  Returns NULL on failure (as instantiate does,
 * src/phaserotate.c:315-372); *status (optional) receives the reason. */
prot_engine* prot_engine_create (const prot_config* cfg, int* status);
/* ~PhaseRotateProc / ~PhaseRotate / cleanup() (src/phaserotate.c:179-223) */
void prot_engine_destroy (prot_engine* e);
/* PhaseRotate::reset (cli/phase-rotate.cc:355-366) / activate -> channel_reset
 * (src/phaserotate.c:169-177, :511-520): clears input history, OLA state,
 * peak table and block position.  The LV2 angle state is NOT reset (Q10).
 * After a call on a stream of the caller's it does not block the host: the zeroing is
 * queued on the engine's stream behind the call's kernels, the next call, on whatever stream,
 * is ordered behind it, and every entry point that reads device state (peaks,
 * statistics, prot_engine_sync) waits for it.  After a call on the engine's own stream
 * (stream = NULL) it returns when that stream has drained, as it always did. */
int prot_engine_reset (prot_engine* e);
/* Streams.  Every device entry point takes `void* stream` (a hipStream_t of the caller's; NULL = the engine's own
 * stream).  Calls on one engine are ordered on the device in the order they were made, whatever streams they name.
 * The engine keeps the handle of the LAST call's stream -- nothing else of it -- until the next entry point that has to
 * come behind that call has returned: prot_engine_reset, a call naming another stream, prot_engine_sync,
 * prot_engine_destroy.  A caller that destroys a stream the engine ran on calls one of those first
 * (prot_engine_sync also waits for the call).  (An event behind every call would need no such rule: measured, 3 us
 * per call on back-to-back 11 us launches.) */
int prot_engine_info (const prot_engine* e, prot_info* out);
/* the time-domain FIR the engine convolves with (firlen floats); reference
 * values: src/phaserotate.c:374-391, cli/phase-rotate.cc:144-161 */
int prot_engine_fir (const prot_engine* e, float* fir_out);
const char* prot_strerror (int status);
/* thread-local text of the last failing HIP call, "" if none */
const char* prot_last_error (void);
int prot_abi_version (void);

/* ---- device-resident streaming path (the roofline path) ----------------- */
#define PROT_F_STATELESS 1u /* history = silence, engine state untouched */
#define PROT_F_HILBERT   2u /* write the Hilbert branch u[t] only (rot ignored) */

/* y[c][t] = ca_c * x[c][t - dry_delay] + sa_c * (fir * x[c])[t - conv_delay]
 * for t in [0, n_out), x = [engine history | d_in[0..n_in) | silence].
 * d_in/d_out are DEVICE pointers, planar, channel c at + c*stride (floats).
 * rot is a HOST array [n_chn][2] = {ca, sa}.  Replaces the block body of
 * process_channel (src/phaserotate.c:629-720, steady branch :710-717) and
 * PhaseRotateProc::process (cli/phase-rotate.cc:181-232) for a whole run of
 * blocks per launch.  Unless PROT_F_STATELESS, the last `history` input
 * frames are kept for the next call.  Asynchronous on `stream`.
 * d_out must not overlap d_in (PROT_EINVAL): workgroups read the input of
 * neighbouring runs while others store output.  The LV2 front end, where hosts
 * do run in place (src/phaserotate.c:779-785), goes through its own buffers.
 * ONE STREAM AT A TIME PER ENGINE: every device call (this one, _ramp, _f16, prot_sweep_dev) stages its rotation,
 * angle list, history pair and workspace in buffers the engine owns.  Calls queued on the same stream are ordered;
 * before moving an engine to another stream the caller synchronises the old one (or makes the new stream wait on
 * an event recorded there).  Two streams on one engine at once race on those buffers.  Different engines are
 * independent -- one per thread / per plugin instance, like the reference's instances. */
int prot_apply_dev (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_in,
                    float* d_out, int64_t out_stride, uint64_t n_out,
                    const float* rot, uint32_t flags, void* stream);

/* same with a per-sample rotation (angle ramp, src/phaserotate.c:673-709):
 * d_rot is a DEVICE array [n_chn][n_out][2] = {ca, sa} per output frame. */
int prot_apply_dev_ramp (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_in,
                         float* d_out, int64_t out_stride, uint64_t n_out,
                         const float* d_rot, uint32_t flags, void* stream);

/* fp16 sample path (BASELINE config 5): d_in/d_out are DEVICE arrays of IEEE half (2 bytes per frame), planar;
 * packed-half butterflies, fp32 overlap-add and rotation; flags as prot_apply_dev (PROT_F_STATELESS; without it
 * the engine carries the fp16 stream's own input history from call to call -- a stream is EITHER fed through the
 * fp32 calls OR through the fp16 calls between two resets).  Every polyphase engine (LV2 at every rate, every CLI block size);
 * FIRs of more than 8192 taps (CLI blksiz 16384, 32768) take two / four launches whose shares are added in the half-precision
 * output -- one more rounding to half per launch, inside the path's stated tolerance (round 4; refused before).
 * Accuracy against the fp32 path on the same fp16-rounded input: rms ~2e-4, max ~2e-3 at |x| <= 0.5 (tests). */
int prot_apply_dev_f16 (prot_engine* e, const void* d_in, int64_t in_stride, uint64_t n_in,
                        void* d_out, int64_t out_stride, uint64_t n_out,
                        const float* rot, uint32_t flags, void* stream);
/* per-frame rotation, as prot_apply_dev_ramp: d_rot = DEVICE float[n_chn][n_out][2] {ca, sa} */
int prot_apply_dev_f16_ramp (prot_engine* e, const void* d_in, int64_t in_stride, uint64_t n_in,
                             void* d_out, int64_t out_stride, uint64_t n_out,
                             const float* d_rot, uint32_t flags, void* stream);

#define PROT_F_STREAM_START 4u /* first chunk of a file: block 0 gets the Q1 treatment */
#define PROT_F_STREAM_END   8u /* last chunk: zero-pad the partial block and add the flush block */

/* Rotated abs-max sweep of a device-resident planar stream (CLI engines only):
 * for angle = ang_start; ; angle += ang_stride (loop bounds exactly as
 * cli/phase-rotate.cc:409-428) d_peaks[c][a] = max(d_peaks[c][a], ...) over
 * all blocks of this chunk, with the first-block (cli/phase-rotate.cc:418-419),
 * angle-0 (:413-414), zero-pad and flush-block (:577-586) rules.  chn = -1:
 * all channels (cli/phase-rotate.cc:434-435).  d_peaks is a DEVICE array
 * [n_chn][PROT_MAXSAMPLE], max-accumulated (so it can be all-reduced with MAX
 * across GPUs); NULL = the engine's own table (see prot_cli_peaks).
 * Replaces PhaseRotate::analyze/thr_process + calc_rotated_peak +
 * dsp_compute_peak (cli/phase-rotate.cc:98-121, :388-444, cli/dsp_peak_calc.h)
 * and analyze_file's block loop (:565-587).  n_frames must be a multiple of
 * blksiz unless PROT_F_STREAM_END is set. */
int prot_sweep_dev (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_frames,
                    int ang_start, int ang_end, int ang_stride, int chn,
                    float* d_peaks, uint32_t flags, void* stream);

/* block until everything queued on the engine's stream -- and on the caller's stream the engine's last call named, whose
 * handle the engine lets go of here ("Streams" above) -- has finished */
int prot_engine_sync (prot_engine* e);

/* ---- host-buffer drop-ins (mirror the reference call for call) ---------- */
/* run()'s DSP for n_samples frames of every channel: in/out are HOST planar
 * port buffers (in == out allowed, src/phaserotate.c:779-785), angle_deg the
 * control-port values.  Handles any n_samples, the block-rate angle ramp
 * (:673-709) and the latency of `latency` frames.  Replaces
 * run()/process_channel (src/phaserotate.c:538-725, :840-843).
 * lvl (optional, [n_chn][2]) receives max|delayed in| is NOT computed here:
 * meters stay in host C (see host/lv2_plugin.c).
 * Calls of one block per channel go through ONE resident kernel per process that serves every eligible LV2 engine (a
 * workgroup per channel, a mailbox per engine): no launch and no stream synchronisation per call.  The
 * library stops and restarts it by itself around every other use of any engine; run() calls of different engines made
 * from different threads are serialised. */
int prot_lv2_run (prot_engine* e, const float* const* in, float* const* out,
                  uint32_t n_samples, const float* angle_deg);

/* The reference's run() never allocates (src/phaserotate.c:774-852).  Neither does prot_lv2_run: its staging
 * buffers are reserved by prot_engine_create for 8192 frames per call and raised here (instantiate / activate
 * time, e.g. from the host's options:maxBlockLength); longer calls are processed in chunks of that length. */
int prot_lv2_set_max_block (prot_engine* e, uint32_t frames);

/* DEFERRED run() (ABI 7; opt-in, default off): the mode in which prot_lv2_run never waits for the device, so that a bundle
 * may declare lv2:hardRTCapable again (lv2ttl/phaserotate.ports.in:7).  frames > 0: a call posts its block to the resident
 * kernel and returns at once with the output `frames` frames OLDER -- the blocking route's samples, delayed by exactly
 * `frames` more (the host adds it to the latency port, src/phaserotate.c:788; the first `frames` frames after a reset are
 * silence).  Calls longer than `frames` are taken in pieces of that length, and wait for the device between the pieces.  If the block of the call before is not back
 * when a call arrives -- a device that is late by a whole period -- silence goes out in its place and the call's own
 * frames are not posted: both are counted (prot_lv2_deferred_stats), nothing waits.  What still blocks: the first call
 * after a reset or after 0.25 s without calls (the kernel is started: 0.1-0.3 ms), a call while another thread creates,
 * destroys, resets or synchronises an engine (the launch route), calls of more than one transform block per channel.
 * While the mode is on the resident kernel has no age limit (it leaves when idle or told to).  frames = 0: off.
 * Call it where activate() may be called (it synchronises); PROT_EUNSUPPORTED for engines whose run() is launched
 * (more than 32 channels, pipelined, direct-form engines). */
int prot_lv2_set_deferred (prot_engine* e, uint32_t frames);
/* calls whose predecessor's block was late (silence went out in its place) / calls whose own frames were never posted */
int prot_lv2_deferred_stats (prot_engine* e, uint64_t* late, uint64_t* dropped);

/* Channel.angle (src/phaserotate.c:53, :560, :725): the interpolated rotation
 * state in turns after the last run; the host compares it with the port's
 * target to arm its delayed meter reset (src/phaserotate.c:611). */
int prot_lv2_get_angle (prot_engine* e, uint32_t chn, float* angle_turns);

/* PhaseRotate::apply (cli/phase-rotate.cc:467-485): one interleaved block of
 * blksiz frames, processed in place; angles are half-degree indices, reduced
 * (a + 360) % 360 exactly as :463 (Q9). */
int prot_cli_apply (prot_engine* e, float* buf, const int* angles);

/* PhaseRotate::analyze (cli/phase-rotate.cc:431-444): one interleaved block. */
int prot_cli_analyze (prot_engine* e, const float* buf, int ang_start, int ang_end,
                      int ang_stride, int chn, int start);

/* analyze_file (cli/phase-rotate.cc:565-587) on an in-memory interleaved
 * float file: whole-file device pass (upload, de-interleave, sweep).
 *
 * prot_cli_file_resident (e, 1): the engine keeps the de-interleaved file in device memory after the first
 * prot_cli_analyze_file of a (data, n_frames) pair, and later calls with the same pair sweep the resident copy without
 * touching the host link (the reference re-reads the file for every pass of its search, cli/phase-rotate.cc:815-929;
 * the upload is 10-40 x the sweep).  The caller promises not to change the file's samples in between; (e, 0) drops the
 * copy and the promise.  Files that do not fit (> 1/4 of the device's memory) stream as before. */
int prot_cli_file_resident (prot_engine* e, int on);
int prot_cli_analyze_file (prot_engine* e, const float* data, uint64_t n_frames,
                           int ang_start, int ang_end, int ang_stride, int chn);

/* the apply loop of main() (cli/phase-rotate.cc:950-1003) on an in-memory
 * interleaved float file: out receives n_frames frames with the blksiz/2
 * latency removed and the tail flushed (done right for any length -- the
 * reference's own tail handling is broken for partial blocks). */
int prot_cli_apply_file (prot_engine* e, const float* data, uint64_t n_frames,
                         const int* angles, float* out);

/* PhaseRotate::peak / peak_all (cli/phase-rotate.cc:275-299) */
float prot_cli_peak (prot_engine* e, int c, int a);
float prot_cli_peak_all (prot_engine* e, int a);
/* whole table, HOST array [n_chn][PROT_MAXSAMPLE] */
int prot_cli_peaks (prot_engine* e, float* out);
/* DEVICE pointer of the engine's own table (for an external MAX all-reduce) */
float* prot_cli_peaks_dev (prot_engine* e);

/* How much of the sweep the engine had to evaluate since the last reset.  The per-angle abs-max
 * (calc_rotated_peak, cli/phase-rotate.cc:98-121) is bounded by r = sqrt(dry^2 + hilbert^2) for
 * every angle, so a frame whose r stays below a lower bound of every final peak cannot change the
 * table: every frame is tested (once per workgroup, against the smallest bound of its four angle
 * groups), the survivors are compacted and only full 32-frame tiles of
 * survivors go through the per-angle evaluation -- the table is bit-identical to evaluating
 * everything.  Units are (wave, 32-frame tile) pairs: `seen` offered, `evaluated` run through the
 * per-angle evaluation (tiles of survivors + tiles fed untested); the ratio is signal-dependent
 * (stationary material: the share of frames near the peaks; a signal that keeps growing: 1).
 * When more than 7/8 of a 1024-frame step survive, the workgroup stops testing for a while (2 steps, doubling while
 * re-probes keep failing, back to 2 after one that succeeds); those tiles are `untested`.  Launches with the pruning switched off
 * (a test hook) run the plain kernel and count nothing.  Work on the engine's own stream is awaited first; launches on a caller's
 * stream are the caller's to await.
 *   prot_sweep_stats_ex: out = { evaluated, seen, untested, surviving frames of the tested tiles } */
int prot_sweep_stats (prot_engine* e, uint64_t* evaluated, uint64_t* seen);
int prot_sweep_stats_ex (prot_engine* e, uint64_t out[4]);
/* ABI 5.  Long sweeps of the engines whose Hilbert pass is the overlap-save kernel (blksiz 2048, 4096) FUSE a first test into that
 * pass: the Hilbert branch is never written, only the (dry, hilbert) pairs of the frames whose r exceeds the bound of the whole
 * call -- the table as it stands at launch, seeded by a pass over every 64th block when the call opens a stream or brings new
 * angles -- go to lists which the matrix-core kernel then evaluates as above (its statistics then count tiles of LISTED frames).
 * out = { channel-frames the fused pass has tested since the last reset, channel-frames it listed }.  Same tables, bit for bit. */
int prot_sweep_stats_fused (prot_engine* e, uint64_t out[2]);

/* ABI 6.  Sharded sweeps in two phases.  A shard that seeds its pruning bound from its own frames lists about twice what the whole
 * file lists against one bound; so: every shard calls prot_sweep_seed_dev on its data (ONLY the seeding pass of the fused route, over
 * about n_blocks device blocks spread through the call -- 64 / shards is plenty; nothing of the engine's state moves, the table gets
 * the sample's maxima), the tables are merged (MAX all-reduce), and prot_sweep_dev on the same data then skips its own seeding
 * and lists against the merged bound.  Every entry a seed raises is a peak of real frames -- the engine's history must be what
 * the main call will find (prot_engine_prime_dev FIRST: the seeding pass may sample the shard's first block) --, so the final
 * merged table is bit-identical to a single pass.  Engines / calls the fused route does not take seed nothing (PROT_OK).
 * prot_engine_prime_dev: n_frames of input pass into the history without being transformed -- what a shard does with the frames
 * in front of its range (cli/phase-rotate.cc: the previous block + the filter's memory). */
int prot_sweep_seed_dev (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_frames, int ang_start,
                         int ang_end, int ang_stride, int chn, float* d_peaks, uint32_t flags, int n_blocks, void* stream);
int prot_engine_prime_dev (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_frames, void* stream);
/* phase one of a shard in ONE call (a shard's fixed costs are what bounds the speed-up of a sharded sweep, and four short calls
 * from a host that is slower than the device's few microseconds per kernel leave gaps between them): prot_engine_reset, the
 * caller's table d_peaks ([n_chn][360], may be NULL: the engine's own) zeroed on `stream`, the n_halo frames in front of the shard
 * primed (d_halo NULL or n_halo 0: the file's first shard), the seeding pass over all channels.  The same state and the same table
 * as the four calls. */
int prot_sweep_shard_seed_dev (prot_engine* e, const float* d_halo, int64_t halo_stride, uint64_t n_halo, const float* d_in, int64_t in_stride,
                               uint64_t n_frames, int ang_start, int ang_end, int ang_stride, float* d_peaks, uint32_t flags, int n_blocks,
                               void* stream);

/* ---- several GPUs of one node, single process --------------------------- */
/* MAX all-reduce of the engines' OWN peak tables (prot_cli_peaks_dev) across
 * their GPUs: one ncclAllReduce(ncclMax, float[n_chn][360]) per engine inside
 * an RCCL group (ncclCommInitAll, one communicator per device, over xGMI).
 * 2.9 KB for stereo: latency-bound.  Engines must sit on distinct devices and
 * have equal n_chn; for engines sharing a device (tests) or when librccl is
 * not loadable the tables are merged through the host instead.  Replaces the
 * thread join + peak_all() max of cli/phase-rotate.cc:441-443, :287-299.
 * After the call every engine holds the merged table. */
int prot_allreduce_peaks (prot_engine* const* engines, int n);

/* analyze_file (cli/phase-rotate.cc:565-587) sharded over n engines: engine r
 * sweeps a run_align-aligned range of frame blocks for all requested angles, primed
 * with the input history in front of its range; the first-block rule stays
 * with engine 0, zero-padding and the flush block with the last one; in two
 * phases (seed, prot_allreduce_peaks, main pass, prot_allreduce_peaks: see
 * prot_sweep_seed_dev).  Bit-identical to prot_cli_analyze_file on one engine. */
int prot_cli_analyze_file_multi (prot_engine* const* engines, int n, const float* data, uint64_t n_frames,
                                 int ang_start, int ang_end, int ang_stride, int chn);

/* SinCosLut::sincos (cli/phase-rotate.cc:41-74) -- the table the device uses */
void prot_sincos_lut (int a, float* s, float* c);

#ifdef __cplusplus
}
#endif
#endif
