// prot_kernels.h -- launch interface between the C-ABI host code
// (prot_engine.hip, prot_apply.hip, prot_lv2.hip, prot_sweep_host.hip, prot_multi.hip) and the gfx950 kernels (prot_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace prot {

constexpr int kMaxSample = 360; // cli/phase-rotate.cc:39
constexpr int kRotInline = 16;  // channels whose steady rotation constants travel as kernel arguments

// One launch of the overlap-add kernel: every workgroup owns a run of
// consecutive hops of one channel-stream.  S = sample type in HBM (float or _Float16).
template <class S>
struct OlaParamsT {
	const S*      in;         // device, planar; channel c at in + c*in_stride
	int64_t       in_stride;
	int64_t       n_in;       // valid input frames; reads at >= n_in are silence
	const S*      hist;       // device, [n_chn][hist_len]: frames before in[0]; NULL = silence
	int32_t       hist_len;
	S*            hist_out;   // device, [n_chn][hist_len] or NULL: the kernel also writes the NEXT call's history
	                          // (the last hist_len frames of [hist | in]) here, every workgroup one slice;
	                          // must not alias hist or in
	S*            out;        // device, planar
	int64_t       out_stride;
	int64_t       n_out;      // emit t in [0, n_out)
	const float2* H;          // device, [partitions][fftlen] FIR spectrum / fftlen, natural bin order
	const float2* tw;         // device, [fftlen] e^{-2 pi i m / fftlen}
	const float*  rot;        // device, [n_chn][2] {ca, sa}              (steady), or NULL: rot_v
	const float*  rot_s;      // device, [n_chn][n_out][2] per-frame      (ramp) or NULL
	int32_t       run_frames; // outputs per workgroup, multiple of 2*hop
	int32_t       conv_delay; // Dc
	int32_t       dry_delay;  // Dd
	int32_t       hilbert_only;
	int32_t       accumulate; // add sa*u to out instead of storing ca*dry + sa*u (FIR partitions > 0)
	// polyphase kernel, one-round launches: runs of unequal length by dispatch rank (workgroup id / rank_cu = 0, 1, 2 (, 3):
	// the first, second, third (, fourth) workgroup a CU receives); rank_cu = 0: every run is run_frames long
	int32_t       rank_cu;
	int32_t       n_rank;     // 3 (fp32: 3 workgroups per CU) or 4 (fp16 samples)
	int32_t       rank_frames[4];
	// olasave_kernel, short launches (round 6): the last hist_extra workgroups of each channel (blockIdx.x >= gridDim.x - hist_extra) copy
	// the next history and do nothing else; the others skip their slice -- the copy's two round trips leave every block's chain
	int32_t       hist_extra;
	// steady rotation by value ({ca, sa} of channels 0..15, read when rot == NULL): two floats per channel ride in the
	// kernel arguments instead of a host-to-device copy in front of every launch (4 us on a 20 us launch)
	float         rot_v[2 * kRotInline];
	// olasave_kernel: per-engine constants (prot_engine_create): f2[12][256] split outer twiddles of thread t (forward at
	// the half-shifted index: W^{(t-1/2) i}, W^{(t-1/2) 4i}; inverse: W^{t i}, W^{4 t i}; i = 1..3), f2[256] the
	// half-shifted W256 table W256^{(c-1/2) k2} at [16 k2 + c], f2[256] the |B| factors of the rows next to DC
	// ((rho[q], rho[q + 3840]) of thread t, q = (t >> 4) + 16 (t & 15)), f2[256] the plain W256 table; sv_gain = mean |B| / 4096
	const float*  sv;
	float         sv_gain;
	int32_t       sv_flags;   // 1: every block through the C++ form (cross-check of the generated loop)
	// olasave_kernel, pipelined stateful launches (PROT_CFG_PIPELINE): consecutive launches run on different streams and hand
	// the history over on the device.  hs_flag != NULL: every workgroup writes its history slice FIRST and counts it in
	// *hs_cnt; the last one stores hs_post (this launch's generation) into hs_flag[hs_post & 7].  Workgroups that read
	// history wait until hs_flag[hs_wait & 7] holds hs_wait (0: nothing to wait for -- the history they read was ordered
	// by the stream).  One slot per generation, NOT one running maximum: with three streams launch g + 1 may publish before
	// launch g - 1 has (its history write waits for nothing), and a reader of generation g - 1 must not mistake that for it;
	// a slot is reused by generation g + 8, which cannot start before g is complete (stream order, and every launch's
	// readers wait for the launch before it: the lead is 5 at most).
	uint32_t*     hs_flag;      // (word kPipeStatusWord of the same array: generation whose wait timed out, 0 = none)
	uint32_t*     hs_cnt;
	uint32_t      hs_wait, hs_post;
	// olasave_kernel, MODE_SWEEP (surv != NULL): the Hilbert pass of the CLI sweep with the sweep's test fused into it -- nothing is
	// written to `out`; for every frame t in [sw_t_begin, n_out) the pair d = x[t - dry_delay] (0 while t < sw_dry_from, Q1),
	// h = hilbert[t] is appended to the list of the wave that computed it if d^2 + h^2 exceeds the bound: the smallest entry of
	// sw_peaks[c][sw_angles[..]] as it stands at launch, squared, less the margin of prot_sweep.hip (sw_thr_zero: no bound --
	// the seeding pass over a sparse sample of the blocks, run_stride).  List of wave w of workgroup g of channel c: surv +
	// ((c * gridDim.x + g) * 4 + w) * surv_cap, its length in surv_cnt at the same index.  max |x| over the frames that pass
	// through goes to sw_peaks[c][sw_raw_slot] (>= 0; sw_raw_tail: the call's last dry_delay input frames too).
	float2*       surv;
	int64_t       surv_cap;
	uint32_t*     surv_cnt;    // [2][n_chn][gridDim.x][4]: the list lengths, and behind them the bound every wave parks for itself
	float*        sw_peaks;
	const int*    sw_angles;
	int32_t       sw_n_angles;
	int32_t       sw_raw_slot;
	int32_t       sw_raw_tail;
	int32_t       sw_thr_zero;
	int64_t       sw_t_begin;
	int64_t       sw_dry_from;
	int32_t       run_stride; // > 0: workgroup g's run starts at frame g * run_stride (instead of g * run_frames)
	// olasave8k_dyn_kernel (round 5): the channel's whole in-range blocks [dyn_lo, dyn_lo + dyn_nb) are not dealt out as runs but
	// CLAIMED one at a time by the resident workgroups (eight contiguous regions, one per XCD, each behind a counter of its own;
	// a workgroup draws from its XCD's region first and moves on when it is empty); the blocks in front of and behind them -- history,
	// silence, a partial last block -- are staged through LDS by the first workgroups as before.  dyn_ctr: channel c's record at
	// dyn_ctr + c * kDynWords -- counters at words 0, 32, ... 224 (128 bytes apart), the exit count at word 256; all zero at launch,
	// left zero by the last workgroup to leave.
	uint32_t*     dyn_ctr;
	int64_t       dyn_lo, dyn_nb;
};
constexpr int kDynWords = 512; // per channel: 2 KiB
typedef OlaParamsT<float>    OlaParams;
typedef OlaParamsT<_Float16> OlaParamsH; // fp16 sample path (BASELINE config 5)

// fftlen 4096; hop = 256 * hopseg, hopseg in {4, 8}
hipError_t launch_ola4096 (const OlaParams& p, int hopseg, int n_chn, hipStream_t s);
// 8192-point overlap-save blocks claimed dynamically (prot_olasave8k.hip); grid_x resident workgroups per channel
hipError_t launch_ola_save8k_dyn (const OlaParams& p, int hop_rows, int n_chn, int grid_x, hipStream_t s);

// two 2048-tap FIR partitions sharing one forward transform (frequency-domain delay line);
// p.H = [2][4096] spectra, p.conv_delay = delay of partition 0, (dry_delay - conv_delay)/256 in {6, 8},
// p.run_frames a multiple of 4096
hipError_t launch_ola4096_p2 (const OlaParams& p, int n_chn, hipStream_t s);

// polyphase form for FIRs that are zero at every even tap (the Hilbert FIRs of > 2049 taps):
// p.H = [n_part][4096] spectra of the 2048-tap partitions of g[s] = fir[2 s + 1] (/ 4096),
// n_part in {1, 2}; hop_rows in {8, 10, 12, 14} (x 512 new frames per block; > 8 only for one partition
// of at most 256 (16 - hop_rows) + 1 taps); p.conv_delay even; p.run_frames a multiple of 512 hop_rows
hipError_t launch_ola_poly (const OlaParams& p, int n_part, int hop_rows, int n_chn, hipStream_t s);
// the same on fp16 samples (no accumulating launches: FIRs of up to 8192 taps): BASELINE config 5
hipError_t launch_ola_poly_f16 (const OlaParamsH& p, int n_part, int hop_rows, int n_chn, hipStream_t s);
// ... one partition, steady rotation, with the radix-16 passes on the matrix core (prot_f16mfma.hip, round 6)
hipError_t launch_ola_poly_f16_mfma (const OlaParamsH& p, int hop_rows, int n_chn, hipStream_t s);
bool       mfma_f16_takes (const OlaParamsH& p, int hop_rows); // does a call qualify (delays, mode)?
int        mfma_f16_wg_per_cu ();                              // resident workgroups per CU its launches are sized for

// overlap-save, table-free form of launch_ola_poly for one partition of an antisymmetric g (prot_olasave.hip): four
// workgroups per CU; hop_rows in {8, 10, 12}, n_rho_rows = 1; p.sv / p.sv_gain set; dry_delay even
hipError_t launch_ola_save (const OlaParams& p, int hop_rows, int n_rho_rows, int n_chn, hipStream_t s);
// the same on 8192-point blocks (prot_olasave8k.hip): 512-thread workgroups, two per CU; hop_rows in {12, 13} (x 1024 new
// frames per block: firlen = 1024 (16 - hop_rows)); p.sv = the five-row table of save_tables8k; no pipelined launches
// ---- the run() server (prot_olasave512.hip): the mailbox of one LV2 engine, in two halves.
// SvServerBox (host -> device): what the host writes and the resident kernel reads -- in DEVICE memory that the CPU stores into through
// the PCIe BAR when the device has a large one (write-combined, posted: 4 KiB in ~0.2 us; the kernel then polls and fetches LOCAL
// memory instead of reading host memory over the link at ~1 GB/s), else in pinned host-coherent memory.  The call's input frames
// ([channel][frames]) and per-frame rotation ([channel][frames][2]) follow it at kServerInOffset.  The host never READS this half
// (a CPU load through the BAR is 5-7 us).
// SvServerAck (device -> host): pinned host-coherent memory the kernel stores into and the host polls.
constexpr int    kServerMaxChn   = 32;
constexpr size_t kServerInOffset = 1024; // bytes from the box to the input frames
// The 512-thread block's constants (prot_olasave512.hip; tools/model/olasave512_model.py tables()), in float2 behind the fifteen rows of
// the 256-thread block's table: six twiddle tables [j - 1][index], j = 1..7 -- forward W4096^{(T - 1/2) j} (512 indices),
// W512^{(t' - 1/2) j} (64), W64^{(d - 1/2) j} (8); inverse conj W4096^{t'' j}, conj W512^{u j}, conj W64^{k1 j} -- and per thread
// (k1 = T >> 6, k2 = (T >> 3) & 7, k3 = T & 7) the |B| factors of its k4 = 0 and k4 = 7 bins over their mean (1 outside the rows next to DC)
constexpr int kSv5Offset = 15 * 256;
constexpr int kSv5fA = 0, kSv5iA = 3584, kSv5fB = 7168, kSv5iB = 7616, kSv5fC = 8064, kSv5iC = 8120, kSv5Rho = 8176, kSv5Size = 8688;
// frames in front of a call's input that the server's workgroup keeps in LDS from call to call (round 6): what a block reaches back
// (dry delay + the context rows in front of the first dry row + 1) has to fit -- post_ola_save_server checks
constexpr int    kSrvHist        = 6144;
constexpr int    kSrvIn          = 512 * 10; // the input of a call (at most a block's new rows) and the silence behind it
struct SvServerBox {
	// ONE 8-byte store, LAST: the doorbell.  bits 0..31 the call's number (1 .. 2^30, never the same twice in a row), bits 32..62 frames
	// per channel of the call, bit 63: leave
	uint64_t          bell;
	uint64_t          pad[7];
	OlaParamsT<float> p;        // what the launch would have carried
};
static_assert (sizeof (SvServerBox) <= kServerInOffset, "the input frames follow the box");
struct SvServerAck {
	uint32_t alive;    // set by the host before the launch, cleared by the kernel as its last store
	uint32_t pad0[15]; // how far each channel's workgroup has come (diagnostics)
	uint32_t done[kServerMaxChn]; // done[c] = the call's number when channel c's output (and its next history) is complete
};
// the slots of ONE resident kernel (round 5): a workgroup per channel of every engine it serves
struct SvServerSlot {
	const SvServerBox* box;
	SvServerAck*       ack;
	unsigned*          relay;    // device words of the slot's engine: its channel 0 relays the doorbell to its other channels
	unsigned           last_seq; // the number of the last call the engine's server took
	int                chn, n_chn; // the slot's channel / the channels of its engine
	int                hop_rows; // 8 or 10: the engine's block (olasave_body<HS>)
	int                no_life_cap; // 1: the engine's workgroups leave when idle or told to, never by age (deferred run())
};
struct SvServerSlots {
	int          n;
	SvServerSlot s[kServerMaxChn];
};
hipError_t launch_ola_save_server (const SvServerSlots& slots, hipStream_t s);
bool       ola_save512_takes (const OlaParamsT<float>& p, int hop_rows, int n_rho_rows, int n_chn, int max_wg);
hipError_t launch_ola_save512 (const OlaParamsT<float>& p, int hop_rows, int n_chn, hipStream_t s);
hipError_t post_ola_save_server (const OlaParamsT<float>& p, int hop_rows, int n_rho_rows, int n_chn, SvServerBox* box);
hipError_t launch_ola_save8k (const OlaParams& p, int hop_rows, int n_chn, hipStream_t s);
// the same on 16384-point blocks (prot_olasave16k.hip: 1024-thread workgroups, hop_rows = 8: CLI blksiz 16384)
hipError_t launch_ola_save16k (const OlaParams& p, int hop_rows, int n_chn, hipStream_t s);

// fp16 sample path (BASELINE config 5): half samples in HBM, packed-half butterflies,
// fp32 overlap-add + rotation.  Stateless, steady rotation.
hipError_t launch_ola4096_f16 (const OlaParamsH& p, int hopseg, int n_chn, hipStream_t s);
hipError_t launch_ola4096_p2_f16 (const OlaParamsH& p, int n_chn, hipStream_t s);

// Rotated abs-max sweep over t in [t_begin, t_end) of one chunk:
//   v = ca[a] * (t_abs >= dry_from ? x[t - delay] : 0) + sa[a] * u[t]
//   peaks[c][a] = max(peaks[c][a], |v|)
constexpr unsigned kPipeStatusWord = 12;       // of prot_engine::d_pipe
constexpr unsigned kPipeSpinMax    = 1u << 21; // ~1 us per look: about two seconds

struct SweepParams {
	const float* x;           // device planar input (dry), channel stride x_stride
	int64_t      x_stride;
	int64_t      n_x;         // valid frames of x (beyond: silence)
	const float* hist;        // [n_chn][hist_len] frames before x[0], or NULL
	int32_t      hist_len;
	const float* u;           // device planar Hilbert branch, u[t] for t in [0, n_u)
	int64_t      u_stride;
	int64_t      t_begin, t_end;
	int64_t      t_abs0;      // absolute stream position of local t = 0
	int64_t      dry_from;    // dry term forced to 0 while t_abs < dry_from (Q1)
	int32_t      delay;       // blksiz / 2
	const int*   angles;      // device, [n_angles] LUT indices 0..359 (never the raw-peak angle 0 entry)
	int32_t      n_angles;
	const float* lut;         // device, [360][2] = {sin, cos}  (SinCosLut)
	float*       peaks;       // device, [n_chn][360], max-accumulated
	int32_t      chn0;        // first channel handled by blockIdx.y = 0
	int32_t      prune;       // evaluate only the frames that can still raise a maximum (exact; see prot_sweep.hip)
	float*       thr;         // device, [n_chn][4], all zeros at launch (the merge kernel of every launch leaves it so): shared pruning bounds, or NULL
	unsigned*    ctr;         // device, [n_chn][32][32] (32 counters per channel, one cache line each), all zeros at launch (the merge
	                          // kernel leaves them so): the next step to hand out
	int32_t      raw_slot;    // >= 0: also max |x| over the frames the launch reads -> peaks[c][raw_slot] (angle 0, Q4); -1: no
	int32_t      raw_tail;    // with raw_slot: the last `delay` frames of x too (a chunk that is not the stream's last: the loaders see them
	                          // only as the next chunk's history, and calc_peak, cli/phase-rotate.cc:413-414, takes the whole new block)
	float*       part;        // device scratch, >= sweep_scratch_floats () floats: per-workgroup maxima of one launch (merged into peaks)
	size_t       part_floats;
	unsigned long long* stats; // device, [4] += {tiles evaluated, tiles seen, tiles fed untested, surviving frames of tested tiles}, or NULL
	// the LISTS as the source (surv != NULL; round 4; "compact" in launch_sweep): the frames come as (d, h) pairs from the lists the fused Hilbert pass left
	// (olasave_kernel MODE_SWEEP) instead of from x, hist and u: step gs of channel c is desc[c * desc_stride + gs] = {first entry
	// (index into surv), entries (1..1024)}, n_steps[c] steps in all (sweep_plan_kernel); x, u, t_begin .. delay are unused
	const float2*   surv;
	const uint2*    desc;
	int64_t         desc_stride;
	const unsigned* n_steps;
	// ... evaluated by sweep_eval_kernel, which tests every listed frame against each BLOCK of 32 angles: thr_blk = device,
	// [n_chn][16] shared per-block bounds (zeros at launch, left so by the merge kernel); sect[b] = {cos psi, sin psi, cos D, sin D}
	// of block b's angles: all of them lie within D of psi (modulo pi; margins included; D <= pi / 2), launch_sweep fills it
	// from angles_host
	float*          thr_blk;
	const int*      angles_host; // the same n_angles LUT indices on the host (lists only)
	float           sect[12][4];
};
constexpr int kSweepThrFloats = 4 + 16; // per channel: thr [n_chn][4], then thr_blk [n_chn][16]
// the step table of a sweep over lists from the list lengths: cnt[c][n_lists] -> desc, n_steps; list l of channel c starts at entry
// (c * n_lists + l) * cap
// (total, optional: += the number of entries of all lists)
hipError_t launch_sweep_plan (const unsigned* cnt, int n_lists, int64_t cap, uint2* desc, int64_t desc_stride, unsigned* n_steps, int n_chn,
                              unsigned long long* total, hipStream_t s);
hipError_t launch_sweep (const SweepParams& p, int n_chn, hipStream_t s);
// the seeding pass's lists against every requested angle by plain vector arithmetic, folded into the table scaled down by 1e-4
hipError_t launch_sweep_seed_eval (const float2* surv, const unsigned* cnt, int n_lists, int64_t cap, const int* angles, int n_angles, const float* lut,
                                   float* peaks, int chn0, int n_chn, float* part, size_t part_floats, hipStream_t s);
size_t     sweep_scratch_floats ();

// peaks[c][slot] = max(peaks[c][slot], max_{t in [0,n)} |x[c][t]|)   (angle 0, Q4)
hipError_t launch_absmax (const float* x, int64_t x_stride, int64_t n, float* peaks, int slot,
                          int chn0, int n_chn, hipStream_t s);

// hist_out[c][i] = virtual(n_in + shift - hist_len + i), virtual = [hist_in | in | silence]  (shift: frames of silence that
// followed the input, e.g. the zero padding and the flush block of a file's last chunk)
hipError_t launch_hist_update (const float* in, int64_t in_stride, int64_t n_in,
                               const float* hist_in, float* hist_out, int hist_len,
                               int n_chn, hipStream_t s, int64_t shift = 0);
hipError_t launch_hist_update_f16 (const _Float16* in, int64_t in_stride, int64_t n_in,
                                   const _Float16* hist_in, _Float16* hist_out, int hist_len,
                                   int n_chn, hipStream_t s);

// up to six regions of 32-bit words zeroed by ONE launch (prot_engine_reset: the separate memsets were 8 launches, ~60 us)
constexpr int kZeroRegions = 6;
struct ZeroRegions {
	uint32_t* p[kZeroRegions];
	uint64_t  n[kZeroRegions]; // words
};
hipError_t launch_zero_regions (const ZeroRegions& z, hipStream_t s);

// interleaved [n][n_chn] <-> planar [n_chn][stride]
hipError_t launch_deinterleave (const float* inter, float* planar, int64_t stride, int64_t n,
                                int n_chn, hipStream_t s);
hipError_t launch_interleave (const float* planar, int64_t stride, float* inter, int64_t n,
                              int n_chn, hipStream_t s);

} // namespace prot
