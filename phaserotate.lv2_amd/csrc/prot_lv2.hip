// prot_lv2.hip -- the run() server (one resident kernel per process), prot_lv2_run and its block-rate angle machine, the deferred run()
// (split from prot_capi.hip in round 6: same code; what the units share is in prot_engine.h)
#include "prot_engine.h"

namespace prot_host {

// the buffers one prot_lv2_run chunk of `frames` frames needs: allocated by prot_engine_create (8192 frames) and
// prot_lv2_set_max_block, never by the audio callback
int lv2_reserve (prot_engine* e, uint32_t frames)
{
	const size_t n = frames, nc = e->info.n_chn;
	int          rc;
	if ((rc = e->h_io.ensure (n * nc * 4 + 2 * nc)) || (rc = e->d_in.ensure (n * nc)) || (rc = e->d_out.ensure (n * nc)) ||
	    (rc = e->d_rot_s.ensure (2 * n * nc))) {
		return rc;
	}
	e->lv2_max_block = std::max (e->lv2_max_block, frames);
	if (e->srv_eligible) { // the server's mailbox: box, then [nc][frames] input, then [nc][frames][2] per-frame rotation
		const size_t need = prot::kServerInOffset + 3 * (size_t)e->lv2_max_block * nc * sizeof (float);
		if (need > e->srv_box_bytes) {
			if (e->srv_box) {
				(void)(e->srv_bar ? hipFree (e->srv_box) : hipHostFree (e->srv_box));
				e->srv_box = nullptr;
			}
			int large_bar = 0;
			(void)hipDeviceGetAttribute (&large_bar, hipDeviceAttributeIsLargeBar, e->device);
			void* q = nullptr;
			e->srv_bar = large_bar != 0 && test_opt (PROT_TEST_LV2_NO_BAR) == 0 && hipExtMallocWithFlags (&q, need, hipDeviceMallocFinegrained) == hipSuccess;
			if (!e->srv_bar && hipHostMalloc (&q, need, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
				e->srv_eligible  = false; // (no mailbox: every call is launched)
				e->srv_box_bytes = 0;
				return PROT_OK;
			}
			e->srv_box       = (char*)q;
			e->srv_box_bytes = need;
			if (e->srv_bar) {
				HIP_TRY (hipMemset (q, 0, sizeof (prot::SvServerBox)));
				HIP_TRY (hipDeviceSynchronize ());
			}
			// (every page touched by the CPU now: the first store into a page through the BAR costs 20-30 us, tools/attic/bar_probe.py --
			// not in the audio callback)
			memset (q, 0, need);
			_mm_sfence ();
		}
	}
	return PROT_OK;
}

// runtime multiplexes all streams of a process onto a few: one on a high-priority stream of its own keeps out of the way of
// every other stream; a second one would sooner or later share a queue with the first and wait behind it).  Round 5: that one
// kernel serves EVERY LV2 engine of the process that is eligible -- a workgroup ("slot") per channel, a mailbox and a doorbell
// per engine -- up to kServerMaxChn channels in all on one device; an engine beyond that takes the launch route.  The ROSTER
// (g_srv_roster) lists the engines whose run() went through the server; whoever starts the kernel starts it for the whole
// roster, so that two plugin instances called in turn do not keep replacing each other's server.
// Every other use of any engine -- a launch on an engine stream, a reset, a synchronisation, create / destroy (hipFree waits
// for the whole device) -- first tells the server to leave and waits for it; the next run() of a roster engine starts it again.
std::mutex                g_srv_mu;
std::vector<prot_engine*> g_srv_roster;           // engines the next kernel is started for
std::vector<prot_engine*> g_srv_members;          // engines the running kernel has slots for (srv_running each)
hipStream_t               g_srv_stream = nullptr; // the stream the running kernel was launched on (its starter's)
// Calls that allocate, free or synchronise the whole device (create, destroy, reserve, reset, sync) HOLD the server off for
// their whole duration: stopping it once is not enough when another thread's run() -- an audio thread calling every few
// milliseconds -- may start it again while hipMalloc / hipFree / hipDeviceSynchronize are still waiting for a device without a
// resident kernel.  run() takes the launch route while the count is non-zero.
int g_srv_hold = 0;

// deferred run(): wait (bounded) until the block the engine posted last has been served -- the server is about to be told to leave
void rt_wait_pending (prot_engine* o)
{
	if (!o->rt_pending || o->rt_pend_ready) {
		return;
	}
	prot::SvServerAck* a  = o->srv_ack.p;
	const auto         t0 = std::chrono::steady_clock::now ();
	for (uint32_t c = 0; c < o->info.n_chn;) {
		if (__atomic_load_n (&a->done[c], __ATOMIC_ACQUIRE) == o->rt_pend_seq) {
			++c;
			continue;
		}
		if (__atomic_load_n (&a->alive, __ATOMIC_ACQUIRE) == 0u || std::chrono::steady_clock::now () - t0 > std::chrono::seconds (2)) {
			return; // (left without it, or not answering: rt_collect deals with that)
		}
	}
	o->rt_pend_ready = true; // (its output stays in the pinned block until the next run() takes it)
}

int server_stop_locked ()
{
	if (g_srv_members.empty ()) {
		return PROT_OK;
	}
	std::vector<prot_engine*> members;
	members.swap (g_srv_members);
	for (prot_engine* o : members) { // every engine's workgroups leave on their own doorbell
		if (o->srv_running) {
			prot::SvServerBox* b = reinterpret_cast<prot::SvServerBox*> (o->srv_box);
			rt_wait_pending (o); // (deferred run(): the block it posted last is still the server's to do -- a leave command would overwrite its doorbell)
			o->srv_seq           = (o->srv_seq % (1u << 30)) + 1u;
			_mm_sfence ();
			__atomic_store_n (&b->bell, (uint64_t)o->srv_seq | ((uint64_t)1 << 63), __ATOMIC_RELEASE);
			_mm_sfence ();
		}
	}
	hipError_t err = g_srv_stream ? hipStreamSynchronize (g_srv_stream) : hipSuccess;
	for (prot_engine* o : members) {
		o->srv_running = false;
	}
	g_srv_stream = nullptr;
	return err == hipSuccess ? PROT_OK : hip_fail (err, "run() server stop");
}
int server_stop_any ()
{
	std::lock_guard<std::mutex> lk (g_srv_mu);
	return server_stop_locked ();
}
ServerHold::ServerHold ()
{
	std::lock_guard<std::mutex> lk (g_srv_mu);
	++g_srv_hold;
	(void)server_stop_locked ();
}
ServerHold::~ServerHold ()
{
	std::lock_guard<std::mutex> lk (g_srv_mu);
	--g_srv_hold;
}
bool server_is_member (const prot_engine* e)
{
	return std::find (g_srv_members.begin (), g_srv_members.end (), e) != g_srv_members.end ();
}
// (g_srv_mu held) may e join the roster: same device as the others, a slot left for each of its channels
bool server_roster_takes (prot_engine* e)
{
	if (std::find (g_srv_roster.begin (), g_srv_roster.end (), e) != g_srv_roster.end ()) {
		return true;
	}
	size_t chn = e->info.n_chn;
	for (const prot_engine* o : g_srv_roster) {
		if (o->device != e->device) {
			return false;
		}
		chn += o->info.n_chn;
	}
	return chn <= (size_t)prot::kServerMaxChn;
}
void server_roster_remove (prot_engine* e)
{
	g_srv_roster.erase (std::remove (g_srv_roster.begin (), g_srv_roster.end (), e), g_srv_roster.end ());
}
// (g_srv_mu held) start the server for the roster (e joins it) behind whatever the engines' own streams still have to do; a
// kernel that is still there -- serving others, or on its way out -- leaves first
int server_start_locked (prot_engine* e)
{
	int rc = server_stop_locked ();
	if (rc) {
		return rc;
	}
	if (e->srv_running) { // (a server of its own that left by itself: over or about to be)
		HIP_TRY (hipStreamSynchronize (e->srv_stream));
		e->srv_running = false;
	}
	if (std::find (g_srv_roster.begin (), g_srv_roster.end (), e) == g_srv_roster.end ()) {
		g_srv_roster.push_back (e);
	}
	prot::SvServerSlots slots {};
	// (a failure anywhere below leaves NO engine of the roster marked alive with a kernel that never started)
	auto undo = [&] () {
		for (prot_engine* o : g_srv_roster) {
			o->srv_ack.p->alive = 0u;
		}
	};
	for (prot_engine* o : g_srv_roster) {
		hipError_t herr = hipEventRecord (o->srv_ev, o->stream);
		if (herr == hipSuccess) {
			herr = hipStreamWaitEvent (e->srv_stream, o->srv_ev, 0);
		}
		if (herr != hipSuccess) {
			undo ();
			return hip_fail (herr, "run() server start (ordering behind an engine's stream)");
		}
		prot::SvServerBox* b = reinterpret_cast<prot::SvServerBox*> (o->srv_box);
		prot::SvServerAck* a = o->srv_ack.p;
		__atomic_store_n (&b->bell, (uint64_t)o->srv_seq, __ATOMIC_RELEASE);
		_mm_sfence ();
		__atomic_store_n (&a->alive, 1u, __ATOMIC_RELEASE);
		for (uint32_t c = 0; c < o->info.n_chn; ++c) {
			prot::SvServerSlot& sl = slots.s[slots.n++];
			sl.box      = b;
			sl.ack      = a;
			sl.relay    = o->d_srv_relay.p;
			sl.last_seq = o->srv_seq;
			sl.chn      = (int)c;
			sl.n_chn    = (int)o->info.n_chn;
			sl.hop_rows = o->hopseg / 2;
			sl.no_life_cap = o->rt_frames > 0 ? 1 : 0; // (deferred run(): a restart every two seconds would be two buffers of silence)
		}
	}
	hipError_t err = prot::launch_ola_save_server (slots, e->srv_stream);
	if (err != hipSuccess) {
		undo ();
		return hip_fail (err, "run() server launch");
	}
	for (prot_engine* o : g_srv_roster) {
		o->srv_running = true;
	}
	g_srv_members = g_srv_roster;
	g_srv_stream  = e->srv_stream;
	e->srv_starts++;
	return PROT_OK;
}

// deferred run(): the resident kernel is started where the host may wait -- activate() / prot_lv2_set_deferred -- and waited for
// until its workgroups poll their doorbells: the first start of a process loads the code object (tens to hundreds of
// milliseconds), and a run() that does not wait would hand out silence for all that time
// (g_srv_mu held) start the server and wait until e's workgroups poll their doorbells
int server_start_wait_locked (prot_engine* e)
{
	prot::SvServerAck* a = e->srv_ack.p;
	const uint32_t     nc = std::min<uint32_t> (e->info.n_chn, 8u);
	for (uint32_t c = 0; c < nc; ++c) {
		__atomic_store_n (&a->pad0[c], 0u, __ATOMIC_RELAXED);
	}
	const int rc = server_start_locked (e);
	if (rc) {
		return rc;
	}
	const auto t0 = std::chrono::steady_clock::now ();
	for (uint32_t c = 0; c < nc;) {
		if (__atomic_load_n (&a->pad0[c], __ATOMIC_ACQUIRE) != 0u) {
			++c;
			continue;
		}
		if (std::chrono::steady_clock::now () - t0 > std::chrono::seconds (5)) {
			break; // (not fatal: the first calls hand out silence until it is there)
		}
	}
	return PROT_OK;
}
int rt_prestart (prot_engine* e)
{
	if (e->rt_frames == 0 || !e->srv_eligible) {
		return PROT_OK;
	}
	std::lock_guard<std::mutex> lk (g_srv_mu);
	if (g_srv_hold != 0 || !server_roster_takes (e)) {
		return PROT_OK; // (another thread holds the server off, or the roster is full: the first run() sorts it out)
	}
	return server_start_wait_locked (e);
}

// ---- deferred run() (prot_lv2_set_deferred): the host ring and the block that is with the server ----
// The ring holds rt_frames frames when a call arrives at rt_pop: every call pushes the block of the call BEFORE it (its output, or
// silence when that is late / was never posted) and pops its own length, so what a call returns is the blocking route's output
// rt_frames frames later -- whatever the calls' lengths (each <= rt_frames).
void rt_push (prot_engine* e, const float* src, uint32_t n)
{
	const uint32_t nc = e->info.n_chn, cap = e->rt_cap;
	if (n > cap - e->rt_fill) {
		n = cap - e->rt_fill; // (cannot happen while every call is <= rt_frames)
	}
	const uint32_t wr = (e->rt_rd + e->rt_fill) % cap, n0 = std::min (n, cap - wr);
	for (uint32_t c = 0; c < nc; ++c) {
		float* r = e->rt_ring.data () + (size_t)c * cap;
		if (src) {
			memcpy (r + wr, src + (size_t)c * n, n0 * sizeof (float));
			memcpy (r, src + (size_t)c * n + n0, (n - n0) * sizeof (float));
		} else {
			memset (r + wr, 0, n0 * sizeof (float));
			memset (r, 0, (n - n0) * sizeof (float));
		}
	}
	e->rt_fill += n;
}
void rt_pop (prot_engine* e, float* const* out, uint32_t off, uint32_t n)
{
	const uint32_t nc = e->info.n_chn, cap = e->rt_cap;
	const uint32_t m = std::min (n, e->rt_fill), n0 = std::min (m, cap - e->rt_rd);
	for (uint32_t c = 0; c < nc; ++c) {
		const float* r = e->rt_ring.data () + (size_t)c * cap;
		memcpy (out[c] + off, r + e->rt_rd, n0 * sizeof (float));
		memcpy (out[c] + off + n0, r, (m - n0) * sizeof (float));
		if (m < n) {
			memset (out[c] + off + m, 0, (n - m) * sizeof (float));
		}
	}
	e->rt_rd   = (e->rt_rd + m) % cap;
	e->rt_fill -= m;
}
void rt_ring_reset (prot_engine* e)
{
	std::fill (e->rt_ring.begin (), e->rt_ring.end (), 0.f);
	e->rt_rd = 0;
	e->rt_fill = e->rt_frames;
	e->rt_pending = e->rt_pend_ready = e->rt_discard = false;
	e->rt_skip_n = 0;
}
// the block posted by the call before this one: into the ring if it is there (ONE look at the done words, no waiting), silence in
// its place -- and its output thrown away when it arrives -- if it is not
int rt_collect (prot_engine* e, bool wait)
{
	if (e->rt_pending) {
		prot::SvServerAck* a = e->srv_ack.p;
		if (wait) { // (the second and later pieces of ONE call longer than rt_frames: there is no period between them to wait out)
			rt_wait_pending (e);
		}
		bool done = e->rt_pend_ready;
		auto look = [&] () {
			bool all = true;
			for (uint32_t c = 0; c < e->info.n_chn && all; ++c) {
				all = __atomic_load_n (&a->done[c], __ATOMIC_ACQUIRE) == e->rt_pend_seq;
			}
			return all;
		};
		if (!done) {
			done = look ();
		}
		if (!done && __atomic_load_n (&a->alive, __ATOMIC_ACQUIRE) == 0u) {
			// the server left (idle time-out) between the check in front of the doorbell and the doorbell: started again with the
			// number of the call before, it takes the block at once (what the blocking route does in its wait loop; rare, and it blocks)
			std::lock_guard<std::mutex> lk (g_srv_mu);
			int rc = server_stop_locked ();
			if (rc) {
				return rc;
			}
			done = look ();
			if (!done && g_srv_hold == 0) {
				prot::SvServerBox* b = reinterpret_cast<prot::SvServerBox*> (e->srv_box);
				e->srv_running = false;
				e->srv_seq     = e->rt_pend_seq - 1u;
				rc             = server_start_locked (e);
				e->srv_seq     = e->rt_pend_seq;
				if (rc) {
					return rc;
				}
				_mm_sfence ();
				__atomic_store_n (&b->bell, (uint64_t)e->rt_pend_seq | ((uint64_t)e->rt_pend_n << 32), __ATOMIC_RELEASE);
				_mm_sfence ();
			}
		}
		if (done) {
			if (!e->rt_discard) {
				rt_push (e, e->rt_pend_out, e->rt_pend_n);
			}
			e->rt_pending = e->rt_pend_ready = e->rt_discard = false;
		} else if (!e->rt_discard) {
			rt_push (e, nullptr, e->rt_pend_n);
			e->rt_discard = true;
			e->rt_late++;
#ifdef PROT_RT_DEBUG
			fprintf (stderr, "rt: late seq %u done %u %u alive %u stages %u %u bellseq %u running %d\n", e->rt_pend_seq, a->done[0], a->done[1], a->alive, a->pad0[0], a->pad0[1], e->srv_seq, (int)e->srv_running);
#endif
		}
#ifdef PROT_RT_DEBUG
		else if (!done) {
			fprintf (stderr, "rt: still late seq %u done %u %u alive %u stages %u %u\n", e->rt_pend_seq, a->done[0], a->done[1], a->alive, a->pad0[0], a->pad0[1]);
		}
#endif
	}
	if (e->rt_skip_n) { // (the call before this one found the server busy: its block was never posted)
		rt_push (e, nullptr, e->rt_skip_n);
		e->rt_skip_n = 0;
	}
	return PROT_OK;
}

} // namespace prot_host

extern "C" {

int prot_lv2_set_max_block (prot_engine* e, uint32_t frames)
{
	if (!e || e->info.kind != PROT_KIND_LV2 || frames == 0 || frames > (1u << 24)) {
		return PROT_EINVAL;
	}
	HIP_TRY (hipSetDevice (e->device));
	ServerHold srv_hold;
	(void)join_pipe (e, e->stream);
	HIP_TRY (sync_own (e));
	return lv2_reserve (e, frames);
}

// one call of at most lv2_max_block frames, `off` frames into the host's port buffers
static int lv2_run_chunk (prot_engine* e, const float* const* in, float* const* out, uint32_t off, uint32_t n_samples,
                          const float* angle_deg, bool wait_pending)
{
	const uint32_t nc = e->info.n_chn;
	const uint32_t P  = e->info.blksiz;
	const size_t   n  = n_samples;
	const bool     deferred = e->rt_frames > 0;
	if (deferred) {
		int rcd = rt_collect (e, wait_pending);
		if (rcd) {
			return rcd;
		}
		if (e->rt_pending) { // the block before this one is still with the device: this call's frames cannot be posted (counted)
			e->rt_skip_n = n_samples;
			e->rt_dropped++;
			rt_pop (e, out, off, n_samples);
			return PROT_OK;
		}
	}
	// (another engine's run() server stays: this call is itself a run(); this engine's own is stopped below if the call
	// does not go through it)
	CallStream    s (e, nullptr, /*keep_server*/ true);

	int rc;
	// Zero-copy (default): the kernels read the input, the per-frame rotation and write the output
	// straight from / to the pinned host block over the host link -- a run() call is two kernel
	// launches and one stream sync, no copy commands.  PROT_LV2_STAGED=1 keeps the staged route
	// (H2D copies -> device buffers -> D2H copy) for comparison.
	// Measured (tools/attic/run_latency.py): 64..1024-frame buffers 37..42 us zero-copy vs 46..47 us staged; at 8192 frames the
	// DMA copies win (70 vs 76 us: 64 KiB over the host link from inside a latency-bound kernel) -- staged above 4096 samples.
	const bool force_staged = test_opt (PROT_TEST_LV2_STAGED) != 0;
	const bool        staged       = force_staged || n * nc > 4096;
	// (buffers: lv2_reserve -- nothing is allocated on this path)
	float* h_in  = e->h_io.p;
	float* h_out = e->h_io.p + n * nc;
	float* h_rot = e->h_io.p + 2 * n * nc; // [nc][n][2]

	// block-rate angle state machine, exactly process_channel's
	// (src/phaserotate.c:563-571 target, :673-709 ramp, :710-717 steady)
	bool steady = true;
	auto target_of = [&] (uint32_t c) {
		float target_angle = angle_deg[c] / -360.f;
		return target_angle < -.5f ? -.5f : target_angle > 0.5f ? 0.5f : target_angle;
	};
	// the common call -- every channel at its target angle, no ramp in its queue -- needs none of the per-frame bookkeeping below:
	// the block position moves on, the rotation of the call is {ca, sa} (what the queue holds in every entry); 1-1.5 us of a 14 us call
	bool all_at_target = true;
	for (uint32_t c = 0; c < nc && all_at_target; ++c) {
		all_at_target = e->lv2[c].rotq_steady && target_of (c) == e->lv2[c].angle;
	}
	// (the input goes into the pinned block only for the routes that read it there: a served call's frames go straight into the mailbox)
	auto fill_h_in = [&] () {
		for (uint32_t c = 0; c < nc; ++c) {
			memcpy (h_in + c * n, in[c] + off, n * sizeof (float));
		}
	};
	for (uint32_t c = 0; c < nc; ++c) {
		Lv2Chn& ch = e->lv2[c];
		if (all_at_target) {
			ch.offset = (uint32_t)((ch.offset + (uint64_t)n_samples) % P);
			h_rot[2 * n * c]     = ch.ca;
			h_rot[2 * n * c + 1] = ch.sa;
			continue;
		}
		const float target_angle = target_of (c);
		float*   r      = h_rot + 2 * n * c;
		uint32_t remain = n_samples, done = 0;
		while (remain > 0) {
			uint32_t ns = std::min (P - ch.offset, remain);
			memcpy (r + 2 * done, &ch.rotq[2 * ch.offset], 2 * ns * sizeof (float));
			done += ns;
			remain -= ns;
			ch.offset += ns;
			if (ch.offset == P) {
				ch.offset   = 0;
				float angle = ch.angle;
				if (target_angle != angle) {
					float da = (target_angle - angle);
					if (fabs (da) > 0.5) {
						if (da < 0) {
							da += 1.f;
						} else {
							da -= 1.f;
						}
					}
					da *= e->interp_nm;
					int         final_ = 0;
					const float thresh = e->interp_th;
					if (da > thresh) {
						da = thresh;
					} else if (da < -thresh) {
						da = -thresh;
					} else {
						final_ = 1;
					}
					static const float twopi = 2 * M_PI; // sin_cos, src/phaserotate.c:122-133
					for (uint32_t i = 0; i < P; ++i) {
						float ca, sa;
						sincosf (angle * twopi, &sa, &ca);
						ch.rotq[2 * i]     = ca;
						ch.rotq[2 * i + 1] = sa;
						angle += da;
					}
					ch.rotq_steady = false;
					if (final_) {
						angle = target_angle;
					}
					if (angle == target_angle) {
						sincosf (angle * twopi, &ch.sa, &ch.ca);
					}
				} else {
					for (uint32_t i = 0; i < P; ++i) {
						ch.rotq[2 * i]     = ch.ca;
						ch.rotq[2 * i + 1] = ch.sa;
					}
					ch.rotq_steady = true;
				}
				ch.angle = angle;
			}
		}
		for (uint32_t i = 1; i < n_samples && steady; ++i) {
			steady = (r[2 * i] == r[0]) && (r[2 * i + 1] == r[1]);
		}
	}

	// The run() server (prot_olasave512.hip): the call's parameters and frames go into the mailbox, the doorbell is rung, the resident
	// kernel does the block and stores the call's number when the output is in h_out -- no launch, no stream synchronisation.
	// One server per process, a slot per channel of every engine on its roster (g_srv_roster); calls of more than one block per channel, staged
	// calls and PROT_TEST_LV2_NO_SERVER take the launch below.  tools/attic/run_latency.py: 38-40 us -> the figure in DESIGN.md.
	bool served = false;
	if (!staged && e->srv_eligible && test_opt (PROT_TEST_LV2_NO_SERVER) == 0) {
		std::lock_guard<std::mutex> lk (g_srv_mu);
		if (g_srv_hold == 0 && server_roster_takes (e)) {
			prot::SvServerBox* b = reinterpret_cast<prot::SvServerBox*> (e->srv_box);
			prot::SvServerAck* a = e->srv_ack.p;
			// the host -> device half of the mailbox is STORED into, never read (through the BAR a CPU load is microseconds): the
			// call's input frames behind the box, the per-frame rotation of a ramp behind them
			float* const srv_in  = reinterpret_cast<float*> (e->srv_box + prot::kServerInOffset);
			float* const srv_rot = srv_in + (size_t)e->lv2_max_block * nc;
			for (uint32_t c = 0; c < nc; ++c) {
				memcpy (srv_in + c * n, in[c] + off, n * sizeof (float));
			}
			if (!steady) {
				memcpy (srv_rot, h_rot, 2 * n * nc * sizeof (float));
			}
			const float* d_rot_c = nullptr;
			if (steady) {
				float* h_rotc = e->h_io.p + 4 * n * nc; // [nc][2]
				for (uint32_t c = 0; c < nc; ++c) {
					h_rotc[2 * c]     = h_rot[2 * n * c];
					h_rotc[2 * c + 1] = h_rot[2 * n * c + 1];
				}
				d_rot_c = h_rotc;
			}
			const int hist_before = e->hist_cur;
			// (steady rotation: by value in the parameters -- a pointer into the pinned block would be one more read over the host link)
			const bool rot_inline = steady && (int)nc <= prot::kRotInline;
			rc = ola_launch (e, srv_in, (int64_t)n, n, h_out, (int64_t)n, n, rot_inline ? nullptr : d_rot_c, steady ? nullptr : srv_rot, false, false, 0, (int)nc, s,
			                 /*advance_hist*/ true, rot_inline ? d_rot_c : nullptr, nullptr, nullptr, b);
			if (rc == PROT_OK) {
				if (!e->srv_running || !server_is_member (e) || __atomic_load_n (&a->alive, __ATOMIC_ACQUIRE) == 0u) {
					// (for the whole roster: the other engines' workgroups come back with this one's; a deferred call waits HERE until the
					// kernel polls -- 0.1-0.3 ms, after an idle time-out only -- or the next calls would find its block late)
					rc = deferred ? server_start_wait_locked (e) : server_start_locked (e);
				}
				if (rc == PROT_OK) {
					e->srv_seq = (e->srv_seq % (1u << 30)) + 1u;
					const uint32_t seq  = e->srv_seq;
					const uint64_t bell = (uint64_t)seq | ((uint64_t)n << 32);
					_mm_sfence (); // (write-combined stores through the BAR: parameters and frames before the doorbell)
					__atomic_store_n (&b->bell, bell, __ATOMIC_RELEASE);
					_mm_sfence ();
					if (deferred) { // the block is the server's now; the next call collects it (rt_collect)
						e->rt_pending    = true;
						e->rt_pend_ready = e->rt_discard = false;
						e->rt_pend_seq   = seq;
						e->rt_pend_n     = n_samples;
						e->rt_pend_out   = h_out;
					}
					// wait for every channel's done word; a server that left by itself between the check above and the doorbell
					// (idle time-out) is started again -- with the number of the last served call, so that it takes this one at once
					const auto t_0 = std::chrono::steady_clock::now ();
					for (uint32_t c = 0; c < nc && !deferred;) {
						if (__atomic_load_n (&a->done[c], __ATOMIC_ACQUIRE) == seq) {
							++c;
							continue;
						}
						if (__atomic_load_n (&a->alive, __ATOMIC_ACQUIRE) == 0u) {
							// this engine's workgroups have left (idle, or the kernel's age): the others' follow on the stop below -- what
							// they served is complete when the kernel is over
							rc = server_stop_locked ();
							if (rc) {
								return rc;
							}
							bool all = true;
							for (uint32_t k = 0; k < nc; ++k) {
								all = all && __atomic_load_n (&a->done[k], __ATOMIC_ACQUIRE) == seq;
							}
							if (all) {
								break;
							}
							e->srv_running = false;
							e->srv_seq     = seq - 1u; // (any number but seq: the new kernel takes the call at once)
							rc             = server_start_locked (e);
							e->srv_seq     = seq;
							if (rc) {
								return rc;
							}
							_mm_sfence ();
							__atomic_store_n (&b->bell, bell, __ATOMIC_RELEASE);
							_mm_sfence ();
							continue;
						}
						if (std::chrono::steady_clock::now () - t_0 > std::chrono::seconds (2)) {
							snprintf (g_err, sizeof (g_err), "run() server: call %u not served within 2 s (stages %u %u, done %u %u, alive %u)", seq, a->pad0[0],
							          a->pad0[1], a->done[0], a->done[1], a->alive);
							// (the engine's state as it was before the call, and no further call of this engine goes to a server that
							// does not answer: the launch route from here on -- ADVICE r4)
							e->hist_cur     = hist_before;
							e->srv_eligible = false;
							server_roster_remove (e);
							return PROT_EHIP;
						}
					}
					e->srv_calls++;
					served = true;
				}
			} else if (rc == PROT_EUNSUPPORTED) {
				(void)hist_before; // (ola_launch returns before it touches the engine's state when the server cannot take the call)
				rc = PROT_OK;
			}
			if (rc) {
				return rc;
			}
			if (!served) {
				(void)server_stop_locked (); // the launch below uses the engine's stream and history
			}
		}
	}
	// deferred mode, a block a blocking route has just computed: it waits in the pinned block like a posted one
	auto hand_back = [&] () {
		if (deferred) {
			if (!served) {
				e->rt_pending    = true;
				e->rt_pend_ready = true;
				e->rt_discard    = false;
				e->rt_pend_n     = n_samples;
				e->rt_pend_out   = h_out;
			}
			rt_pop (e, out, off, n_samples);
			return;
		}
		for (uint32_t c = 0; c < nc; ++c) {
			memcpy (out[c] + off, h_out + c * n, n * sizeof (float));
		}
	};
	if (served) {
		e->pos += n;
		hand_back ();
		return PROT_OK;
	}
	fill_h_in ();
	if (!staged) {
		const float* d_rot_c = nullptr;
		if (steady) {
			float* h_rotc = e->h_io.p + 4 * n * nc; // [nc][2]
			for (uint32_t c = 0; c < nc; ++c) {
				h_rotc[2 * c]     = h_rot[2 * n * c];
				h_rotc[2 * c + 1] = h_rot[2 * n * c + 1];
			}
			d_rot_c = h_rotc;
		}
		rc = ola_launch (e, h_in, (int64_t)n, n, h_out, (int64_t)n, n, d_rot_c, steady ? nullptr : h_rot, false, false, 0, (int)nc, s,
		                 /*advance_hist*/ true); // one kernel per run(): the OLA launch writes the next history as well
		if (rc) {
			return rc;
		}
		e->pos += n;
		HIP_TRY (hipStreamSynchronize (s)); // (polling hipStreamQuery instead measured the same: the runtime already spins)
		hand_back ();
		return PROT_OK;
	}
	{ // (the staged route uses the engine's stream and history: not beside this engine's own server)
		std::lock_guard<std::mutex> lk (g_srv_mu);
		if (server_is_member (e)) {
			(void)server_stop_locked ();
		}
	}
	HIP_TRY (hipMemcpyAsync (e->d_in.p, h_in, n * nc * sizeof (float), hipMemcpyHostToDevice, s));
	if (steady) {
		float rot[2 * 64];
		std::vector<float> rot_big;
		float* rp = rot;
		if (nc > 64) {
			rot_big.resize (2 * (size_t)nc);
			rp = rot_big.data ();
		}
		for (uint32_t c = 0; c < nc; ++c) {
			rp[2 * c]     = h_rot[2 * n * c];
			rp[2 * c + 1] = h_rot[2 * n * c + 1];
		}
		rc = prot_apply_dev (e, e->d_in.p, (int64_t)n, n, e->d_out.p, (int64_t)n, n, rp, 0, s);
	} else {
		HIP_TRY (hipMemcpyAsync (e->d_rot_s.p, h_rot, 2 * n * nc * sizeof (float), hipMemcpyHostToDevice, s));
		rc = prot_apply_dev_ramp (e, e->d_in.p, (int64_t)n, n, e->d_out.p, (int64_t)n, n, e->d_rot_s.p, 0, s);
	}
	if (rc) {
		return rc;
	}
	HIP_TRY (hipMemcpyAsync (h_out, e->d_out.p, n * nc * sizeof (float), hipMemcpyDeviceToHost, s));
	HIP_TRY (hipStreamSynchronize (s));
	hand_back ();
	return PROT_OK;
}

int prot_lv2_run (prot_engine* e, const float* const* in, float* const* out, uint32_t n_samples,
                  const float* angle_deg)
{
	if (!e || !in || !out || !angle_deg || e->info.kind != PROT_KIND_LV2) {
		return PROT_EINVAL;
	}
	if (n_samples == 0) {
		return PROT_OK;
	}
	HIP_TRY (hipSetDevice (e->device));
	// calls longer than the reserved block length are cut up: the callback never allocates (src/phaserotate.c run()
	// does not either).  In-place port buffers (in == out) are fine: a chunk's input is copied out before its output lands
	const uint32_t step = e->rt_frames > 0 ? std::min (e->lv2_max_block, e->rt_frames) : e->lv2_max_block;
	for (uint32_t off = 0; off < n_samples; off += step) {
		const int rc = lv2_run_chunk (e, in, out, off, std::min (step, n_samples - off), angle_deg, /*wait_pending*/ off > 0);
		if (rc) {
			return rc;
		}
	}
	return PROT_OK;
}

int prot_lv2_get_angle (prot_engine* e, uint32_t chn, float* angle_turns)
{
	if (!e || !angle_turns || e->info.kind != PROT_KIND_LV2 || chn >= e->lv2.size ()) {
		return PROT_EINVAL;
	}
	*angle_turns = e->lv2[chn].angle;
	return PROT_OK;
}

int prot_lv2_set_deferred (prot_engine* e, uint32_t frames)
{
	if (!e || e->info.kind != PROT_KIND_LV2 || frames > (1u << 20)) {
		return PROT_EINVAL;
	}
	if (frames > 0 && !e->srv_eligible) {
		snprintf (g_err, sizeof (g_err), "prot_lv2_set_deferred: this engine's run() does not go through the resident kernel");
		return PROT_EUNSUPPORTED;
	}
	HIP_TRY (hipSetDevice (e->device));
	{
	ServerHold srv_hold; // (the server leaves -- behind the block it may still hold -- and comes back without / with its age limit)
	(void)join_pipe (e, e->stream);
	HIP_TRY (sync_own (e));
	if (frames > 0) {
		const int rc = lv2_reserve (e, std::max (frames, e->lv2_max_block));
		if (rc) {
			return rc;
		}
		try {
			e->rt_ring.assign ((size_t)2 * frames * e->info.n_chn, 0.f);
		} catch (...) {
			return PROT_ENOMEM;
		}
	} else {
		std::vector<float> ().swap (e->rt_ring);
	}
	e->rt_frames = frames;
	e->rt_cap    = 2 * frames;
	rt_ring_reset (e);
	}
	return rt_prestart (e);
}

int prot_lv2_deferred_stats (prot_engine* e, uint64_t* late, uint64_t* dropped)
{
	if (!e || e->info.kind != PROT_KIND_LV2) {
		return PROT_EINVAL;
	}
	if (late) {
		*late = e->rt_late;
	}
	if (dropped) {
		*dropped = e->rt_dropped;
	}
	return PROT_OK;
}

} // extern "C"

