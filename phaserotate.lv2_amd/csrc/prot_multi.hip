// prot_multi.hip -- several engines of one process: RCCL through dlopen, cached communicators, prot_allreduce_peaks, prot_cli_analyze_file_multi
// (split from prot_capi.hip in round 6: same code; what the units share is in prot_engine.h)
#include "prot_engine.h"

namespace prot_host {

struct Rccl {
	void* lib = nullptr;
	int (*CommInitAll) (void**, int, const int*) = nullptr;
	int (*CommDestroy) (void*)                    = nullptr;
	int (*GroupStart) ()                          = nullptr;
	int (*GroupEnd) ()                            = nullptr;
	int (*AllReduce) (const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
	std::string override_path;
	bool load ()
	{
		if (lib) {
			return true;
		}
		// `override_path` (prot_internal_set_rccl_library, tests only: a counting stand-in) replaces the system's RCCL
		const char* over = override_path.empty () ? nullptr : override_path.c_str ();
		for (const char* name : { over ? over : "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
			lib = dlopen (name, RTLD_NOW | RTLD_LOCAL);
			if (lib || over) {
				break;
			}
		}
		if (!lib) {
			return false;
		}
		CommInitAll = (int (*) (void**, int, const int*))dlsym (lib, "ncclCommInitAll");
		CommDestroy = (int (*) (void*))dlsym (lib, "ncclCommDestroy");
		GroupStart  = (int (*) ())dlsym (lib, "ncclGroupStart");
		GroupEnd    = (int (*) ())dlsym (lib, "ncclGroupEnd");
		AllReduce   = (int (*) (const void*, void*, size_t, int, int, void*, hipStream_t))dlsym (lib, "ncclAllReduce");
		if (!(CommInitAll && CommDestroy && GroupStart && GroupEnd && AllReduce)) {
			// not an RCCL: forget it, so that later calls fall back to the host merge instead of calling null pointers
			dlclose (lib);
			lib = nullptr;
			CommInitAll = nullptr, CommDestroy = nullptr, GroupStart = nullptr, GroupEnd = nullptr, AllReduce = nullptr;
			return false;
		}
		return true;
	}
};
Rccl       g_rccl;
std::mutex g_rccl_lock;
// communicators are expensive (hundreds of ms for ncclCommInitAll on 8 GPUs against ~2 ms per GPU for a sharded
// sweep, and the CLI all-reduces once per analysis pass): one set per list of devices, created on first use, kept
// until the last engine of the process is destroyed
std::map<std::vector<int>, std::vector<void*>> g_comms;
int                                            g_engines = 0; // live engines (guarded by g_rccl_lock)

void rccl_drop_comms_locked ()
{
	for (auto& kv : g_comms) {
		for (size_t i = 0; i < kv.second.size (); ++i) {
			if (kv.second[i]) {
				(void)hipSetDevice (kv.first[i]);
				g_rccl.CommDestroy (kv.second[i]);
			}
		}
	}
	g_comms.clear ();
}
int rccl_set_override (const char* path)
{
	std::lock_guard<std::mutex> lk (g_rccl_lock);
	if (g_rccl.lib) {
		return PROT_EINVAL;
	}
	g_rccl.override_path = path ? path : "";
	return PROT_OK;
}


// RCCL through dlopen: the library is only needed by multi-GPU callers
constexpr int kNcclFloat32 = 7, kNcclMax = 2; // rccl.h: ncclDataType_t / ncclRedOp_t

} // namespace prot_host

extern "C" {

int prot_allreduce_peaks (prot_engine* const* engines, int n)
{
	if (!engines || n < 1) {
		return PROT_EINVAL;
	}
	const size_t count = (size_t)PROT_MAXSAMPLE * engines[0]->info.n_chn;
	bool distinct = true;
	for (int i = 0; i < n; ++i) {
		if (!engines[i] || engines[i]->info.n_chn != engines[0]->info.n_chn) {
			return PROT_EINVAL;
		}
		engines[i]->h_peaks_valid = false;
		for (int k = 0; k < i; ++k) {
			distinct = distinct && engines[k]->device != engines[i]->device;
		}
	}
	// PROT_ALLREDUCE=host: merge through the host even on distinct devices; =rccl: take the collective even when the
	// engines share a device (only meaningful with a stand-in library: real RCCL wants one rank per device)
	const int   mode      = test_opt (PROT_TEST_ALLREDUCE); // (prot_internal.h: 1 = through the host, 2 = the collective even on one device,
	                                                        // 3 = ... and even for ONE engine: the real library on a one-GPU box)
	if (n == 1 && mode != 3) {
		return PROT_OK;
	}
	const bool  want_host = mode == 1;
	const bool  force     = mode == 2 || mode == 3;
	std::lock_guard<std::mutex> lk (g_rccl_lock);
	if (mode == 3 && !g_rccl.load ()) {
		snprintf (g_err, sizeof (g_err), "no RCCL library could be loaded");
		return PROT_EUNSUPPORTED;
	}
	if ((distinct || force) && !want_host && g_rccl.load ()) {
		std::vector<int> devs (n);
		for (int i = 0; i < n; ++i) {
			devs[i] = engines[i]->device;
		}
		auto it = g_comms.find (devs);
		if (it == g_comms.end ()) {
			std::vector<void*> comms (n, nullptr);
			if (g_rccl.CommInitAll (comms.data (), n, devs.data ()) != 0) {
				snprintf (g_err, sizeof (g_err), "ncclCommInitAll failed");
				return PROT_EHIP;
			}
			it = g_comms.emplace (devs, comms).first;
		}
		const std::vector<void*>& comms = it->second;
		// no early return between GroupStart and GroupEnd: an open group would swallow every later call
		int bad = g_rccl.GroupStart ();
		for (int i = 0; i < n && !bad; ++i) {
			bad = hipSetDevice (engines[i]->device) != hipSuccess;
			bad = bad || g_rccl.AllReduce (engines[i]->d_peaks.p, engines[i]->d_peaks.p, count, kNcclFloat32, kNcclMax, comms[i],
			                               engines[i]->stream);
		}
		bad = g_rccl.GroupEnd () || bad;
		for (int i = 0; i < n; ++i) {
			(void)hipSetDevice (engines[i]->device);
			bad = (hipStreamSynchronize (engines[i]->stream) != hipSuccess) || bad;
		}
		if (bad) {
			snprintf (g_err, sizeof (g_err), "ncclAllReduce failed");
			g_comms.erase (it); // do not reuse communicators after a failed collective (not destroyed: they may be wedged)
			return PROT_EHIP;
		}
		return PROT_OK;
	}
	// same-device engines (tests) or no RCCL: merge through the host
	std::vector<float> acc (count, 0.f), tmp (count);
	for (int i = 0; i < n; ++i) {
		int rc = prot_cli_peaks (engines[i], tmp.data ());
		if (rc) {
			return rc;
		}
		for (size_t k = 0; k < count; ++k) {
			acc[k] = std::max (acc[k], tmp[k]);
		}
	}
	for (int i = 0; i < n; ++i) {
		HIP_TRY (hipSetDevice (engines[i]->device));
		HIP_TRY (hipMemcpy (engines[i]->d_peaks.p, acc.data (), count * sizeof (float), hipMemcpyHostToDevice));
		engines[i]->h_peaks_valid = false;
	}
	return PROT_OK;
}

int prot_cli_analyze_file_multi (prot_engine* const* engines, int n, const float* data, uint64_t n_frames, int ang_start,
                                 int ang_end, int ang_stride, int chn)
{
	if (!engines || n < 1 || (!data && n_frames)) {
		return PROT_EINVAL;
	}
	for (int i = 0; i < n; ++i) {
		if (!engines[i] || engines[i]->info.kind != PROT_KIND_CLI || engines[i]->info.blksiz != engines[0]->info.blksiz ||
		    engines[i]->info.n_chn != engines[0]->info.n_chn) {
			return PROT_EINVAL;
		}
	}
	// contiguous block ranges on alignment-unit boundaries (whole blksiz blocks AND whole device
	// transform blocks, file_align_frames): a shard that starts there reproduces the single-engine
	// arithmetic bit for bit
	const uint64_t P     = engines[0]->info.blksiz;
	const uint64_t nb    = (n_frames + P - 1) / P;
	const uint64_t ab    = file_align_frames (engines[0]) / P; // blocks per alignment unit
	const uint64_t units = (nb + ab - 1) / ab;
	std::vector<uint64_t> b0 (n + 1, 0);
	for (int r = 0; r < n; ++r) {
		const uint64_t k = units / n + ((uint64_t)r < units % n ? 1 : 0);
		b0[r + 1]        = std::min (nb, b0[r] + ab * k);
	}
	std::vector<int>         rcs (n, PROT_OK);
	std::vector<std::thread> th;
	int                      last_nonempty = 0;
	for (int r = 0; r < n; ++r) {
		if (b0[r + 1] > b0[r]) {
			last_nonempty = r;
		}
	}
	// Two phases (round 5): every engine seeds the pruning bound from a sparse sample of ITS range's first chunk, the sampled tables
	// are merged (one more 2.9 KB MAX all-reduce), and the main passes list against the GLOBAL bound -- a shard cannot know that the
	// loud passage is somewhere else: on its own bound it lists twice the frames the whole file lists.  Whatever the sample raises
	// in a table is a peak of real frames of the file, so the merged table is a valid lower bound of the final one for every
	// engine, and the final merge is bit-identical to a single pass (max is exact and order-free).
	const int seed_blocks = std::max (8, 64 / n);
	for (int phase = (n > 1 ? 1 : 0); phase <= (n > 1 ? 2 : 0); ++phase) {
		th.clear ();
		for (int r = 0; r < n; ++r) {
			const uint64_t f0 = b0[r] * P;
			const uint64_t f1 = std::min (b0[r + 1] * P, n_frames);
			const bool     is_last = r == last_nonempty;
			if (f1 <= f0 && !(is_last && r == 0)) {
				continue; // more engines than block pairs
			}
			th.emplace_back ([=, &rcs] () {
				rcs[r] = analyze_range (engines[r], data, f0, f1 - f0, r == 0, is_last, ang_start, ang_end, ang_stride, chn, phase, seed_blocks);
			});
		}
		for (auto& t : th) {
			t.join ();
		}
		for (int r = 0; r < n; ++r) {
			if (rcs[r]) {
				return rcs[r];
			}
		}
		const int rc = prot_allreduce_peaks (engines, n);
		if (rc) {
			return rc;
		}
	}
	return PROT_OK;
}

} // extern "C"

