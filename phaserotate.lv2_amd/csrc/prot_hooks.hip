// prot_hooks.hip -- TEST HOOKS (csrc/prot_internal.h), compiled into libprot_hip_hooks.so only (-DPROT_TESTHOOKS): nothing here is in the shipped library
// (split from prot_capi.hip in round 6: same code; what the units share is in prot_engine.h)
#include "prot_engine.h"

extern "C" {

#ifdef PROT_TESTHOOKS
// TEST HOOK, not part of include/prot_hip.h (declared in csrc/prot_internal.h): the library that Rccl::load() opens in
// place of the system's RCCL -- an explicit call of the test process, never the environment (the library is also
// loaded into LV2 hosts).  Only before the first all-reduce; NULL / "" restores the default.
int prot_internal_fused_lists (prot_engine* e, unsigned long long out[4])
{
	if (!e || !out) {
		return PROT_EINVAL;
	}
	out[0] = (unsigned long long)(uintptr_t)e->d_surv.p;
	out[1] = (unsigned long long)(uintptr_t)e->d_surv_cnt.p;
	out[2] = (unsigned long long)e->last_n_lists;
	out[3] = (unsigned long long)e->last_cap;
	return PROT_OK;
}

int prot_internal_server_stats (prot_engine* e, unsigned long long out[2])
{
	if (!e || !out) {
		return PROT_EINVAL;
	}
	out[0] = e->srv_calls;
	out[1] = e->srv_starts;
	return PROT_OK;
}
#ifdef PROT_SERVER_STAMPS
// diagnostic builds (tools/ab_build.py): 100 MHz ticks of the last call's stages (doorbell seen, parameters read, block done, done stored)
int prot_internal_server_stamps (prot_engine* e, unsigned out[4])
{
	if (!e || !out || !e->srv_ack.p) {
		return PROT_EINVAL;
	}
	for (int i = 0; i < 4; ++i) {
		out[i] = e->srv_ack.p->pad0[8 + i];
	}
	return PROT_OK;
}
#endif

// What an LV2 host's audio thread sees: `calls` calls of prot_lv2_run on `frames` frames per channel straight from C (no binding in
// between), into us[3]: mean microseconds per call, the slowest single call, its index.  The input is a fixed pseudo-random block, the angle 90 degrees.
int prot_internal_time_lv2_run (prot_engine* e, uint32_t frames, uint32_t calls, double us[3])
{
	if (!e || !us || e->info.kind != PROT_KIND_LV2 || frames == 0 || calls == 0) {
		return PROT_EINVAL;
	}
	const uint32_t nc = e->info.n_chn;
	std::vector<float>        buf ((size_t)2 * nc * frames);
	std::vector<const float*> in (nc);
	std::vector<float*>       out (nc);
	std::vector<float>        ang (nc, 90.f);
	uint32_t                  x = 12345u;
	for (size_t i = 0; i < (size_t)nc * frames; ++i) {
		x ^= x << 13, x ^= x >> 17, x ^= x << 5;
		buf[i] = (float)(x >> 8) * (1.f / 16777216.f) - 0.5f;
	}
	for (uint32_t c = 0; c < nc; ++c) {
		in[c]  = buf.data () + (size_t)c * frames;
		out[c] = buf.data () + (size_t)(nc + c) * frames;
	}
	for (uint32_t k = 0; k < 50; ++k) {
		const int rc = prot_lv2_run (e, in.data (), out.data (), frames, ang.data ());
		if (rc) {
			return rc;
		}
	}
	const auto t_0 = std::chrono::steady_clock::now ();
	auto       t_p = t_0;
	double     worst = 0.0;
	uint32_t   worst_at = 0;
	for (uint32_t k = 0; k < calls; ++k) {
		const int rc = prot_lv2_run (e, in.data (), out.data (), frames, ang.data ());
		if (rc) {
			return rc;
		}
		const auto   t_n = std::chrono::steady_clock::now ();
		const double d   = std::chrono::duration<double, std::micro> (t_n - t_p).count ();
		if (d > worst) {
			worst    = d;
			worst_at = k;
		}
		t_p = t_n;
	}
	us[0] = std::chrono::duration<double, std::micro> (t_p - t_0).count () / calls;
	us[1] = worst;            // the slowest single call
	us[2] = (double)worst_at; // ... and which
	return PROT_OK;
}

int prot_internal_time_lv2_run_paced (prot_engine* e, uint32_t frames, uint32_t calls, uint32_t gap_us, double us[3])
{
	if (!e || !us || e->info.kind != PROT_KIND_LV2 || frames == 0 || calls == 0) {
		return PROT_EINVAL;
	}
	const uint32_t nc = e->info.n_chn;
	std::vector<float>        buf ((size_t)2 * nc * frames);
	std::vector<const float*> in (nc);
	std::vector<float*>       out (nc);
	std::vector<float>        ang (nc, 90.f);
	uint32_t                  x = 12345u;
	for (size_t i = 0; i < (size_t)nc * frames; ++i) {
		x ^= x << 13, x ^= x >> 17, x ^= x << 5;
		buf[i] = (float)(x >> 8) * (1.f / 16777216.f) - 0.5f;
	}
	for (uint32_t c = 0; c < nc; ++c) {
		in[c]  = buf.data () + (size_t)c * frames;
		out[c] = buf.data () + (size_t)(nc + c) * frames;
	}
	double   sum = 0.0, worst = 0.0;
	uint32_t worst_at = 0;
	for (uint32_t k = 0; k < 50 + calls; ++k) {
		const auto t_a = std::chrono::steady_clock::now ();
		const int  rc  = prot_lv2_run (e, in.data (), out.data (), frames, ang.data ());
		if (rc) {
			return rc;
		}
		const auto   t_b = std::chrono::steady_clock::now ();
		const double d   = std::chrono::duration<double, std::micro> (t_b - t_a).count ();
		if (k >= 50) {
			sum += d;
			if (d > worst) {
				worst    = d;
				worst_at = k - 50;
			}
		}
		while (std::chrono::duration<double, std::micro> (std::chrono::steady_clock::now () - t_b).count () < (double)gap_us) {
		}
	}
	us[0] = sum / calls;
	us[1] = worst;
	us[2] = (double)worst_at;
	return PROT_OK;
}

// The same for a host that runs several plugin instances in turn (round 5: one resident kernel serves them all): `calls` rounds of
// one prot_lv2_run on each of the n engines, into us[3]: mean microseconds per CALL (not per round), the slowest single call, its index.
int prot_internal_time_lv2_run_multi (prot_engine* const* es, uint32_t n, uint32_t frames, uint32_t calls, double us[3])
{
	if (!es || n == 0 || n > 16 || !us || frames == 0 || calls == 0) {
		return PROT_EINVAL;
	}
	uint32_t nc_max = 0;
	for (uint32_t i = 0; i < n; ++i) {
		if (!es[i] || es[i]->info.kind != PROT_KIND_LV2) {
			return PROT_EINVAL;
		}
		nc_max = std::max (nc_max, es[i]->info.n_chn);
	}
	std::vector<float>        buf ((size_t)2 * nc_max * frames);
	std::vector<const float*> in (nc_max);
	std::vector<float*>       out (nc_max);
	std::vector<float>        ang (nc_max, 90.f);
	uint32_t                  x = 12345u;
	for (size_t i = 0; i < (size_t)nc_max * frames; ++i) {
		x ^= x << 13, x ^= x >> 17, x ^= x << 5;
		buf[i] = (float)(x >> 8) * (1.f / 16777216.f) - 0.5f;
	}
	for (uint32_t c = 0; c < nc_max; ++c) {
		in[c]  = buf.data () + (size_t)c * frames;
		out[c] = buf.data () + (size_t)(nc_max + c) * frames;
	}
	double   worst = 0.0;
	uint32_t worst_at = 0;
	auto     t_0 = std::chrono::steady_clock::now ();
	for (uint32_t k = 0; k < 50 + calls; ++k) {
		if (k == 50) {
			t_0 = std::chrono::steady_clock::now ();
		}
		for (uint32_t i = 0; i < n; ++i) {
			const auto t_a = std::chrono::steady_clock::now ();
			const int  rc  = prot_lv2_run (es[i], in.data (), out.data (), frames, ang.data ());
			if (rc) {
				return rc;
			}
			const double d = std::chrono::duration<double, std::micro> (std::chrono::steady_clock::now () - t_a).count ();
			if (k >= 50 && d > worst) {
				worst    = d;
				worst_at = (k - 50) * n + i;
			}
		}
	}
	us[0] = std::chrono::duration<double, std::micro> (std::chrono::steady_clock::now () - t_0).count () / ((double)calls * n);
	us[1] = worst;
	us[2] = (double)worst_at;
	return PROT_OK;
}

int prot_internal_set_test_option (int option, int value)
{
	if (option < 0 || option >= PROT_TEST_OPTIONS) {
		return PROT_EINVAL;
	}
	g_test_opt_v[option].store (value);
	return PROT_OK;
}

int prot_internal_set_rccl_library (const char* path)
{
	return rccl_set_override (path);
}

#endif // PROT_TESTHOOKS

} // extern "C"

