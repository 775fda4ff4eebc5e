// prot_engine.hip -- errors, SinCosLut, FIR design, per-engine tables; create / destroy / reset / sync / info; streams of a call
// (split from prot_capi.hip in round 6: same code; what the units share is in prot_engine.h)
#include "prot_engine.h"

namespace prot_host {

#ifdef PROT_TESTHOOKS
std::atomic<int> g_test_opt_v[PROT_TEST_OPTIONS];
#endif
thread_local char g_err[256] = "";

int hip_fail (hipError_t e, const char* what)
{
	snprintf (g_err, sizeof (g_err), "%s: %s", what, hipGetErrorString (e));
	return e == hipErrorOutOfMemory ? PROT_ENOMEM : PROT_EHIP;
}


// ---- SinCosLut (cli/phase-rotate.cc:41-74) --------------------------------
float          g_lut[PROT_MAXSAMPLE * 2]; // {sin, cos}
std::once_flag g_lut_once;

void lut_init ()
{
	float mp = 2.f * M_PI / PROT_SUBSAMPLE / -360.0; // :46
	for (int i = 0; i < PROT_MAXSAMPLE; ++i) {
		sincosf (mp * i, &g_lut[2 * i], &g_lut[2 * i + 1]); // :52
	}
}

// ---- FIR design -------------------------------------------------------------
// Closed form of "spectrum j(-1)^k -> c2r_N -> x (0.5/N)(1 - cos 2 pi n/N)"
// (src/phaserotate.c:374-391, cli/phase-rotate.cc:144-161), in double, rounded
// to float once (the reference stores the FIR as float too).
void fir_design (int n, std::vector<float>& fir)
{
	fir.assign (n, 0.f);
	for (int i = 0; i < n; ++i) {
		const int m = i - n / 2;
		if (m & 1) {
			const double a = M_PI * (double)m / (double)n;
			fir[i] = (float)(-(2.0 / n) * (std::cos (a) / std::sin (a)) * 0.5 *
			                 (1.0 - std::cos (2.0 * M_PI * (double)i / (double)n)));
		}
	}
}

// in-place complex FFT, double, radix-2 (host, init-time only)
void fft_f64 (std::vector<double>& re, std::vector<double>& im)
{
	const size_t n = re.size ();
	for (size_t i = 1, j = 0; i < n; ++i) {
		size_t bit = n >> 1;
		for (; j & bit; bit >>= 1) {
			j ^= bit;
		}
		j ^= bit;
		if (i < j) {
			std::swap (re[i], re[j]);
			std::swap (im[i], im[j]);
		}
	}
	for (size_t len = 2; len <= n; len <<= 1) {
		const size_t half = len / 2;
		for (size_t k = 0; k < half; ++k) {
			const double a = -2.0 * M_PI * (double)k / (double)len;
			const double wr = std::cos (a), wi = std::sin (a);
			for (size_t i = k; i < n; i += len) {
				const size_t j  = i + half;
				const double tr = re[j] * wr - im[j] * wi;
				const double ti = re[j] * wi + im[j] * wr;
				re[j] = re[i] - tr;
				im[j] = im[i] - ti;
				re[i] += tr;
				im[i] += ti;
			}
		}
	}
}

// Constants of olasave_kernel (prot_olasave.hip) and of the 512-thread block (prot_olasave512.hip) for ONE partition of odd taps g (g[s] = fir[2 s + 1], Lg of them, a multiple
// of 512): H[k] = sum g[s] W^{sk} = -i B[k] W^{(Lg/2) k} W^{-k/2} with B real IF g is antisymmetric about its centre.
// Returns false (the engine keeps olapoly_kernel and its spectrum table) unless g is antisymmetric in its float values and
// rho = -B is flat to 1.5e-7 outside `rows` x 256 bins on either side of DC for rows = 1 or 2.
// Layout of `tab` (floats): f2[12][256] split outer twiddles of thread t (forward at the half-shifted index: W^{(t-1/2) i},
// W^{(t-1/2) 4 i}; inverse: W^{t i}, W^{4 t i}; i = 1..3), f2[256] W256^{(c-1/2) k2} at [16 k2 + c], f2[rows][256]
// (rho[q + 256 i], rho[q + 256 (15 - i)]) / mean of thread t, q = (t >> 4) + 16 (t & 15).  gain = mean rho / F.
bool save_tables (const std::vector<float>& fir, uint32_t F, std::vector<float>& tab, float& gain, int& rows)
{
	const size_t Lg = fir.size () / 2;
	if (F != 4096 || Lg % 512 != 0 || Lg < 512 || Lg > 2048) {
		return false;
	}
	std::vector<double> re (F, 0.0), im (F, 0.0);
	for (size_t s = 0; s < Lg; ++s) {
		if (fir[2 * s + 1] != -fir[2 * (Lg - 1 - s) + 1] || fir[2 * s] != 0.f) {
			return false;
		}
		re[s] = fir[2 * s + 1];
	}
	fft_f64 (re, im);
	std::vector<double> rho (F);
	for (uint32_t k = 0; k < F; ++k) {
		// conj of P[k] = -i e^{-2 pi i (Lg/2) k / F} e^{+2 pi i k / (2F)}
		const double a  = 2.0 * M_PI * ((double)(Lg / 2) * k / F - (double)k / (2.0 * F));
		const double pr = -std::sin (a), pi = -std::cos (a); // P = -i e^{-ia} = (-sin a, -cos a)
		const double br = re[k] * pr + im[k] * pi, bi = im[k] * pr - re[k] * pi; // H conj(P)
		if (std::fabs (bi) > 1e-9) {
			return false;
		}
		rho[k] = -br;
	}
	for (rows = 1; rows <= 2; ++rows) {
		double mean = 0;
		const uint32_t k0 = 256 * rows, k1 = F - 256 * rows;
		for (uint32_t k = k0; k < k1; ++k) {
			mean += rho[k];
		}
		mean /= (double)(k1 - k0);
		double dev = 0;
		for (uint32_t k = k0; k < k1; ++k) {
			dev = std::max (dev, std::fabs (rho[k] / mean - 1.0));
		}
		if (dev > 1.5e-7) {
			continue;
		}
		gain = (float)(mean / F);
		tab.assign (2 * (size_t)15 * 256, 0.f); // rows 0..11 twiddles, 12 W256 half-shifted, 13 rho, 14 W256 plain
		auto put = [&] (size_t idx, double ang) {
			tab[2 * idx]     = (float)std::cos (ang);
			tab[2 * idx + 1] = (float)std::sin (ang);
		};
		for (int t = 0; t < 256; ++t) {
			for (int i = 1; i <= 3; ++i) {
				put ((size_t)(i - 1) * 256 + t, -2.0 * M_PI * (t - 0.5) * i / F);
				put ((size_t)(3 + i - 1) * 256 + t, -2.0 * M_PI * (t - 0.5) * 4 * i / F);
				put ((size_t)(6 + i - 1) * 256 + t, -2.0 * M_PI * (double)t * i / F);
				put ((size_t)(9 + i - 1) * 256 + t, -2.0 * M_PI * (double)t * 4 * i / F);
			}
			put ((size_t)12 * 256 + t, -2.0 * M_PI * ((t & 15) - 0.5) * (t >> 4) / 256.0);
			put ((size_t)14 * 256 + t, -2.0 * M_PI * (double)(((t & 15) * (t >> 4)) & 255) / 256.0);
			const int q = (t >> 4) + 16 * (t & 15);
			for (int i = 0; i < 1; ++i) { // (rows == 2 engines keep olapoly_kernel)
				tab[2 * ((size_t)(13 + i) * 256 + t)]     = (float)(rho[q + 256 * i] / mean);
				tab[2 * ((size_t)(13 + i) * 256 + t) + 1] = (float)(rho[q + 256 * (15 - i)] / mean);
			}
		}
		if (rows == 1) {
			// the 512-thread block's constants behind them (prot_kernels.h kSv5*; tools/model/olasave512_model.py tables())
			tab.resize (2 * (size_t)(prot::kSv5Offset + prot::kSv5Size), 0.f);
			auto put5 = [&] (size_t idx, double ang) { put ((size_t)prot::kSv5Offset + idx, ang); };
			for (int j = 1; j <= 7; ++j) {
				for (int t = 0; t < 512; ++t) {
					put5 ((size_t)prot::kSv5fA + 512 * (j - 1) + t, -2.0 * M_PI * (t - 0.5) * j / 4096.0);
					put5 ((size_t)prot::kSv5iA + 512 * (j - 1) + t, +2.0 * M_PI * (double)((t * j) & 4095) / 4096.0);
				}
				for (int t = 0; t < 64; ++t) {
					put5 ((size_t)prot::kSv5fB + 64 * (j - 1) + t, -2.0 * M_PI * (t - 0.5) * j / 512.0);
					put5 ((size_t)prot::kSv5iB + 64 * (j - 1) + t, +2.0 * M_PI * (double)((t * j) & 511) / 512.0);
				}
				for (int t = 0; t < 8; ++t) {
					put5 ((size_t)prot::kSv5fC + 8 * (j - 1) + t, -2.0 * M_PI * (t - 0.5) * j / 64.0);
					put5 ((size_t)prot::kSv5iC + 8 * (j - 1) + t, +2.0 * M_PI * (double)((t * j) & 63) / 64.0);
				}
			}
			for (int t = 0; t < 512; ++t) {
				const int k1 = t >> 6, k2 = (t >> 3) & 7, k3 = t & 7, kk = k1 + 8 * k2 + 64 * k3;
				const size_t i5 = 2 * ((size_t)prot::kSv5Offset + prot::kSv5Rho + t);
				tab[i5]     = k3 < 4 ? (float)(rho[kk] / mean) : 1.f;
				tab[i5 + 1] = k3 >= 4 ? (float)(rho[kk + 3584] / mean) : 1.f;
			}
		}
		return true;
	}
	return false;
}

// The same for the 8192-point block (prot_olasave8k.hip; tools/gen/gen_olasave8k.py device_table): five rows of 512 float2 --
// W_N^{(T - 1/2)}, W_N^{4 (T - 1/2)} (N = 8192), the half-shifted and the plain table of the middle passes ([ke][k2][c] and
// its transpose: W512^{(c - 1/2) ke} W256^{(c - 1/2) k2}, W512^{c ke} W256^{c k2}), the |B| factors of the 512 bins either side
// of DC per thread -- and gain = mean |B| / 8192.  firlen = 1024 NT, NT in {3, 4}.
bool save_tables8k (const std::vector<float>& fir, std::vector<float>& tab, float& gain)
{
	const uint32_t F  = 8192;
	const size_t   Lg = fir.size () / 2;
	if (Lg % 512 != 0 || (Lg != 1536 && Lg != 2048 && Lg != 4096)) { // 3 / 4 / 8 context rows of 512 complex samples
		return false;
	}
	std::vector<double> re (F, 0.0), im (F, 0.0);
	for (size_t s = 0; s < Lg; ++s) {
		if (fir[2 * s + 1] != -fir[2 * (Lg - 1 - s) + 1] || fir[2 * s] != 0.f) {
			return false;
		}
		re[s] = fir[2 * s + 1];
	}
	fft_f64 (re, im);
	std::vector<double> rho (F);
	for (uint32_t k = 0; k < F; ++k) {
		const double a  = 2.0 * M_PI * ((double)(Lg / 2) * k / F - (double)k / (2.0 * F));
		const double pr = -std::sin (a), pi = -std::cos (a);
		const double br = re[k] * pr + im[k] * pi, bi = im[k] * pr - re[k] * pi;
		if (std::fabs (bi) > 1e-9) {
			return false;
		}
		rho[k] = -br;
	}
	double mean = 0;
	for (uint32_t k = 512; k < F - 512; ++k) {
		mean += rho[k];
	}
	mean /= (double)(F - 1024);
	for (uint32_t k = 512; k < F - 512; ++k) {
		if (std::fabs (rho[k] / mean - 1.0) > 1.5e-7) {
			return false;
		}
	}
	gain = (float)(mean / F);
	tab.assign (2 * (size_t)5 * 512, 0.f);
	auto put = [&] (size_t idx, double ang) {
		tab[2 * idx]     = (float)std::cos (ang);
		tab[2 * idx + 1] = (float)std::sin (ang);
	};
	for (int T = 0; T < 512; ++T) {
		put ((size_t)T, -2.0 * M_PI * (T - 0.5) / F);
		put ((size_t)512 + T, -2.0 * M_PI * (T - 0.5) * 4.0 / F);
		const int ke = T >> 8, hi = (T >> 4) & 15, lo = T & 15;
		put ((size_t)2 * 512 + T, -2.0 * M_PI * ((lo - 0.5) * ke / 512.0 + (lo - 0.5) * hi / 256.0)); // [ke][k2 = hi][c = lo]
		put ((size_t)3 * 512 + T, -2.0 * M_PI * ((double)hi * ke / 512.0 + (double)(hi * lo) / 256.0)); // [ke][c = hi][k2 = lo]
		const int q = hi + 16 * ke + 32 * lo; // pass-3 layout: k1 = hi, k2 = lo
		tab[2 * ((size_t)4 * 512 + T)]     = (float)(rho[q] / mean);
		tab[2 * ((size_t)4 * 512 + T) + 1] = (float)(rho[q + 7680] / mean);
	}
	return true;
}

// The 16384-point block's constants (prot_olasave16k.hip, tools/gen/gen_olasave16k.py: device_table): five rows of 1024 float2 --
// W_N^{(T - 1/2)}, W_N^{4 (T - 1/2)} (N = 16384), the half-shifted and the plain table of the middle passes ([ke][k2][c] and its
// transpose: W1024^{(c - 1/2) ke} W256^{(c - 1/2) k2}, W1024^{k2 ke} W256^{c k2}), the |B| factors of the 1024 bins either side of
// DC per thread -- and gain = mean |B| / 16384.  firlen = 16384 (8192 odd taps = 8 context rows).
bool save_tables16k (const std::vector<float>& fir, std::vector<float>& tab, float& gain)
{
	const uint32_t F  = 16384;
	const size_t   Lg = fir.size () / 2;
	if (Lg != 8192 && Lg != 4096) { // 8 / 4 context rows of 1024 complex samples
		return false;
	}
	std::vector<double> re (F, 0.0), im (F, 0.0);
	for (size_t s = 0; s < Lg; ++s) {
		if (fir[2 * s + 1] != -fir[2 * (Lg - 1 - s) + 1] || fir[2 * s] != 0.f) {
			return false;
		}
		re[s] = fir[2 * s + 1];
	}
	fft_f64 (re, im);
	std::vector<double> rho (F);
	for (uint32_t k = 0; k < F; ++k) {
		const double a  = 2.0 * M_PI * ((double)(Lg / 2) * k / F - (double)k / (2.0 * F));
		const double pr = -std::sin (a), pi = -std::cos (a);
		const double br = re[k] * pr + im[k] * pi, bi = im[k] * pr - re[k] * pi;
		if (std::fabs (bi) > 1e-9) {
			return false;
		}
		rho[k] = -br;
	}
	double mean = 0;
	for (uint32_t k = 1024; k < F - 1024; ++k) {
		mean += rho[k];
	}
	mean /= (double)(F - 2048);
	for (uint32_t k = 1024; k < F - 1024; ++k) {
		if (std::fabs (rho[k] / mean - 1.0) > 1.5e-7) {
			return false;
		}
	}
	gain = (float)(mean / F);
	tab.assign (2 * (size_t)5 * 1024, 0.f);
	auto put = [&] (size_t idx, double ang) {
		tab[2 * idx]     = (float)std::cos (ang);
		tab[2 * idx + 1] = (float)std::sin (ang);
	};
	for (int T = 0; T < 1024; ++T) {
		put ((size_t)T, -2.0 * M_PI * (T - 0.5) / F);
		put ((size_t)1024 + T, -2.0 * M_PI * (T - 0.5) * 4.0 / F);
		const int ke = T >> 8, hi = (T >> 4) & 15, lo = T & 15;
		put ((size_t)2 * 1024 + T, -2.0 * M_PI * ((lo - 0.5) * ke / 1024.0 + (lo - 0.5) * hi / 256.0)); // [ke][k2 = hi][c = lo]
		put ((size_t)3 * 1024 + T, -2.0 * M_PI * ((double)hi * ke / 1024.0 + (double)(hi * lo) / 256.0)); // [ke][c = hi][k2 = lo]
		const int q = hi + 16 * ke + 64 * lo; // pass-3 layout: k1 = hi, k2 = lo
		tab[2 * ((size_t)4 * 1024 + T)]     = (float)(rho[q] / mean);
		tab[2 * ((size_t)4 * 1024 + T) + 1] = (float)(rho[q + 15360] / mean);
	}
	return true;
}

int zero_state (prot_engine* e, hipStream_t zs)
{
	zs = zs ? zs : e->stream;
	// one launch: the history buffer that becomes current (the others of the ring are written in full before they are
	// read), the peak table, the sweep statistics, the half-precision history that becomes current
	const size_t hn = (size_t)e->info.history * e->info.n_chn;
	prot::ZeroRegions z {};
	z.p[0] = reinterpret_cast<uint32_t*> (e->d_hist[0].p);
	z.n[0] = hn;
	z.p[1] = reinterpret_cast<uint32_t*> (e->d_peaks.p);
	z.n[1] = (size_t)PROT_MAXSAMPLE * e->info.n_chn;
	z.p[2] = reinterpret_cast<uint32_t*> (e->d_stats.p);
	z.n[2] = 4 * sizeof (unsigned long long) / sizeof (uint32_t);
	z.p[3] = reinterpret_cast<uint32_t*> (e->d_hist16[0].p);
	z.n[3] = e->d_hist16[0].p ? (hn + 1) / 2 : 0;
	// what a sweep launch shares (bounds, step counters): its merge kernel leaves them zero, a launch that failed or was
	// aborted between the two does not -- a reset restores the known-good state
	z.p[4] = reinterpret_cast<uint32_t*> (e->d_thr.p);
	z.n[4] = e->d_thr.p ? prot::kSweepThrFloats * (size_t)e->info.n_chn : 0;
	z.p[5] = reinterpret_cast<uint32_t*> (e->d_ctr.p);
	z.n[5] = e->d_ctr.p ? (size_t)e->info.n_chn * 32 * 32 : 0;
	hipError_t err = prot::launch_zero_regions (z, zs);
	if (err != hipSuccess) {
		return hip_fail (err, "state reset launch");
	}
	if (e->d_fused_surv.p) {
		(void)hipMemsetAsync (e->d_fused_surv.p, 0, sizeof (unsigned long long), zs);
	}
	e->fused_frames  = 0;
	e->sweep_seeded  = false;
	e->hist16_cur    = 0;
	e->hist_cur      = 0;
	e->pos           = 0;
	e->h_peaks_valid = false;
	return PROT_OK;
}

uint64_t gcd_u64 (uint64_t a, uint64_t b)
{
	while (b) {
		const uint64_t t = a % b;
		a                = b;
		b                = t;
	}
	return a;
}

// frames: CLI files are cut (into device chunks, into shards for several engines) only at multiples of this,
// so that every piece starts on a blksiz block AND on a device transform block: bit-identical to one pass
uint64_t file_align_frames (const prot_engine* e)
{
	const uint64_t P = e->info.blksiz, A = std::max<uint32_t> (1, e->info.run_align);
	return P / gcd_u64 (P, A) * A;
}

// whole alignment units, <= 2^22 frames per channel (<= 16 MiB per channel per staging buffer)
uint64_t file_chunk_frames (const prot_engine* e)
{
	const uint64_t L = file_align_frames (e);
	return std::max<uint64_t> (L, ((uint64_t)1 << 22) / L * L);
}

// ev_last on the stream of the last call when that was a caller's: whoever waits for ev_last comes behind that call.  The handle
// is the caller's to keep alive until the engine has let go of it (prot_hip.h, "streams"): asking the runtime whether a stream
// still exists is no way out -- hipStreamQuery on a destroyed stream crashes in this runtime (tried, round 5).
bool mark_last (prot_engine* e)
{
	if (!e->last_foreign || e->last_stream_id == nullptr || e->ev_last == nullptr) {
		return false;
	}
	return hipEventRecord (e->ev_last, (hipStream_t)const_cast<void*> (e->last_stream_id)) == hipSuccess;
}

hipStream_t pick_stream (prot_engine* e, void* stream, bool keep_server)
{
	if (!keep_server) {
		(void)server_stop_any ();
	}
	hipStream_t s = stream ? (hipStream_t)stream : e->stream;
	(void)join_pipe (e, s);
	if (e->last_foreign && (const void*)s != e->last_stream_id) {
		// the call before this one ran on a stream of the caller's: this one reads the state it left (history, tables)
		if (mark_last (e)) {
			(void)hipStreamWaitEvent (s, e->ev_last, 0);
		}
	}
	if (s != e->stream && e->own_dirty) {
		// an earlier call ran on the engine's own stream and nobody has waited for it since: this one, on a stream of the caller's,
		// reads the state it left
		if (e->ev_own && hipEventRecord (e->ev_own, e->stream) == hipSuccess) {
			(void)hipStreamWaitEvent (s, e->ev_own, 0);
		} else {
			(void)hipStreamSynchronize (e->stream);
		}
		e->own_dirty = false;
	}
	if (e->reset_pending) {
		// prot_engine_reset behind a call on a caller's stream does not block the host: its zeroing sits on the engine's stream behind
		// that call's event; a call that arrives on another stream is put behind it here
		if (s != e->reset_stream) {
			(void)hipStreamWaitEvent (s, e->ev_reset, 0);
		}
		e->reset_pending = false;
	}
	return s;
}
// the engine's own stream AND a reset that went to a caller's stream: what the readers of device state wait for
hipError_t sync_own (prot_engine* e)
{
	hipError_t err = hipStreamSynchronize (e->stream);
	if (err == hipSuccess) {
		e->own_dirty = false;
	}
	if (err == hipSuccess && e->reset_pending && e->reset_stream != e->stream) {
		err = hipEventSynchronize (e->ev_reset);
	}
	return err;
}

// copy a small host array to device memory in stream order without keeping a reference to `src`
int stage_small (prot_engine* e, void* dst_dev, const void* src, size_t bytes, hipStream_t s)
{
	if (bytes > (size_t)prot_engine::kSlotBytes) { // many channels: rare, take the synchronous route
		HIP_TRY (hipStreamSynchronize (s));
		HIP_TRY (hipMemcpy (dst_dev, src, bytes, hipMemcpyHostToDevice));
		return PROT_OK;
	}
	const int i = e->slot_next;
	e->slot_next = (i + 1) % prot_engine::kSlots;
	HIP_TRY (hipEventSynchronize (e->slot_ev[i])); // no-op unless 64 copies are still in flight
	char* slot = e->h_slots.p + (size_t)i * prot_engine::kSlotBytes;
	memcpy (slot, src, bytes);
	HIP_TRY (hipMemcpyAsync (dst_dev, slot, bytes, hipMemcpyHostToDevice, s));
	HIP_TRY (hipEventRecord (e->slot_ev[i], s));
	return PROT_OK;
}

} // namespace prot_host

extern "C" {

int prot_abi_version (void)
{
	return PROT_ABI_VERSION;
}

const char* prot_strerror (int status)
{
	switch (status) {
		case PROT_OK: return "ok";
		case PROT_EINVAL: return "invalid argument";
		case PROT_ENODEV: return "no usable gfx950 device / HIP runtime";
		case PROT_ENOMEM: return "out of memory";
		case PROT_EHIP: return "HIP call failed";
		case PROT_EUNSUPPORTED: return "configuration not supported by this build";
		default: return "unknown status";
	}
}

const char* prot_last_error (void)
{
	return g_err;
}

void prot_sincos_lut (int a, float* s, float* c)
{
	std::call_once (g_lut_once, lut_init);
	a  = ((a % PROT_MAXSAMPLE) + PROT_MAXSAMPLE) % PROT_MAXSAMPLE;
	*s = g_lut[2 * a];
	*c = g_lut[2 * a + 1];
}

// What prot_engine_create works out on the host before it touches the device (plan_engine), and the tables the block kernels it
// picks need (pick_block_kernels); upload_engine allocates and uploads, setup_lv2 reserves run()'s buffers and the server's mailbox.
struct EnginePlan {
	prot_info           inf {};
	std::vector<float>  fir_taps;
	uint32_t            F = 4096, n_part = 0, part_len = 0;
	int                 hopseg = 0;
	bool                use_p2 = false, use_poly = false;
	std::vector<float2> H_h, Hp_h, tw_h;
	std::vector<float>  sv_h, sv8_h, sv16_h;
};

// kind + parameter -> FIR, form (direct / polyphase), partitions, hop, history: host arithmetic only
static int plan_engine (const prot_config* cfg, EnginePlan& pl)
{
	prot_info&          inf      = pl.inf;
	std::vector<float>& fir_taps = pl.fir_taps;

	inf.kind  = cfg->kind;
	inf.n_chn = cfg->n_chn;
	if (cfg->kind == PROT_KIND_CLI) {
		const uint32_t P = (uint32_t)cfg->param;
		if (P < 1024 || P > 32768 || (P & (P - 1)) || (double)P != cfg->param) { // cli/phase-rotate.cc:673, :753-755
			return PROT_EINVAL;
		}
		inf.blksiz     = P;
		inf.firlen     = P;      // cli/phase-rotate.cc:154 (c2r of _parsiz points)
		inf.dry_delay  = P / 2;  // :220  fin = &in[_firlen]
		inf.conv_delay = 0;
		inf.latency    = P / 2;  // :963
	} else if (cfg->kind == PROT_KIND_LV2) {
		const double rate = cfg->param;
		if (!(rate > 0)) {
			return PROT_EINVAL;
		}
		uint32_t fftlen, firlen; // src/phaserotate.c:278-289
		if (rate < 64000) {
			fftlen = 512;
			firlen = 3072;
		} else if (rate < 128000) {
			fftlen = 1024;
			firlen = 4096;
		} else {
			fftlen = 2048;
			firlen = 8192;
		}
		inf.blksiz     = fftlen / 2;                // parsiz :292
		inf.firlen     = firlen;
		inf.conv_delay = inf.blksiz;                // block buffering latency
		inf.latency    = inf.blksiz + firlen / 2;   // :297
		inf.dry_delay  = inf.latency;
	} else {
		return PROT_EINVAL;
	}
	// device OLA geometry.  One 4096-point FFT per hop pair; FIRs of up to 3072 taps
	// fit one transform (hop = 4096 - 3072 = 1024, or 2048 for <= 2049 taps).  Longer
	// FIRs (CLI blksiz >= 4096, LV2 >= 64 kHz) are split into 2048-tap partitions, each
	// convolved with hop 2048 and delayed by 2048 frames more than the one before; the
	// partial Hilbert branches add up in the output (uniformly partitioned OLA, the
	// scheme the reference's own LV2 engine uses with 256..1024-frame partitions,
	// src/phaserotate.c:640-655).
	const uint32_t F = pl.F;
	if (cfg->fftlen != 0 && cfg->fftlen != F) {
		snprintf (g_err, sizeof (g_err), "fftlen %u not supported; the device transform is 4096 points", cfg->fftlen);
		return PROT_EUNSUPPORTED;
	}
	const uint32_t seg = F / 16;
	// FIRs longer than one 2048-tap partition: partitions are convolved in PAIRS that share a
	// forward transform (ola4096p2_kernel); pair s covers taps [4096 s, 4096 s + 4096) and is
	// delayed by 4096 s frames; pairs after the first accumulate into the output.
	const bool no_p2  = tuning_env ("PROT_NO_P2") != nullptr || (cfg->flags & PROT_CFG_SINGLE_FFT_PER_HOP);
	bool       use_p2 = !no_p2 && inf.firlen > F / 2 + 1;
	// ... and, the Hilbert FIR being zero at every even tap (checked below on the designed taps),
	// in POLYPHASE form: the stream read as complex samples x[2j-1] + i x[2j] is convolved with
	// g[s] = fir[2 s + 1] (olapoly_kernel) -- half the taps, blocks of 4096 frames, one launch per
	// PAIR of 2048-tap partitions of g.  PROT_CFG_DIRECT_FORM keeps the two-partition direct form.
	fir_design ((int)inf.firlen, fir_taps);
	// Which form is the default is decided by measurement (tools/attic/order_check.py, 10 interleaved rounds on
	// two boxes, ranges disjoint): polyphase wins wherever the odd taps leave room for a longer hop --
	// 3072 taps (LV2 48 kHz, 5120 frames per block) -9 / -10 %, 2048 taps (CLI 2048, 6144) -9 / -11 %,
	// 1024 taps (CLI 1024, 7168) -21 / -24 % -- and from 8192 taps up, where it halves the launches
	// (1.31 -> 0.40 ms, 5.95 -> 2.63 ms per 2^27 frames at 8192 / 32768 taps); at 4096 taps (LV2 96 kHz,
	// CLI 4096: hop 4096 either way) the direct two-partition form was 6-7 % faster in round 1, at 2 workgroups per CU.
	// PROT_CFG_POLYPHASE / PROT_CFG_DIRECT_FORM force one or the other (cross-checks).
	const bool long_hop    = inf.firlen >= 1024 && inf.firlen / 2 <= 1537; // g leaves >= 2 extra rows of 256 samples
	// (round 2: with three workgroups per CU the polyphase kernel also wins at 4096 taps, 0.33 against 0.43 ms for the
	// two-partition direct form at LV2 96 kHz -- it is the default for every Hilbert FIR now)
	const bool poly_wanted = true;
	bool use_poly = (use_p2 || long_hop || (cfg->flags & PROT_CFG_POLYPHASE)) && poly_wanted && tuning_env ("PROT_NO_POLY") == nullptr &&
	                !(cfg->flags & (PROT_CFG_DIRECT_FORM | PROT_CFG_SINGLE_FFT_PER_HOP)) && inf.firlen % 4 == 0 &&
	                inf.firlen >= 1024 && inf.conv_delay % 2 == 0;
	for (uint32_t i = 0; use_poly && i < inf.firlen; i += 2) {
		use_poly = fir_taps[i] == 0.f;
	}
	int&      hopseg = pl.hopseg;
	uint32_t& n_part = pl.n_part;
	uint32_t& part_len = pl.part_len;
	if (use_poly) {
		part_len = F / 2;                                         // taps of g per partition
		n_part   = (inf.firlen / 2 + part_len - 1) / part_len;
		n_part   = n_part > 1 ? (n_part + 1) & ~1u : 1;           // one partition, or whole pairs
		// rows of 256 complex samples that are new per block: 16 minus what the tail of g needs
		// (g of 2048 taps: 8; 1536 taps, the LV2 48 kHz filter: 10 -> 5120 frames per transform pair)
		const uint32_t glen = inf.firlen / 2;
		uint32_t       rows = 8;
		if (n_part == 1) {
			rows = 16 - (glen - 1 + 255) / 256;
			rows = std::min<uint32_t> (14, std::max<uint32_t> (8, rows & ~1u));
		}
		hopseg   = (int)(2 * rows);                               // x 256 = new frames per block
		use_p2   = false;
	} else if (use_p2) {
		part_len = F / 2;
		n_part   = 2 * ((inf.firlen + F - 1) / F); // whole pairs (a missing last partition is all-zero taps)
		hopseg   = 8;
	} else if (inf.firlen <= F - F / 4) {
		n_part   = 1;
		part_len = inf.firlen;
		hopseg   = (inf.firlen - 1 <= F / 2) ? 8 : 4; // hop = F/2 or F/4
	} else {
		part_len = F / 2;
		n_part   = (inf.firlen + part_len - 1) / part_len;
		hopseg   = 8;
	}
	inf.fftlen     = F;
	inf.hop        = hopseg * seg;
	inf.partitions = n_part;
	// one device transform covers hop frames (polyphase) or a PAIR of hops (direct forms: two hops ride as re/im)
	inf.run_align  = use_poly ? inf.hop : 2 * inf.hop;
	// history: conv delay (+ partition delays) + the halo the OLA run start reaches back, and the dry delay
	uint32_t H = inf.conv_delay + (n_part - 1) * part_len + F;
	if (use_poly) { // a run starts n_part' blocks early (n_part' = partitions per launch), later launches are delayed by
		            // 4096 frames per partition of g; one more frame for the x[2j-1] of the first sample
		const uint32_t blk = (uint32_t)hopseg * seg;
		H = inf.conv_delay + (n_part > 2 ? (n_part - 2) * F : 0) + std::min<uint32_t> (2, n_part) * blk + 2;
	}
	H          = std::max (H, inf.dry_delay);
	inf.history = (H + 255) & ~255u;

	pl.use_p2   = use_p2;
	pl.use_poly = use_poly;
	return PROT_OK;
}

// which block kernel convolves (overlap-save on 4096- / 8192- / 16384-point blocks, or the table forms) and its tables; the pipelined
// route's streams
static int pick_block_kernels (prot_engine* e, const prot_config* cfg, EnginePlan& pl)
{
	prot_info&                 inf      = pl.inf;
	const std::vector<float>&  fir_taps = pl.fir_taps;
	const uint32_t             F        = pl.F;
	const uint32_t             n_part = pl.n_part, part_len = pl.part_len;
	const int                  hopseg   = pl.hopseg;
	const bool                 use_p2 = pl.use_p2, use_poly = pl.use_poly;
	(void)F, (void)n_part, (void)part_len, (void)hopseg, (void)use_p2, (void)use_poly, (void)fir_taps, (void)inf;
	std::vector<float>& sv_h = pl.sv_h;
	std::vector<float>& sv8_h = pl.sv8_h;
	std::vector<float>& sv16_h = pl.sv16_h;

	if (use_poly && n_part == 1 && !(cfg->flags & PROT_CFG_OVERLAP_ADD) && tuning_env ("PROT_NO_SAVE") == nullptr &&
	    2 * (inf.dry_delay - inf.conv_delay) == 256 * (32 - (uint32_t)hopseg) && inf.dry_delay % 2 == 0) {
		e->sv_flags = test_opt (PROT_TEST_EDGE_PATH_ONLY) ? 1 : 0; // (test hook: every block through the staged path)
		e->use_save = hopseg <= 24 && save_tables (fir_taps, F, sv_h, e->sv_gain, e->sv_rows) && e->sv_rows == 1;
		if (e->use_save && e->cus > 0 && tuning_env ("PROT_WG_PER_CU") == nullptr) {
			e->wg_slots = e->cus * 4; // 128 VGPRs, 38 KiB of LDS
		}
	}
	if (e->use_save && inf.kind == PROT_KIND_LV2 && !(cfg->flags & PROT_CFG_BLOCK4096) && (inf.firlen == 3072 || inf.firlen == 4096) &&
	    tuning_env ("PROT_NO_SAVE8K") == nullptr) {
		e->use_save8 = save_tables8k (fir_taps, sv8_h, e->sv8_gain);
		e->sv8_rows  = 16 - (int)(inf.firlen / 1024);
	}
	// CLI blksiz 4096 (2048 odd taps = 4 context rows of an 8192-point block, 12288 new frames per block): every fp32 launch on
	// these blocks, as for blksiz 8192 below (rotate 0.295 -> 0.27 ms, the fused sweep's Hilbert pass -20 %)
	if (e->use_save && inf.kind == PROT_KIND_CLI && inf.firlen == 4096 && inf.conv_delay == 0 && inf.dry_delay == 2048 &&
	    !(cfg->flags & (PROT_CFG_BLOCK4096 | PROT_CFG_OVERLAP_ADD | PROT_CFG_POLYPHASE)) && !(e->sv_flags & 1) && tuning_env ("PROT_NO_SAVE8K") == nullptr) {
		e->use_save8 = save_tables8k (fir_taps, sv8_h, e->sv8_gain);
		if (e->use_save8) {
			e->sv8_rows   = 12;
			e->sv8_always = true;
			inf.hop       = 12288;
			inf.run_align = 12288;
			inf.history   = std::max<uint32_t> (inf.history, 4096 + 256);
			e->info.hop = inf.hop, e->info.run_align = inf.run_align, e->info.history = inf.history;
			if (e->cus > 0 && tuning_env ("PROT_WG_PER_CU") == nullptr) {
				e->wg_slots = e->cus * 2;
			}
		}
	}
	// LV2 at 176.4 / 192 kHz (8192 taps = 8 context rows of an 8192-point block, one table-free partition): long launches only, like
	// the other LV2 rates (shorter ones keep olapoly_kernel's two partitions on 4096-point transforms)
	if (use_poly && n_part == 2 && inf.kind == PROT_KIND_LV2 && inf.firlen == 8192 && 2 * (inf.dry_delay - inf.conv_delay) == 8192 &&
	    inf.dry_delay % 2 == 0 && inf.history >= inf.dry_delay + 4096 + 2 &&
	    !(cfg->flags & (PROT_CFG_BLOCK4096 | PROT_CFG_OVERLAP_ADD | PROT_CFG_POLYPHASE)) && tuning_env ("PROT_NO_SAVE8K") == nullptr) {
		e->use_save8 = save_tables8k (fir_taps, sv8_h, e->sv8_gain);
		e->sv8_rows  = 8;
	}
	// The CLI's default engine at 44.1 / 48 kHz (blksiz 8192, cli/phase-rotate.cc:749-755): 4096 odd taps = 8 context rows of an
	// 8192-point block -- ONE table-free partition, 8192 new frames per block, instead of two 2048-tap partitions with a spectrum
	// table on 4096-point transforms (olapoly_kernel, which stays for the fp16 path and under the flags that ask for it).  EVERY
	// fp32 launch of such an engine takes this kernel, whatever its length: one form per CLI engine is what makes chunked and
	// sharded sweeps bit-identical (the block grid starts at the call's first frame either way).
	if (use_poly && n_part == 2 && inf.kind == PROT_KIND_CLI && inf.firlen == 8192 && inf.conv_delay == 0 && inf.dry_delay == 4096 &&
	    !(cfg->flags & (PROT_CFG_BLOCK4096 | PROT_CFG_OVERLAP_ADD | PROT_CFG_POLYPHASE)) && tuning_env ("PROT_NO_SAVE8K") == nullptr) {
		e->use_save8 = save_tables8k (fir_taps, sv8_h, e->sv8_gain);
		if (e->use_save8) {
			e->sv8_rows   = 8;
			e->sv8_always = true;
			inf.hop       = 8192;
			inf.run_align = 8192;
			inf.history   = std::max<uint32_t> (inf.history, 8192 + 256);
			e->info.hop = inf.hop, e->info.run_align = inf.run_align, e->info.history = inf.history;
			if (e->cus > 0 && tuning_env ("PROT_WG_PER_CU") == nullptr) {
				e->wg_slots = e->cus * 2;
			}
		}
	}
	// The CLI's default engine at 88.2 / 96 kHz (blksiz 16384, cli/phase-rotate.cc:749-755): 8192 odd taps = 8 context rows of a
	// 16384-POINT block -- ONE table-free partition, 16384 new frames per block, ONE launch (olasave16k_kernel, 1024-thread
	// workgroups, one per CU) instead of two accumulating olapoly_kernel launches of two 2048-tap partitions each through a
	// workspace in HBM.  Every fp32 launch of such an engine takes it (one form per CLI engine: chunked and sharded sweeps stay
	// bit-identical); olapoly_kernel stays for its fp16 path and under the flags that ask for it.
	if (use_poly && inf.kind == PROT_KIND_CLI && inf.firlen == 16384 && inf.conv_delay == 0 && inf.dry_delay == 8192 &&
	    !(cfg->flags & (PROT_CFG_BLOCK4096 | PROT_CFG_OVERLAP_ADD | PROT_CFG_POLYPHASE)) && tuning_env ("PROT_NO_SAVE16K") == nullptr) {
		e->use_save16 = save_tables16k (fir_taps, sv16_h, e->sv16_gain);
		e->sv16_rows  = 8;
		if (e->use_save16) {
			inf.hop       = 16384;
			inf.run_align = 16384;
			inf.history   = std::max<uint32_t> (inf.history, 16384 + 256);
			e->info.hop = inf.hop, e->info.run_align = inf.run_align, e->info.history = inf.history;
			if (e->cus > 0 && tuning_env ("PROT_WG_PER_CU") == nullptr) {
				e->wg_slots = e->cus;
			}
		}
	}
	// ... and the default at 44.1 / 48 kHz, blksiz 8192 (4096 odd taps = 4 context rows of a 16384-point block): 12 of 16 rows new
	// (24576 frames per block) where the 8192-point block of round 4 has 8 of 16 -- a third less arithmetic per frame
	if (e->use_save8 && e->sv8_always && e->sv8_rows == 8 && inf.kind == PROT_KIND_CLI && inf.firlen == 8192 && tuning_env ("PROT_NO_SAVE16K") == nullptr &&
	    tuning_env ("PROT_NO_SAVE16K_8192") == nullptr) {
		e->use_save16 = save_tables16k (fir_taps, sv16_h, e->sv16_gain);
		if (e->use_save16) {
			e->sv16_rows  = 12;
			e->use_save8  = false;
			e->sv8_always = false;
			inf.hop       = 24576;
			inf.run_align = 24576;
			inf.history   = std::max<uint32_t> (inf.history, 8192 + 256);
			e->info.hop = inf.hop, e->info.run_align = inf.run_align, e->info.history = inf.history;
			if (e->cus > 0 && tuning_env ("PROT_WG_PER_CU") == nullptr) {
				e->wg_slots = e->cus;
			}
		}
	}
	const char* pipe_env = tuning_env ("PROT_PIPELINE");
	if (e->use_save && ((cfg->flags & PROT_CFG_PIPELINE) || (pipe_env && atoi (pipe_env) != 0)) && !(pipe_env && atoi (pipe_env) == 0)) {
		// consecutive stateful prot_apply_dev calls on the engine's own streams overlap
		bool ok = e->d_pipe.ensure (16) == PROT_OK && hipMemset (e->d_pipe.p, 0, 16 * sizeof (uint32_t)) == hipSuccess;
		// (the engine's main stream is one of the three: the HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware
		// queues, 4 by default, and two pipe streams that share a queue do not overlap -- 10 us per launch instead of 5.5)
		e->pipe_stream[0] = e->stream;
		for (int i = 1; ok && i < prot_engine::kPipe; ++i) {
			ok = hipStreamCreateWithFlags (&e->pipe_stream[i], hipStreamNonBlocking) == hipSuccess;
		}
		for (int i = 0; ok && i <= prot_engine::kPipe; ++i) {
			ok = hipEventCreateWithFlags (&e->pipe_ev[i], hipEventDisableTiming) == hipSuccess;
		}
		if (!ok) {
			return (PROT_EHIP);
		}
		e->pipeline = true;
	}
	return PROT_OK;
}

// spectra of the partitions, twiddles, every device buffer of the engine and their first contents
static int upload_engine (prot_engine* e, const prot_config* cfg, EnginePlan& pl)
{
	prot_info&                 inf      = pl.inf;
	const std::vector<float>&  fir_taps = pl.fir_taps;
	const uint32_t             F        = pl.F;
	const uint32_t             n_part = pl.n_part, part_len = pl.part_len;
	const int                  hopseg   = pl.hopseg;
	const bool                 use_p2 = pl.use_p2, use_poly = pl.use_poly;
	(void)F, (void)n_part, (void)part_len, (void)hopseg, (void)use_p2, (void)use_poly, (void)fir_taps, (void)inf;
	std::vector<float2>& H_h = pl.H_h;
	std::vector<float2>& Hp_h = pl.Hp_h;
	std::vector<float2>& tw_h = pl.tw_h;
	const std::vector<float>& sv_h = pl.sv_h;
	const std::vector<float>& sv8_h = pl.sv8_h;
	const std::vector<float>& sv16_h = pl.sv16_h;
	tw_h.resize (F);

	// FIR -> spectrum / F (the 1/fftlen of the OLA c2r, cli/phase-rotate.cc:142,
	// src/phaserotate.c:393, folded into the table)
	e->fir      = fir_taps;
	e->part_len = part_len;
	// spectra / F of the partitions of `taps` (n partitions of plen taps)
	auto spectra = [&] (const std::vector<float>& taps, uint32_t n, uint32_t plen, std::vector<float2>& dst) {
		dst.assign ((size_t)F * n, make_float2 (0.f, 0.f));
		for (uint32_t sp = 0; sp < n; ++sp) {
			std::vector<double> re (F, 0.0), im (F, 0.0);
			for (uint32_t i = 0; i < plen && (size_t)sp * plen + i < taps.size (); ++i) {
				re[i] = taps[(size_t)sp * plen + i];
			}
			fft_f64 (re, im);
			for (uint32_t k = 0; k < F; ++k) {
				dst[(size_t)sp * F + k] = make_float2 ((float)(re[k] / F), (float)(im[k] / F));
			}
		}
	};
	if (use_poly) {
		std::vector<float> g (inf.firlen / 2);
		for (size_t i = 0; i < g.size (); ++i) {
			g[i] = fir_taps[2 * i + 1];
		}
		spectra (g, n_part, part_len, H_h);
	} else {
		spectra (fir_taps, n_part, part_len, H_h);
	}
	// LV2 engines with the 4096-tap FIR run the direct form (faster per block), which needs TWO halo blocks
	// per run; the polyphase form needs one.  For launches so short that a workgroup gets only a few blocks
	// (run() buffers, batched buffers) the halo dominates, so those launches use the polyphase kernel --
	// same filter, results equal to a rounding error.  fp32 launches of LV2 engines only: CLI engines keep one fp32
	// form per engine, which is what makes sharded sweeps bit-identical whatever the shard sizes.  The fp16 sample
	// path of every 4096-tap engine uses these tables too (prot_apply_dev_f16).
	bool                poly_taps_ok = inf.firlen % 4 == 0 && inf.conv_delay % 2 == 0;
	for (uint32_t i = 0; poly_taps_ok && i < inf.firlen; i += 2) {
		poly_taps_ok = fir_taps[i] == 0.f;
	}
	if (use_p2 && !use_poly && inf.firlen == F && poly_taps_ok && (cfg->flags & ~PROT_CFG_PIPELINE) == 0 && tuning_env ("PROT_NO_POLY") == nullptr) {
		std::vector<float> g (inf.firlen / 2);
		for (size_t i = 0; i < g.size (); ++i) {
			g[i] = fir_taps[2 * i + 1];
		}
		spectra (g, 1, F / 2, Hp_h);
	}
	for (uint32_t k = 0; k < F; ++k) {
		const double a = -2.0 * M_PI * (double)k / (double)F;
		tw_h[k] = make_float2 ((float)std::cos (a), (float)std::sin (a));
	}
	const size_t hn = (size_t)inf.history * inf.n_chn;
	int rc;
	if ((rc = e->d_H.ensure ((size_t)F * n_part)) ||
	    (!Hp_h.empty () && (rc = e->d_Hp.ensure (Hp_h.size ()))) || (rc = e->d_tw.ensure (F)) || (rc = e->d_lut.ensure (PROT_MAXSAMPLE * 2)) ||
	    (rc = e->d_rot.ensure (2 * (size_t)inf.n_chn)) || (rc = e->d_hist[0].ensure (hn)) || (rc = e->d_hist[2].ensure (hn)) || (rc = e->d_hist[3].ensure (hn)) ||
	    (use_poly && ((rc = e->d_hist16[0].ensure (hn)) || (rc = e->d_hist16[1].ensure (hn)))) ||
	    (rc = e->d_hist[1].ensure (hn)) || (rc = e->d_peaks.ensure ((size_t)PROT_MAXSAMPLE * inf.n_chn)) ||
	    (rc = e->d_angles.ensure (PROT_MAXSAMPLE + 8)) || (rc = e->d_thr.ensure (prot::kSweepThrFloats * (size_t)inf.n_chn)) || (rc = e->d_ctr.ensure ((size_t)inf.n_chn * 32 * 32)) ||
	    (rc = e->d_stats.ensure (4)) || (e->use_save && (rc = e->d_sv.ensure (sv_h.size ()))) ||
	    (e->use_save8 && ((rc = e->d_sv8.ensure (sv8_h.size ())) || (rc = e->d_dyn.ensure ((size_t)prot_engine::kDynSets * inf.n_chn * prot::kDynWords)))) ||
	    (e->use_save16 && (rc = e->d_sv16.ensure (sv16_h.size ())))) {
		return (rc);
	}
	if (e->use_save16 && hipMemcpy (e->d_sv16.p, sv16_h.data (), sv16_h.size () * sizeof (float), hipMemcpyHostToDevice) != hipSuccess) {
		return (PROT_EHIP);
	}
	if (e->use_save8 && hipMemset (e->d_dyn.p, 0, (size_t)prot_engine::kDynSets * inf.n_chn * prot::kDynWords * sizeof (uint32_t)) != hipSuccess) {
		return (PROT_EHIP);
	}
	e->dyn_off = (cfg->flags & PROT_CFG_STATIC_RUNS) != 0 || tuning_env ("PROT_STATIC") != nullptr;
	if (e->use_save8 && hipMemcpy (e->d_sv8.p, sv8_h.data (), sv8_h.size () * sizeof (float), hipMemcpyHostToDevice) != hipSuccess) {
		return (PROT_EHIP);
	}
	if (e->use_save && hipMemcpy (e->d_sv.p, sv_h.data (), sv_h.size () * sizeof (float), hipMemcpyHostToDevice) != hipSuccess) {
		return (PROT_EHIP);
	}
	e->info.form      = e->use_save16 ? PROT_FORM_POLY_SAVE16K : e->use_save8 ? PROT_FORM_POLY_SAVE8K : e->use_save ? PROT_FORM_POLY_SAVE : use_poly ? PROT_FORM_POLY_OLA : PROT_FORM_DIRECT;
	e->info.wg_per_cu = e->cus > 0 ? (uint32_t)(e->wg_slots / e->cus) : 0;
	if ((!Hp_h.empty () && hipMemcpy (e->d_Hp.p, Hp_h.data (), Hp_h.size () * sizeof (float2), hipMemcpyHostToDevice) != hipSuccess) ||
	    hipMemcpy (e->d_H.p, H_h.data (), H_h.size () * sizeof (float2), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy (e->d_tw.p, tw_h.data (), F * sizeof (float2), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy (e->d_lut.p, g_lut, sizeof (g_lut), hipMemcpyHostToDevice) != hipSuccess) {
		return (PROT_EHIP);
	}
	if (hipMemset (e->d_thr.p, 0, prot::kSweepThrFloats * (size_t)inf.n_chn * sizeof (float)) != hipSuccess ||
	    hipMemset (e->d_ctr.p, 0, (size_t)inf.n_chn * 32 * 32 * sizeof (unsigned)) != hipSuccess || zero_state (e) != PROT_OK ||
	    hipStreamSynchronize (e->stream) != hipSuccess) {
		return (PROT_EHIP);
	}

	return PROT_OK;
}

// LV2 engines: run()'s buffers, the run() server's mailbox and stream, the angle state machine
static int setup_lv2 (prot_engine* e)
{
	const prot_info& inf = e->info;

	if (inf.kind == PROT_KIND_LV2) {
		e->h_io.coherent = true; // (the server reads and writes the port block while it is resident)
		if (lv2_reserve (e, 8192) != PROT_OK) { // run() buffers up front; prot_lv2_set_max_block raises it
			return (PROT_ENOMEM);
		}
		// the run() server: engines on the overlap-save kernel (both LV2 rates), up to 32 channels, not pipelined
		if (e->use_poly && e->use_save && e->d_pipe.p == nullptr && inf.n_chn <= (uint32_t)prot::kServerMaxChn && (e->hopseg == 16 || e->hopseg == 20)) {
			int lo = 0, hi = 0;
			(void)hipDeviceGetStreamPriorityRange (&lo, &hi); // (hi = the numerically smallest = greatest priority)
			e->srv_ack.coherent = true;
			if (e->srv_ack.ensure (1) == PROT_OK && e->d_srv_relay.ensure (16) == PROT_OK &&
			    hipStreamCreateWithPriority (&e->srv_stream, hipStreamNonBlocking, hi) == hipSuccess &&
			    hipEventCreateWithFlags (&e->srv_ev, hipEventDisableTiming) == hipSuccess) {
				memset (e->srv_ack.p, 0, sizeof (prot::SvServerAck));
				e->srv_eligible = true;
				if (lv2_reserve (e, 8192) != PROT_OK) { // (again, now with the mailbox)
					return (PROT_ENOMEM);
				}
			}
		}
		e->interp_th = inf.blksiz * 1e-6f; // src/phaserotate.c:295
		e->interp_nm = 1.f / inf.blksiz;   // :296
		e->lv2.resize (inf.n_chn);
		for (auto& c : e->lv2) {
			c.rotq.assign ((size_t)inf.blksiz * 2, 0.f);
			for (uint32_t i = 0; i < inf.blksiz; ++i) {
				c.rotq[2 * i] = 1.f; // cos 0
			}
		}
	}
	return PROT_OK;
}

prot_engine* prot_engine_create (const prot_config* cfg, int* status)
{
	int  st_dummy;
	int& st = status ? *status : st_dummy;
	st      = PROT_OK;
	g_err[0] = 0;
	if (!cfg || cfg->n_chn < 1 || cfg->n_chn > 65535) {
		st = PROT_EINVAL;
		return nullptr;
	}
	std::call_once (g_lut_once, lut_init);
	ServerHold srv_hold; // (engine creation allocates, uploads and synchronises: nothing resident until it returns)

	EnginePlan pl;
	pl.inf.kind  = cfg->kind;
	pl.inf.n_chn = cfg->n_chn;
	if ((st = plan_engine (cfg, pl)) != PROT_OK) {
		return nullptr;
	}
	const prot_info& inf      = pl.inf;
	const int        hopseg   = pl.hopseg;
	const uint32_t   n_part   = pl.n_part;
	const bool       use_p2 = pl.use_p2, use_poly = pl.use_poly;

	int ndev = 0;
	hipError_t he = hipGetDeviceCount (&ndev);
	if (he != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
		st = PROT_ENODEV;
		snprintf (g_err, sizeof (g_err), "hipGetDeviceCount: %s (count %d, wanted device %d)",
		          hipGetErrorString (he), ndev, cfg->device);
		return nullptr;
	}
	if (hipSetDevice (cfg->device) != hipSuccess) {
		st = PROT_ENODEV;
		return nullptr;
	}

	prot_engine* e = new (std::nothrow) prot_engine ();
	if (!e) {
		st = PROT_ENOMEM;
		return nullptr;
	}
	e->info   = inf;
	e->device = cfg->device;
	e->hopseg   = hopseg;
	e->use_p2   = use_p2;
	e->use_poly = use_poly;
	{
		int cus = 0;
		if (hipDeviceGetAttribute (&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) {
			// 53 KiB of LDS per workgroup, 3 per CU -- except the kernels with two partitions in one launch (70 KiB, 2 per CU)
			e->wg_slots = cus * ((use_p2 || (use_poly && n_part >= 2)) ? 2 : 3);
			e->cus      = cus;
		}
	}
	if (const char* o = tuning_env ("PROT_WG_PER_CU")) { // tuning
		int cus = 256;
		(void)hipDeviceGetAttribute (&cus, hipDeviceAttributeMultiprocessorCount, cfg->device);
		e->wg_slots = cus * atoi (o);
		e->cus      = cus;
	}
	if (const char* o = tuning_env ("PROT_PAIRS_PER_RUN")) {
		e->pairs_per_run_override = atoi (o);
	}

	auto fail = [&] (int code) -> prot_engine* {
		st = code;
		prot_engine_destroy (e);
		return nullptr;
	};

	(void)hipEventCreateWithFlags (&e->ev_reset, hipEventDisableTiming);
	(void)hipEventCreateWithFlags (&e->ev_last, hipEventDisableTiming);
	(void)hipEventCreateWithFlags (&e->ev_own, hipEventDisableTiming);
	if (hipStreamCreateWithFlags (&e->stream, hipStreamNonBlocking) != hipSuccess) {
		return fail (PROT_EHIP);
	}
	if (e->h_slots.ensure ((size_t)prot_engine::kSlots * prot_engine::kSlotBytes) != PROT_OK) {
		return fail (PROT_ENOMEM);
	}
	for (int i = 0; i < prot_engine::kSlots; ++i) {
		if (hipEventCreateWithFlags (&e->slot_ev[i], hipEventDisableTiming) != hipSuccess) {
			return fail (PROT_EHIP);
		}
	}

	int rc;
	if ((rc = pick_block_kernels (e, cfg, pl)) || (rc = upload_engine (e, cfg, pl)) || (rc = setup_lv2 (e))) {
		return fail (rc);
	}
	e->h_peaks.assign ((size_t)PROT_MAXSAMPLE * inf.n_chn, 0.f);
	{
		std::lock_guard<std::mutex> lk (g_rccl_lock);
		++g_engines;
		e->counted = true;
	}
	return e;
}

void prot_engine_destroy (prot_engine* e)
{
	if (!e) {
		return;
	}
	(void)hipSetDevice (e->device);
	ServerHold srv_hold; // (whoever's it is: hipFree waits for the whole device)
	{
		std::lock_guard<std::mutex> lk (g_srv_mu);
		server_roster_remove (e);
	}
	if (e->srv_stream) {
		(void)hipStreamSynchronize (e->srv_stream);
		(void)hipStreamDestroy (e->srv_stream);
		e->srv_stream = nullptr;
	}
	if (e->srv_ev) {
		(void)hipEventDestroy (e->srv_ev);
		e->srv_ev = nullptr;
	}
	if (e->srv_box) {
		(void)(e->srv_bar ? hipFree (e->srv_box) : hipHostFree (e->srv_box));
		e->srv_box = nullptr;
	}
	e->srv_ack.release ();
	e->d_srv_relay.release ();
	for (int i = 0; i < prot_engine::kPipe; ++i) {
		if (e->pipe_stream[i]) {
			(void)hipStreamSynchronize (e->pipe_stream[i]);
		}
	}
	if (e->stream) {
		(void)hipStreamSynchronize (e->stream);
	}
	if (e->reset_pending && e->ev_reset) { // a reset's zeroing on a caller's stream writes the buffers released below
		(void)hipEventSynchronize (e->ev_reset);
	}
	e->d_H.release ();
	e->d_sv.release ();
	e->d_sv8.release ();
	e->d_sv16.release ();
	e->d_dyn.release ();
	e->d_surv.release ();
	e->d_surv_cnt.release ();
	e->d_nsteps.release ();
	e->d_desc.release ();
	e->d_fused_surv.release ();
	e->d_Hp.release ();
	e->d_tw.release ();
	e->d_lut.release ();
	e->d_rot.release ();
	e->d_hist[0].release ();
	e->d_hist[1].release ();
	e->d_hist[2].release ();
	e->d_hist[3].release ();
	e->d_hist16[0].release ();
	e->d_hist16[1].release ();
	e->d_peaks.release ();
	e->d_in.release ();
	e->d_out.release ();
	e->d_u.release ();
	e->d_inter.release ();
	e->d_inter_b.release ();
	e->d_file.release ();
	e->d_in_b.release ();
	for (int i = 0; i < 2; ++i) {
		if (e->ev_copied[i]) {
			(void)hipEventDestroy (e->ev_copied[i]);
		}
		if (e->ev_free[i]) {
			(void)hipEventDestroy (e->ev_free[i]);
		}
	}
	if (e->copy_stream) {
		(void)hipStreamDestroy (e->copy_stream);
	}
	e->d_rot_s.release ();
	e->d_angles.release ();
	e->d_thr.release ();
	e->d_ctr.release ();
	e->d_part.release ();
	e->d_stats.release ();
	e->h_io.release ();
	e->h_slots.release ();
	for (int i = 0; i < prot_engine::kSlots; ++i) {
		if (e->slot_ev[i]) {
			(void)hipEventDestroy (e->slot_ev[i]);
		}
	}
	for (int i = 1; i < prot_engine::kPipe; ++i) { // ([0] is the main stream)
		if (e->pipe_stream[i]) {
			(void)hipStreamDestroy (e->pipe_stream[i]);
		}
	}
	for (int i = 0; i <= prot_engine::kPipe; ++i) {
		if (e->pipe_ev[i]) {
			(void)hipEventDestroy (e->pipe_ev[i]);
		}
	}
	e->d_pipe.release ();
	if (e->stream) {
		(void)hipStreamDestroy (e->stream);
	}
	if (e->ev_last) {
		(void)hipEventSynchronize (e->ev_last);
		(void)hipEventDestroy (e->ev_last);
		e->ev_last = nullptr;
	}
	if (e->ev_own) {
		(void)hipEventDestroy (e->ev_own);
		e->ev_own = nullptr;
	}
	if (e->ev_reset) {
		(void)hipEventDestroy (e->ev_reset);
	}
	if (e->counted) {
		std::lock_guard<std::mutex> lk (g_rccl_lock);
		if (--g_engines == 0) {
			rccl_drop_comms_locked (); // the last engine takes the cached communicators with it
		}
	}
	delete e;
}

static int engine_reset_held (prot_engine* e);
int prot_engine_reset (prot_engine* e)
{
	if (!e) {
		return PROT_EINVAL;
	}
	HIP_TRY (hipSetDevice (e->device));
	const int rc_reset = engine_reset_held (e);
	return rc_reset ? rc_reset : rt_prestart (e); // (deferred run(): the resident kernel is back before the first run())
}

static int engine_reset_held (prot_engine* e)
{
	ServerHold srv_hold;
	// Behind a call on a caller's stream: stream-ordered, and no host wait (a host that analyses file after file, the bench's loop:
	// a blocking wait per pass costs 0.05-0.25 ms by the state of the runtime's queues; tools/attic/process_state_probe.py).  The zeroing
	// goes to the ENGINE's stream behind an event recorded on the last call's stream now (mark_last);
	// pick_stream puts a call that arrives on another stream behind ev_reset, the readers of device state wait for the engine's
	// stream (sync_own).
	bool foreign = false;
	if (e->last_foreign) {
		// (when the event cannot be recorded behind the last call, nothing would order the zeroing behind the kernels that
		// still use the history and the tables: wait for that stream on the host, or give up)
		if (e->ev_reset != nullptr && mark_last (e)) {
			foreign = true;
		} else if (e->last_stream_id == nullptr || hipStreamSynchronize ((hipStream_t)const_cast<void*> (e->last_stream_id)) != hipSuccess) {
			snprintf (g_err, sizeof (g_err), "prot_engine_reset: cannot order the reset behind the last call's stream");
			return PROT_EHIP;
		}
	}
	hipStream_t zs      = e->stream;
	if (foreign) {
		HIP_TRY (hipStreamWaitEvent (zs, e->ev_last, 0));
	}
	(void)join_pipe (e, zs);
	int rc = zero_state (e, zs);
	if (rc) {
		return rc;
	}
	for (auto& c : e->lv2) { // channel_reset: offset = 0, buf_out = 0 (angle state kept, Q10)
		c.offset = 0;
	}
	if (foreign) {
		HIP_TRY (hipEventRecord (e->ev_reset, zs));
		e->reset_stream  = zs;
		e->reset_pending = true;
		e->own_dirty     = false; // (ev_reset sits behind everything the engine's stream held)
	} else {
		// behind a call on the engine's own stream (stream = NULL): the caller has no way to order anything of its own against it
		// but the host -- a table of the caller's zeroed between reset and the next call, say -- so the reset still ends with the wait
		HIP_TRY (hipStreamSynchronize (zs));
		e->reset_pending = false;
		e->own_dirty     = false;
	}
	e->sweep_seeded_ext = false;
	e->seed_chunk_ready = false;
	if (e->rt_frames > 0) { // deferred run(): rt_frames frames of silence again (activate: src/phaserotate.c:538-560 clears every buffer)
		rt_ring_reset (e);
	}
	e->last_foreign   = false; // (whatever comes next is ordered behind the engine's stream, which is behind that call)
	e->last_stream_id = nullptr;
	return PROT_OK;
}

int prot_engine_info (const prot_engine* e, prot_info* out)
{
	if (!e || !out) {
		return PROT_EINVAL;
	}
	*out = e->info;
	return PROT_OK;
}

int prot_engine_fir (const prot_engine* e, float* fir_out)
{
	if (!e || !fir_out) {
		return PROT_EINVAL;
	}
	memcpy (fir_out, e->fir.data (), e->fir.size () * sizeof (float));
	return PROT_OK;
}

int prot_engine_sync (prot_engine* e)
{
	if (!e) {
		return PROT_EINVAL;
	}
	HIP_TRY (hipSetDevice (e->device));
	ServerHold srv_hold;
	(void)join_pipe (e, e->stream);
	if (mark_last (e)) { // the last call ran on a stream of the caller's: wait for it too, and let go of the handle
		HIP_TRY (hipEventSynchronize (e->ev_last));
	}
	e->last_foreign   = false;
	e->last_stream_id = nullptr;
	HIP_TRY (sync_own (e));
	if (e->pipe_used && e->d_pipe.p) {
		// did a pipelined launch give up waiting for the history of the one before it (prot_olasave.hip: bounded wait)?
		uint32_t st = 0;
		HIP_TRY (hipMemcpy (&st, e->d_pipe.p + prot::kPipeStatusWord, sizeof (st), hipMemcpyDeviceToHost));
		e->pipe_used = false;
		if (st != 0u) {
			(void)hipMemset (e->d_pipe.p + prot::kPipeStatusWord, 0, sizeof (st));
			snprintf (g_err, sizeof (g_err), "pipelined launch: the history of generation %u never arrived (outputs since then are invalid)", st);
			return PROT_EHIP;
		}
	}
	return PROT_OK;
}

// ---------------------------------------------------------------- device API
} // extern "C"

