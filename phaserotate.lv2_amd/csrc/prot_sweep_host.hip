// prot_sweep_host.hip -- prot_sweep_dev and its routes (fused / separate passes, seeding, priming), the peak table's readers, the CLI's block and file calls
// (split from prot_capi.hip in round 6: same code; what the units share is in prot_engine.h)
#include "prot_engine.h"

extern "C" {

// one prot_sweep_dev call, as its three parts see it
struct SweepCall {
	const float* d_in;
	int64_t      in_stride;
	uint64_t     n_frames, n_out;
	uint32_t     P;
	bool         beg, end, raw0, fuse_poly;
	int          c0, nc, n_ang, seed_blocks;
	float*       pk;
	const int*   lut_idx;
	hipStream_t  s;
};

// FUSED (round 4): the sweep's test rides in the Hilbert pass (MODE_SWEEP: nothing but the surviving (d, h) pairs -- a few percent
// of the frames -- is written), a seeding pass over a sparse sample of the call first when the table holds no bound yet, then plan ->
// matrix-core evaluation of the lists -> merge; seed_blocks > 0: the seeding pass only (prot_sweep_seed_dev)
static int sweep_fused (prot_engine* e, const SweepCall& k)
{
	const float* const d_in = k.d_in;
	const int64_t      in_stride = k.in_stride;
	const uint64_t     n_frames = k.n_frames, n_out = k.n_out;
	const uint32_t     P = k.P;
	const bool         beg = k.beg, end = k.end, raw0 = k.raw0;
	const int          c0 = k.c0, nc = k.nc, n_ang = k.n_ang, seed_blocks = k.seed_blocks;
	float* const       pk = k.pk;
	const int* const   lut_idx = k.lut_idx;
	hipStream_t        s = k.s;
	int                rc = PROT_OK;
	(void)beg, (void)end, (void)raw0, (void)c0, (void)nc, (void)n_ang, (void)seed_blocks, (void)pk, (void)lut_idx, (void)P, (void)n_frames, (void)d_in, (void)in_stride, (void)rc;
	const bool fuse_poly = k.fuse_poly;
	const float* hist_old = e->d_hist[e->hist_cur].p; (void)hist_old;
	if (e->pos == 0 && !e->sweep_seeded_ext) {
		e->sweep_seeded = false;
	}
	if (e->sweep_seeded_ext && seed_blocks == 0) { // the bound was seeded (and merged across shards) by the caller: two-phase sharded sweeps
		e->sweep_seeded     = true;
		e->sweep_seeded_ext = false;
	}
	int rc2;
	const bool first_fused = e->d_fused_surv.p == nullptr;
	// where the Hilbert launches write: nowhere (one launch: the overlap-save kernel lists only), or the workspace for the partial
	// sums of the launches before the last
	float*       f_out     = const_cast<float*> (d_in);
	int64_t      f_ostride = in_stride;
	if (fuse_poly) {
		const size_t ustride = (size_t)((n_out + 255) & ~(uint64_t)255);
		int          rcu     = e->d_u.ensure (ustride * e->info.n_chn);
		if (rcu) {
			return rcu;
		}
		f_out     = e->d_u.p;
		f_ostride = (int64_t)ustride;
	}
	if ((rc2 = e->d_part.ensure (prot::sweep_scratch_floats ())) || (rc2 = e->d_fused_surv.ensure (1))) {
		return rc2;
	}
	if (first_fused) {
		HIP_TRY (hipMemsetAsync (e->d_fused_surv.p, 0, sizeof (unsigned long long), s));
	}
	// the bound is seeded from a sparse sample of the call: every k-th block, about 64 of them, at most every 8th
	const uint64_t n_blocks    = n_out / (uint64_t)e->info.run_align;
	const bool     sample_pass = seed_blocks > 0 || (!e->sweep_seeded && n_blocks >= 64);
	auto fill = [&] (FusedSweep& fs, int pass) {
		fs.peaks         = pk;
		fs.angles        = e->d_angles.p;
		fs.n_angles      = n_ang;
		fs.raw_slot      = raw0 ? 0 : -1;
		fs.raw_tail      = (raw0 && !end && pass == 1) ? 1 : 0;
		fs.sample        = pass == 0;
		fs.stride_blocks = (int)std::max<uint64_t> (8, n_blocks / 64); // (every 64th ... 512th block of an hour of material:
		// the same times to the run-to-run spread; the bound hardly depends on it)
		if (seed_blocks > 0) { // a shard's share of the sample (prot_sweep_seed_dev)
			fs.stride_blocks = (int)std::max<uint64_t> (1, std::min<uint64_t> (n_blocks / (uint64_t)seed_blocks, (uint64_t)1 << 20));
		}
		if (const char* o = tuning_env ("PROT_FUSE_STRIDE")) {
			fs.stride_blocks = std::max (2, atoi (o));
		}
		fs.t_begin       = beg ? P / 2 : 0; // :418-419
		fs.dry_from      = beg ? P : 0;     // (local frames: the first block's dry term is the silent previous block, Q1)
	};
	{ // size the list buffers for both passes before anything is launched
		size_t need_surv = 0, need_cnt = 0, need_desc = 0;
		for (int pass = sample_pass ? 0 : 1; pass < 2; ++pass) {
			FusedSweep fs {};
			fill (fs, pass);
			fs.plan_only = true;
			rc = ola_launch (e, d_in, in_stride, n_frames, f_out, f_ostride, n_out, nullptr, nullptr, true, false, c0, nc, s, false, nullptr,
			                 nullptr, &fs);
			if (rc) {
				return rc;
			}
			need_surv = std::max (need_surv, (size_t)fs.n_lists * fs.cap * nc + 2048);
			need_cnt  = std::max (need_cnt, 2 * (size_t)fs.n_lists * nc);
			need_desc = std::max (need_desc, ((size_t)fs.n_lists + (size_t)(n_out / 1024) + 8) * nc);
		}
		if ((rc2 = e->d_surv.ensure (need_surv)) || (rc2 = e->d_surv_cnt.ensure (need_cnt)) || (rc2 = e->d_desc.ensure (need_desc)) ||
		    (rc2 = e->d_nsteps.ensure ((size_t)e->info.n_chn))) {
			return rc2;
		}
	}
	for (int pass = sample_pass ? 0 : 1; pass < (seed_blocks > 0 ? 1 : 2); ++pass) {
		FusedSweep fs {};
		fill (fs, pass);
		rc = ola_launch (e, d_in, in_stride, n_frames, f_out, f_ostride, n_out, nullptr, nullptr, true, false, c0, nc, s,
		                 /*advance_hist*/ pass == 1, nullptr, nullptr, &fs);
		if (rc) {
			return rc;
		}
		e->last_n_lists = fs.n_lists;
		e->last_cap     = fs.cap;
		if (pass == 0 && test_opt (PROT_TEST_SEED_MFMA) == 0) {
			// the seeding pass's lists: a lower bound is all they have to give (prot_sweep.hip: sweep_seed_eval_kernel)
			hipError_t err0 = prot::launch_sweep_seed_eval (e->d_surv.p, e->d_surv_cnt.p, fs.n_lists, fs.cap, e->d_angles.p, n_ang, e->d_lut.p, pk, c0, nc,
			                                                e->d_part.p, prot::sweep_scratch_floats (), s);
			if (err0 == hipSuccess) {
				continue;
			}
			if (err0 != hipErrorInvalidValue) { // (InvalidValue: more lists than the scratch table holds -- the matrix-core route below)
				return hip_fail (err0, "sweep seed evaluation launch");
			}
		}
		const int64_t desc_stride = (int64_t)fs.n_lists + (int64_t)(n_out / 1024) + 8;
		hipError_t    err = prot::launch_sweep_plan (e->d_surv_cnt.p, fs.n_lists, fs.cap, e->d_desc.p, desc_stride, e->d_nsteps.p, nc,
		                                             pass == 1 ? e->d_fused_surv.p : nullptr, s);
		if (err != hipSuccess) {
			return hip_fail (err, "sweep plan launch");
		}
		prot::SweepParams sp {};
		sp.surv        = e->d_surv.p;
		sp.desc        = e->d_desc.p;
		sp.desc_stride = desc_stride;
		sp.n_steps     = e->d_nsteps.p;
		sp.t_begin     = 0;
		sp.t_end       = pass == 0 ? (int64_t)((n_out / ((uint64_t)fs.stride_blocks * e->info.run_align) + 1) * e->info.run_align) : (int64_t)n_out;
		sp.angles      = e->d_angles.p;
		sp.n_angles    = n_ang;
		sp.lut         = e->d_lut.p;
		sp.peaks       = pk;
		sp.chn0        = c0;
		sp.prune       = 1;
		sp.thr         = e->d_thr.p;
		sp.thr_blk     = e->d_thr.p + 4 * (size_t)e->info.n_chn;
		sp.angles_host = lut_idx;
		sp.stats       = e->d_stats.p;
		sp.ctr         = e->d_ctr.p;
		sp.part        = e->d_part.p;
		sp.part_floats = prot::sweep_scratch_floats ();
		sp.raw_slot    = -1; // (the raw peak went into the table from the fused pass)
		err            = prot::launch_sweep (sp, nc, s);
		if (err != hipSuccess) {
			(void)hipMemsetAsync (e->d_thr.p, 0, prot::kSweepThrFloats * (size_t)e->info.n_chn * sizeof (float), s);
			(void)hipMemsetAsync (e->d_ctr.p, 0, (size_t)e->info.n_chn * 32 * 32 * sizeof (unsigned), s);
			return hip_fail (err, "sweep kernel launch (lists)");
		}
	}
	if (seed_blocks > 0) {
		e->sweep_seeded_ext = true;
		return PROT_OK;
	}
	e->sweep_seeded = true;
	e->fused_frames += n_out * (uint64_t)nc;
	return PROT_OK;
}

// the separate passes (short calls, unpruned sweeps, engines without a listing kernel): Hilbert branch to the workspace, then
// sweep_kernel over input and workspace
static int sweep_separate (prot_engine* e, const SweepCall& k)
{
	const float* const d_in = k.d_in;
	const int64_t      in_stride = k.in_stride;
	const uint64_t     n_frames = k.n_frames, n_out = k.n_out;
	const uint32_t     P = k.P;
	const bool         beg = k.beg, end = k.end, raw0 = k.raw0;
	const int          c0 = k.c0, nc = k.nc, n_ang = k.n_ang, seed_blocks = k.seed_blocks;
	float* const       pk = k.pk;
	const int* const   lut_idx = k.lut_idx;
	hipStream_t        s = k.s;
	int                rc = PROT_OK;
	(void)beg, (void)end, (void)raw0, (void)c0, (void)nc, (void)n_ang, (void)seed_blocks, (void)pk, (void)lut_idx, (void)P, (void)n_frames, (void)d_in, (void)in_stride, (void)rc;
	const bool unpruned = test_opt (PROT_TEST_SWEEP_UNPRUNED) != 0;
	// 1. Hilbert branch u[t], t in [0, n_out)
	const size_t ustride = (size_t)((n_out + 255) & ~(uint64_t)255);
	rc = e->d_u.ensure (ustride * e->info.n_chn);
	if (rc) {
		return rc;
	}
	const float* hist_old = e->d_hist[e->hist_cur].p; // stays intact: the launch writes the next buffer of the ring
	// (the launch also advances the input history -- fused into the kernel when all channels are swept, a small launch behind
	// it otherwise: state step 4 of the reference, the reference has consumed the data blocks; the sweep below still reads
	// the buffer of the ring that was current on entry)
	rc = ola_launch (e, d_in, in_stride, n_frames, e->d_u.p, (int64_t)ustride, n_out, nullptr, nullptr, true,
	                 false, c0, nc, s, /*advance_hist*/ true);
	if (rc) {
		return rc;
	}
	// 2. rotated peaks
	if (n_ang > 0) {
		prot::SweepParams sp {};
		sp.x        = d_in;
		sp.x_stride = in_stride;
		sp.n_x      = (int64_t)n_frames;
		sp.hist     = hist_old;
		sp.hist_len = (int)e->info.history;
		sp.u        = e->d_u.p;
		sp.u_stride = (int64_t)ustride;
		sp.t_begin  = beg ? P / 2 : 0;               // :418-419: first block looks at hil[firlen..parsiz) only
		sp.t_end    = (int64_t)n_out;
		sp.t_abs0   = (int64_t)e->pos;
		sp.dry_from = beg ? (int64_t)e->pos + P : 0; // ... paired with the (silent) previous block
		sp.delay    = (int)P / 2;
		sp.angles   = e->d_angles.p;
		sp.n_angles = n_ang;
		sp.lut      = e->d_lut.p;
		sp.peaks    = pk;
		sp.chn0     = c0;
		sp.prune       = unpruned ? 0 : 1; // exact either way; the unpruned sweep is for timing and tests (prot_internal.h)
		sp.thr         = e->d_thr.p;
		sp.stats       = e->d_stats.p;
		if ((rc = e->d_part.ensure (prot::sweep_scratch_floats ())) != PROT_OK) {
			return rc;
		}
		sp.ctr         = e->d_ctr.p;
		sp.part        = e->d_part.p;
		sp.part_floats = prot::sweep_scratch_floats ();
		sp.raw_slot    = raw0 ? 0 : -1; // the raw peak rides along with the sweep (no pass of its own over the input)
		sp.raw_tail    = (raw0 && !end) ? 1 : 0; // ... and is complete for this call's frames when the call returns
		// bounds are per launch: they refer to THIS call's angle set
		// (p.thr is all zeros here: zeroed once when the engine was created, and again by every launch's merge kernel)
		hipError_t err = prot::launch_sweep (sp, nc, s);
		if (err != hipSuccess) {
			// the launch may have died between the sweep and its merge kernel: the shared bounds and step counters are
			// not known to be zero any more
			(void)hipMemsetAsync (e->d_thr.p, 0, prot::kSweepThrFloats * (size_t)e->info.n_chn * sizeof (float), s);
			(void)hipMemsetAsync (e->d_ctr.p, 0, (size_t)e->info.n_chn * 32 * 32 * sizeof (unsigned), s);
			return hip_fail (err, "sweep kernel launch");
		}
	}
	// 3. angle 0: raw peak of the (un-delayed) blocks -- on its own only when no other angle was asked for
	if (raw0 && n_ang == 0) {
		hipError_t err = prot::launch_absmax (d_in, in_stride, (int64_t)n_frames, pk, 0, c0, nc, s);
		if (err != hipSuccess) {
			return hip_fail (err, "absmax kernel launch");
		}
	}
	return PROT_OK;
}

// the padding / flush frames of a call are silence that also entered the history
static int sweep_pad_history (prot_engine* e, const SweepCall& k)
{
	const float* const d_in = k.d_in;
	const int64_t      in_stride = k.in_stride;
	const uint64_t     n_frames = k.n_frames, n_out = k.n_out;
	const uint32_t     P = k.P;
	const bool         beg = k.beg, end = k.end, raw0 = k.raw0;
	const int          c0 = k.c0, nc = k.nc, n_ang = k.n_ang, seed_blocks = k.seed_blocks;
	float* const       pk = k.pk;
	const int* const   lut_idx = k.lut_idx;
	hipStream_t        s = k.s;
	int                rc = PROT_OK;
	(void)beg, (void)end, (void)raw0, (void)c0, (void)nc, (void)n_ang, (void)seed_blocks, (void)pk, (void)lut_idx, (void)P, (void)n_frames, (void)d_in, (void)in_stride, (void)rc;
	// 4. state: the reference has consumed nb blocks (zero padding included): the data frames went into the history with
	// the Hilbert launch above
	if (n_out > n_frames) {
		// padding/flush frames are silence that also entered the history
		const uint64_t extra = n_out - n_frames;
		// shift history by `extra` frames of silence
		const int    H    = (int)e->info.history;
		const int    cur  = e->hist_cur;
		const float* hin  = e->d_hist[cur].p;
		float*       hout = e->d_hist[(cur + 1) % prot_engine::kHist].p;
		if (nc != (int)e->info.n_chn) {
			HIP_TRY (hipMemcpyAsync (hout, hin, (size_t)H * e->info.n_chn * sizeof (float), hipMemcpyDeviceToDevice, s));
		}
		// (one small launch: hout[c][i] = [hin | silence](extra + i), the input being empty)
		hipError_t herr = prot::launch_hist_update (hin + (size_t)c0 * H, (int64_t)H, 0, hin + (size_t)c0 * H, hout + (size_t)c0 * H, H, nc, s,
		                                            (int64_t)std::min<uint64_t> (extra, (uint64_t)H));
		if (herr != hipSuccess) {
			return hip_fail (herr, "history shift launch");
		}
		e->hist_cur = (cur + 1) % prot_engine::kHist;
	}
	return PROT_OK;
}

static int sweep_dev_impl (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_frames, int ang_start,
                           int ang_end, int ang_stride, int chn, float* d_peaks, uint32_t flags, void* stream, int seed_blocks)
{
	if (!e || !d_in || e->info.kind != PROT_KIND_CLI || ang_stride < 1 || chn >= (int)e->info.n_chn) {
		return PROT_EINVAL;
	}
	const uint32_t P   = e->info.blksiz;
	const bool     end = flags & PROT_F_STREAM_END;
	const bool     beg = flags & PROT_F_STREAM_START;
	if (!end && (n_frames % P) != 0) {
		return PROT_EINVAL;
	}
	HIP_TRY (hipSetDevice (e->device));
	// The fused route's lists can hold every frame of the call (one 8-byte entry per channel-frame, whatever survives) and its step
	// descriptors address them with 32-bit entry offsets: a call of more than 2^31 channel-frames (12 GiB of input; 1 h at 96 kHz
	// with seven channels) is taken in two halves on a chunk boundary, each a call of its own -- exactly what a host that feeds the
	// file in chunks does, so the table is the same bit for bit (above ~4.29e9 channel-frames the offsets wrapped
	// silently).  PROT_TEST_FUSED_SPLIT_LOG2 lowers the limit for the test of this path.
	{
		const int      lg    = test_opt (PROT_TEST_FUSED_SPLIT_LOG2);
		const uint64_t limit = (uint64_t)1 << (lg > 0 ? lg : 31);
		const uint64_t ncc   = chn < 0 ? e->info.n_chn : 1;
		const uint64_t L     = file_align_frames (e);
		if (seed_blocks == 0 && n_frames * ncc > limit && n_frames >= 2 * L) {
			const uint64_t half = std::max<uint64_t> (L, (n_frames / 2) / L * L);
			int rc = sweep_dev_impl (e, d_in, in_stride, half, ang_start, ang_end, ang_stride, chn, d_peaks, flags & ~(uint32_t)PROT_F_STREAM_END, stream, 0);
			if (rc) {
				return rc;
			}
			return sweep_dev_impl (e, d_in + half, in_stride, n_frames - half, ang_start, ang_end, ang_stride, chn, d_peaks,
			                       flags & ~(uint32_t)PROT_F_STREAM_START, stream, 0);
		}
	}
	// (a long call in two pieces -- the first eighth listed against the seeding pass's bound, the rest against the table it leaves --
	// was measured and changes nothing: 1.77-1.80 / 4.12 ms in one piece, 1.76-2.09 / 4.07-4.32 ms in two, harmonic / noise, cfg4)
	// (two routes: the overlap-save kernel's MODE_SWEEP -- blksiz 2048, 4096 --, and for the long FIRs whose Hilbert branch takes
	// several accumulating olapoly launches -- blksiz 16384, 32768 -- the LAST of those launches, which then tests the complete
	// Hilbert value instead of storing it; the earlier launches leave their partial sums in the engine's Hilbert workspace)
	// (... and the one launch of CLI blksiz 1024, whose 14-row blocks the overlap-save kernel does not take)
	const bool fuse_poly = e->use_poly && !e->use_save && !e->sv8_always && !e->use_save16 &&
	                       ((e->info.partitions >= 4 && e->hopseg == 16) || (e->info.partitions == 1 && e->hopseg == 28));
	const bool fusable = ((e->use_save && (e->hopseg == 16 || e->hopseg == 24) && !(e->sv_flags & 1)) || fuse_poly || ((e->sv8_always || e->use_save16) && !(e->sv_flags & 1))) &&
	                     !test_opt (PROT_TEST_SWEEP_UNPRUNED) && !test_opt (PROT_TEST_SWEEP_UNFUSED);
	CallStream  s (e, stream);
	const int   c0  = chn < 0 ? 0 : chn;                       // cli/phase-rotate.cc:434-435
	const int   nc  = chn < 0 ? (int)e->info.n_chn : 1;
	float*      pk  = d_peaks ? d_peaks : e->d_peaks.p;
	if (!d_peaks) {
		e->h_peaks_valid = false;
	}

	// blocks of this chunk: data blocks (last one zero-padded, :577-580) + flush block (:585-586)
	const uint64_t nb    = (n_frames + P - 1) / P + (end ? 1 : 0);
	const uint64_t n_out = nb * P;
	if (n_out == 0) {
		return PROT_OK;
	}

	// angle list with the loop bounds of cli/phase-rotate.cc:409-428
	int  lut_idx[PROT_MAXSAMPLE + 8];
	int  n_ang = 0;
	bool raw0  = false;
	for (int angle = ang_start; angle <= ang_end;) {
		const int a = ((angle % PROT_MAXSAMPLE) + PROT_MAXSAMPLE) % PROT_MAXSAMPLE; // (angle + MAXSAMPLE) % MAXSAMPLE for angle > -360
		if (angle == 0) {
			raw0 = true; // :413-414 peak of the un-delayed block
		} else if (n_ang < PROT_MAXSAMPLE) {
			lut_idx[n_ang++] = a;
		}
		angle += ang_stride;
		if (angle >= ang_end) {
			break;
		}
	}

	// the angle list of a file analysis is the same for every chunk: uploaded when it changes (or the stream does)
	const bool new_angles = n_ang > 0 && (e->angles_stream != s || e->angles_last.size () != (size_t)n_ang || memcmp (e->angles_last.data (), lut_idx, n_ang * sizeof (int)));
	int rc = PROT_OK;
	if (new_angles) {
		rc = stage_small (e, e->d_angles.p, lut_idx, n_ang * sizeof (int), s);
		if (rc) {
			return rc;
		}
		e->angles_last.assign (lut_idx, lut_idx + n_ang);
		e->angles_stream = s;
		e->sweep_seeded  = false;
		e->sweep_seeded_ext = false; // (a seed for ANOTHER angle list or stream: this call seeds itself -- ADVICE r5)
	}
	const bool fused = fusable && n_ang > 0 && n_out >= ((uint64_t)1 << 20); // (calls long enough to carry five more launches)
	if (seed_blocks > 0 && !fused) {
		return PROT_OK;
	}
	const SweepCall k { d_in, in_stride, n_frames, n_out, P, beg, end, raw0, fuse_poly, c0, nc, n_ang, seed_blocks, pk, lut_idx, s };
	rc = fused ? sweep_fused (e, k) : sweep_separate (e, k);
	if (rc || seed_blocks > 0) {
		return rc;
	}
	// state: the reference has consumed nb blocks (zero padding included): the data frames went into the history with the Hilbert launch
	if ((rc = sweep_pad_history (e, k)) != PROT_OK) {
		return rc;
	}
	e->pos += n_out;
	return PROT_OK;
}

int prot_sweep_dev (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_frames, int ang_start,
                    int ang_end, int ang_stride, int chn, float* d_peaks, uint32_t flags, void* stream)
{
	return sweep_dev_impl (e, d_in, in_stride, n_frames, ang_start, ang_end, ang_stride, chn, d_peaks, flags, stream, 0);
}

int prot_sweep_seed_dev (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_frames, int ang_start,
                         int ang_end, int ang_stride, int chn, float* d_peaks, uint32_t flags, int n_blocks, void* stream)
{
	if (n_blocks < 1) {
		return PROT_EINVAL;
	}
	return sweep_dev_impl (e, d_in, in_stride, n_frames, ang_start, ang_end, ang_stride, chn, d_peaks, flags, stream, n_blocks);
}

int prot_engine_prime_dev (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_frames, void* stream)
{
	if (!e || (!d_in && n_frames)) {
		return PROT_EINVAL;
	}
	if (n_frames == 0) {
		return PROT_OK;
	}
	HIP_TRY (hipSetDevice (e->device));
	CallStream s (e, stream);
	const int  rc = hist_advance (e, d_in, in_stride, n_frames, 0, (int)e->info.n_chn, s);
	if (rc == PROT_OK) {
		e->pos += n_frames;
	}
	return rc;
}

int prot_sweep_shard_seed_dev (prot_engine* e, const float* d_halo, int64_t halo_stride, uint64_t n_halo, const float* d_in, int64_t in_stride,
                               uint64_t n_frames, int ang_start, int ang_end, int ang_stride, float* d_peaks, uint32_t flags, int n_blocks,
                               void* stream)
{
	if (!e || !d_in || n_blocks < 1 || e->info.kind != PROT_KIND_CLI) {
		return PROT_EINVAL;
	}
	int rc = prot_engine_reset (e);
	if (rc) {
		return rc;
	}
	if (d_peaks) {
		// (behind whatever the caller's stream still does with the table, in front of the seeding pass's kernels: same stream)
		HIP_TRY (hipSetDevice (e->device));
		HIP_TRY (hipMemsetAsync (d_peaks, 0, (size_t)PROT_MAXSAMPLE * e->info.n_chn * sizeof (float), stream ? (hipStream_t)stream : e->stream));
	}
	if (d_halo && n_halo && (rc = prot_engine_prime_dev (e, d_halo, halo_stride, n_halo, stream)) != PROT_OK) {
		return rc;
	}
	return sweep_dev_impl (e, d_in, in_stride, n_frames, ang_start, ang_end, ang_stride, -1, d_peaks, flags, stream, n_blocks);
}

float* prot_cli_peaks_dev (prot_engine* e)
{
	return e ? e->d_peaks.p : nullptr;
}

int prot_sweep_stats_ex (prot_engine* e, uint64_t out[4])
{
	if (!e || !out) {
		return PROT_EINVAL;
	}
	HIP_TRY (hipSetDevice (e->device));
	(void)join_pipe (e, e->stream);
	HIP_TRY (sync_own (e));
	unsigned long long h[4] = { 0, 0, 0, 0 };
	HIP_TRY (hipMemcpy (h, e->d_stats.p, sizeof (h), hipMemcpyDeviceToHost));
	for (int i = 0; i < 4; ++i) {
		out[i] = h[i];
	}
	return PROT_OK;
}

int prot_sweep_stats_fused (prot_engine* e, uint64_t out[2])
{
	if (!e || !out) {
		return PROT_EINVAL;
	}
	HIP_TRY (hipSetDevice (e->device));
	(void)join_pipe (e, e->stream);
	HIP_TRY (sync_own (e));
	unsigned long long kept = 0;
	if (e->d_fused_surv.p) {
		HIP_TRY (hipMemcpy (&kept, e->d_fused_surv.p, sizeof (kept), hipMemcpyDeviceToHost));
	}
	out[0] = e->fused_frames;
	out[1] = kept;
	return PROT_OK;
}

int prot_sweep_stats (prot_engine* e, uint64_t* evaluated, uint64_t* seen)
{
	if (!e || !evaluated || !seen) {
		return PROT_EINVAL;
	}
	uint64_t  h[4];
	const int rc = prot_sweep_stats_ex (e, h);
	if (rc == PROT_OK) {
		*evaluated = h[0];
		*seen      = h[1];
	}
	return rc;
}

int prot_cli_peaks (prot_engine* e, float* out)
{
	if (!e || !out) {
		return PROT_EINVAL;
	}
	HIP_TRY (hipSetDevice (e->device));
	(void)join_pipe (e, e->stream);
	HIP_TRY (sync_own (e));
	HIP_TRY (hipMemcpy (out, e->d_peaks.p, (size_t)PROT_MAXSAMPLE * e->info.n_chn * sizeof (float), hipMemcpyDeviceToHost));
	return PROT_OK;
}

static int refresh_peaks (prot_engine* e)
{
	if (e->h_peaks_valid) {
		return PROT_OK;
	}
	int rc = prot_cli_peaks (e, e->h_peaks.data ());
	if (rc == PROT_OK) {
		e->h_peaks_valid = true;
	}
	return rc;
}

float prot_cli_peak_all (prot_engine* e, int a)
{
	if (!e || refresh_peaks (e)) {
		return NAN;
	}
	float p = 0; // cli/phase-rotate.cc:287-299
	if (a < 0) {
		a += PROT_MAXSAMPLE;
	}
	a = a % PROT_MAXSAMPLE;
	for (uint32_t c = 0; c < e->info.n_chn; ++c) {
		p = std::max (p, e->h_peaks[(size_t)c * PROT_MAXSAMPLE + a]);
	}
	return p;
}

float prot_cli_peak (prot_engine* e, int c, int a)
{
	if (!e) {
		return NAN;
	}
	if (c < 0 || (uint32_t)c >= e->info.n_chn) { // cli/phase-rotate.cc:278 (its '>' would read out of bounds at c == n_chn)
		return prot_cli_peak_all (e, a);
	}
	if (refresh_peaks (e)) {
		return NAN;
	}
	if (a < 0) {
		a += PROT_MAXSAMPLE;
	}
	return e->h_peaks[(size_t)c * PROT_MAXSAMPLE + (a % PROT_MAXSAMPLE)];
}

// ------------------------------------------------------------ host drop-ins
int prot_cli_apply (prot_engine* e, float* buf, const int* angles)
{
	if (!e || !buf || !angles || e->info.kind != PROT_KIND_CLI) {
		return PROT_EINVAL;
	}
	HIP_TRY (hipSetDevice (e->device));
	const uint32_t nc = e->info.n_chn;
	const size_t   P  = e->info.blksiz;
	CallStream    s (e, nullptr);
	int rc;
	if ((rc = e->d_inter.ensure (P * nc)) || (rc = e->d_in.ensure (P * nc)) || (rc = e->d_out.ensure (P * nc))) {
		return rc;
	}
	std::vector<float> rot (2 * (size_t)nc);
	for (uint32_t c = 0; c < nc; ++c) {
		const int a = ((angles[c] % PROT_MAXSAMPLE) + PROT_MAXSAMPLE) % PROT_MAXSAMPLE; // :463
		rot[2 * c]     = g_lut[2 * a + 1]; // ca
		rot[2 * c + 1] = g_lut[2 * a];     // sa
	}
	HIP_TRY (hipMemcpyAsync (e->d_inter.p, buf, P * nc * sizeof (float), hipMemcpyHostToDevice, s));
	hipError_t err = prot::launch_deinterleave (e->d_inter.p, e->d_in.p, (int64_t)P, (int64_t)P, (int)nc, s);
	if (err != hipSuccess) {
		return hip_fail (err, "deinterleave launch");
	}
	rc = prot_apply_dev (e, e->d_in.p, (int64_t)P, P, e->d_out.p, (int64_t)P, P, rot.data (), 0, s);
	if (rc) {
		return rc;
	}
	err = prot::launch_interleave (e->d_out.p, (int64_t)P, e->d_inter.p, (int64_t)P, (int)nc, s);
	if (err != hipSuccess) {
		return hip_fail (err, "interleave launch");
	}
	HIP_TRY (hipMemcpyAsync (buf, e->d_inter.p, P * nc * sizeof (float), hipMemcpyDeviceToHost, s));
	HIP_TRY (hipStreamSynchronize (s));
	return PROT_OK;
}

int prot_cli_analyze (prot_engine* e, const float* buf, int ang_start, int ang_end, int ang_stride, int chn,
                      int start)
{
	if (!e || !buf || e->info.kind != PROT_KIND_CLI) {
		return PROT_EINVAL;
	}
	HIP_TRY (hipSetDevice (e->device));
	const uint32_t nc = e->info.n_chn;
	const size_t   P  = e->info.blksiz;
	CallStream    s (e, nullptr);
	int rc;
	if ((rc = e->d_inter.ensure (P * nc)) || (rc = e->d_in.ensure (P * nc))) {
		return rc;
	}
	HIP_TRY (hipMemcpyAsync (e->d_inter.p, buf, P * nc * sizeof (float), hipMemcpyHostToDevice, s));
	hipError_t err = prot::launch_deinterleave (e->d_inter.p, e->d_in.p, (int64_t)P, (int64_t)P, (int)nc, s);
	if (err != hipSuccess) {
		return hip_fail (err, "deinterleave launch");
	}
	rc = prot_sweep_dev (e, e->d_in.p, (int64_t)P, P, ang_start, ang_end, ang_stride, chn, nullptr,
	                     start ? PROT_F_STREAM_START : 0, s);
	if (rc) {
		return rc;
	}
	HIP_TRY (hipStreamSynchronize (s));
	return PROT_OK;
}

int prot_cli_analyze_file (prot_engine* e, const float* data, uint64_t n_frames, int ang_start, int ang_end,
                           int ang_stride, int chn)
{
	if (!e || (!data && n_frames) || e->info.kind != PROT_KIND_CLI) {
		return PROT_EINVAL;
	}
	HIP_TRY (hipSetDevice (e->device));
	const uint32_t nc = e->info.n_chn;
	const uint64_t P  = e->info.blksiz;
	CallStream    s (e, nullptr);
	(void)P;
	if (e->file_resident && n_frames > 0) {
		size_t free_b = 0, total_b = 0;
		const size_t need = (size_t)n_frames * nc * sizeof (float);
		const bool   have = e->file_key == data && e->file_n == n_frames && e->d_file.p != nullptr;
		if (have || (hipMemGetInfo (&free_b, &total_b) == hipSuccess && need <= total_b / 4 && 3 * need <= free_b)) {
			if (!have) {
				// upload once: chunks through the interleaved staging buffer, de-interleaved straight into the planar copy
				const uint64_t chunk = file_chunk_frames (e);
				e->file_stride       = (n_frames + 63) & ~(uint64_t)63;
				int            rc    = e->d_file.ensure ((size_t)e->file_stride * nc);
				if (rc == PROT_OK) {
					rc = e->d_inter.ensure (std::min<uint64_t> (chunk, n_frames) * nc);
				}
				if (rc) {
					return rc;
				}
				e->file_key = nullptr;
				for (uint64_t pos = 0; pos < n_frames; pos += chunk) {
					const uint64_t n = std::min (chunk, n_frames - pos);
					HIP_TRY (hipMemcpyAsync (e->d_inter.p, data + pos * nc, n * nc * sizeof (float), hipMemcpyHostToDevice, s));
					hipError_t err = prot::launch_deinterleave (e->d_inter.p, e->d_file.p + pos, (int64_t)e->file_stride, (int64_t)n, (int)nc, s);
					if (err != hipSuccess) {
						return hip_fail (err, "deinterleave launch");
					}
				}
				e->file_key = data;
				e->file_n   = n_frames;
			}
			return prot_sweep_dev (e, e->d_file.p, (int64_t)e->file_stride, n_frames, ang_start, ang_end, ang_stride, chn, nullptr,
			                       PROT_F_STREAM_START | PROT_F_STREAM_END, s);
		}
	}
	return analyze_range (e, data, 0, n_frames, true, true, ang_start, ang_end, ang_stride, chn);
}

int prot_cli_file_resident (prot_engine* e, int on)
{
	if (!e || e->info.kind != PROT_KIND_CLI) {
		return PROT_EINVAL;
	}
	HIP_TRY (hipSetDevice (e->device));
	e->file_resident = on != 0;
	if (!on) {
		(void)join_pipe (e, e->stream);
		HIP_TRY (sync_own (e));
		e->d_file.release ();
		e->file_key = nullptr;
		e->file_n   = 0;
	}
	return PROT_OK;
}

// ------------------------------------------------------------------ several GPUs
} // extern "C"

namespace prot_host {

// sweep frames [f0, f0 + nf) of an in-memory interleaved file on one engine
static int analyze_range_impl (prot_engine* e, const float* data, uint64_t f0, uint64_t nf, bool is_first, bool is_last,
                               int ang_start, int ang_end, int ang_stride, int chn, int phase, int seed_blocks);
int analyze_range (prot_engine* e, const float* data, uint64_t f0, uint64_t nf, bool is_first, bool is_last,
                   int ang_start, int ang_end, int ang_stride, int chn, int phase, int seed_blocks)
{
	const int rc = analyze_range_impl (e, data, f0, nf, is_first, is_last, ang_start, ang_end, ang_stride, chn, phase, seed_blocks);
	if (rc != PROT_OK) {
		// an error exit of the double-buffered path may leave an upload from the caller's `data` or kernels on the
		// staging buffers in flight: drain both streams before the caller frees the file or destroys the engine
		if (e->copy_stream) {
			(void)hipStreamSynchronize (e->copy_stream);
		}
		(void)hipStreamSynchronize (e->stream);
	}
	return rc;
}
static int analyze_range_impl (prot_engine* e, const float* data, uint64_t f0, uint64_t nf, bool is_first, bool is_last,
                               int ang_start, int ang_end, int ang_stride, int chn, int phase, int seed_blocks)
{
	HIP_TRY (hipSetDevice (e->device));
	const uint32_t nc = e->info.n_chn;
	CallStream    s (e, nullptr);
	const uint64_t chunk = file_chunk_frames (e);
	const uint64_t H     = e->info.history;
	const uint64_t cap   = std::max<uint64_t> (std::min<uint64_t> (chunk, std::max<uint64_t> (nf, 1)), H);
	const bool     dbl   = nf > chunk; // more than one chunk: upload of chunk k+1 under the sweep of chunk k
	int rc;
	if ((rc = e->d_inter.ensure (cap * nc)) || (rc = e->d_in.ensure (cap * nc)) || (rc = e->d_out.ensure (cap * nc)) ||
	    (dbl && ((rc = e->d_inter_b.ensure (cap * nc)) || (rc = e->d_in_b.ensure (cap * nc))))) {
		return rc;
	}
	if (dbl && !e->copy_stream) {
		HIP_TRY (hipStreamCreateWithFlags (&e->copy_stream, hipStreamNonBlocking));
		for (int i = 0; i < 2; ++i) {
			HIP_TRY (hipEventCreateWithFlags (&e->ev_copied[i], hipEventDisableTiming));
			HIP_TRY (hipEventCreateWithFlags (&e->ev_free[i], hipEventDisableTiming));
		}
	}
	if (!is_first && phase != 2) {
		// prime the input history with the H frames in front of the range: they only have to pass into the history (round 5: a copy
		// kernel instead of a Hilbert launch whose output was thrown away)
		const uint64_t h = std::min<uint64_t> (H, f0);
		HIP_TRY (hipMemcpyAsync (e->d_inter.p, data + (f0 - h) * nc, h * nc * sizeof (float), hipMemcpyHostToDevice, s));
		hipError_t err = prot::launch_deinterleave (e->d_inter.p, e->d_in.p, (int64_t)cap, (int64_t)h, (int)nc, s);
		if (err != hipSuccess) {
			return hip_fail (err, "deinterleave launch");
		}
		rc = hist_advance (e, e->d_in.p, (int64_t)cap, h, 0, (int)nc, s);
		if (rc) {
			return rc;
		}
		e->pos += h;
		HIP_TRY (hipStreamSynchronize (s));
	}
	if (phase == 1) {
		// the range's first chunk goes up now and stays where the main pass expects it; the bound is seeded from a sparse sample of it
		const uint64_t n = std::min (chunk, nf);
		e->seed_chunk_ready = false;
		if (n == 0) {
			return PROT_OK;
		}
		HIP_TRY (hipMemcpyAsync (e->d_inter.p, data + f0 * nc, n * nc * sizeof (float), hipMemcpyHostToDevice, s));
		hipError_t err = prot::launch_deinterleave (e->d_inter.p, e->d_in.p, (int64_t)cap, (int64_t)n, (int)nc, s);
		if (err != hipSuccess) {
			return hip_fail (err, "deinterleave launch");
		}
		const bool     last = n >= nf;
		const uint32_t fl   = (is_first ? PROT_F_STREAM_START : 0) | (last && is_last ? PROT_F_STREAM_END : 0);
		rc = prot_sweep_seed_dev (e, e->d_in.p, (int64_t)cap, n, ang_start, ang_end, ang_stride, chn, nullptr, fl, seed_blocks, s);
		if (rc) {
			return rc;
		}
		HIP_TRY (hipStreamSynchronize (s));
		e->seed_chunk_ready = true;
		return PROT_OK;
	}
	const bool have_first = phase == 2 && e->seed_chunk_ready;
	e->seed_chunk_ready   = false;
	uint64_t pos   = 0;
	bool     first = true;
	int      i     = 0;
	bool     used[2] = { false, false };
	do {
		const uint64_t n    = std::min (chunk, nf - pos);
		const bool     last = pos + n >= nf;
		const int      b    = dbl ? (i & 1) : 0;
		float*         di   = b ? e->d_inter_b.p : e->d_inter.p;
		float*         dn   = b ? e->d_in_b.p : e->d_in.p;
		hipStream_t    cs   = dbl ? e->copy_stream : s;
		if (n && !(have_first && i == 0)) { // (the seeding phase of a sharded analysis has put the first chunk there already)
			if (dbl && used[b]) {
				HIP_TRY (hipStreamWaitEvent (cs, e->ev_free[b], 0)); // the sweep that read this pair two chunks ago
			}
			HIP_TRY (hipMemcpyAsync (di, data + (f0 + pos) * nc, n * nc * sizeof (float), hipMemcpyHostToDevice, cs));
			hipError_t err = prot::launch_deinterleave (di, dn, (int64_t)cap, (int64_t)n, (int)nc, cs);
			if (err != hipSuccess) {
				return hip_fail (err, "deinterleave launch");
			}
			if (dbl) {
				HIP_TRY (hipEventRecord (e->ev_copied[b], cs));
				HIP_TRY (hipStreamWaitEvent (s, e->ev_copied[b], 0));
			}
		}
		// an empty file never sets `start` (cli/phase-rotate.cc:573-576)
		const uint32_t fl = (first && n && is_first ? PROT_F_STREAM_START : 0) | (last && is_last ? PROT_F_STREAM_END : 0);
		if (n || (last && is_last)) {
			rc = prot_sweep_dev (e, dn, (int64_t)cap, n, ang_start, ang_end, ang_stride, chn, nullptr, fl, s);
			if (rc) {
				return rc;
			}
		}
		if (dbl) {
			HIP_TRY (hipEventRecord (e->ev_free[b], s));
			used[b] = true;
		} else {
			HIP_TRY (hipStreamSynchronize (s)); // one pair of staging buffers: reused by the next chunk
		}
		pos += n;
		first = false;
		++i;
	} while (pos < nf);
	if (dbl) {
		HIP_TRY (hipStreamSynchronize (s));
	}
	return PROT_OK;
}

} // namespace prot_host

extern "C" {

int prot_cli_apply_file (prot_engine* e, const float* data, uint64_t n_frames, const int* angles, float* out)
{
	if (!e || !data || !angles || !out || e->info.kind != PROT_KIND_CLI) {
		return PROT_EINVAL;
	}
	if (n_frames == 0) {
		return PROT_OK;
	}
	HIP_TRY (hipSetDevice (e->device));
	const uint32_t nc  = e->info.n_chn;
	const uint64_t P   = e->info.blksiz;
	const uint64_t lat = P / 2; // cli/phase-rotate.cc:963
	CallStream    s (e, nullptr);
	const uint64_t chunk = file_chunk_frames (e);
	const uint64_t cap   = std::min<uint64_t> (chunk, n_frames) + lat;
	int rc;
	if ((rc = e->d_inter.ensure (cap * nc)) || (rc = e->d_in.ensure (cap * nc)) || (rc = e->d_out.ensure (cap * nc))) {
		return rc;
	}
	std::vector<float> rot (2 * (size_t)nc);
	for (uint32_t c = 0; c < nc; ++c) {
		const int a = ((angles[c] % PROT_MAXSAMPLE) + PROT_MAXSAMPLE) % PROT_MAXSAMPLE;
		rot[2 * c]     = g_lut[2 * a + 1];
		rot[2 * c + 1] = g_lut[2 * a];
	}
	const bool resident = e->file_resident && e->file_key == data && e->file_n == n_frames && e->d_file.p != nullptr;
	uint64_t   pos      = 0;
	while (pos < n_frames) {
		const uint64_t n     = std::min (chunk, n_frames - pos);
		const bool     last  = pos + n >= n_frames;
		const uint64_t n_out = n + (last ? lat : 0); // flush: latency frames of silence in (:993-1001)
		hipError_t err = hipSuccess;
		if (resident) { // the de-interleaved file is on the device already (prot_cli_file_resident): no upload
			rc = prot_apply_dev (e, e->d_file.p + pos, (int64_t)e->file_stride, n, e->d_out.p, (int64_t)cap, n_out, rot.data (), 0, s);
		} else {
			HIP_TRY (hipMemcpyAsync (e->d_inter.p, data + pos * nc, n * nc * sizeof (float), hipMemcpyHostToDevice, s));
			err = prot::launch_deinterleave (e->d_inter.p, e->d_in.p, (int64_t)cap, (int64_t)n, (int)nc, s);
			if (err != hipSuccess) {
				return hip_fail (err, "deinterleave launch");
			}
			rc = prot_apply_dev (e, e->d_in.p, (int64_t)cap, n, e->d_out.p, (int64_t)cap, n_out, rot.data (), 0, s);
		}
		if (rc) {
			return rc;
		}
		err = prot::launch_interleave (e->d_out.p, (int64_t)cap, e->d_inter.p, (int64_t)n_out, (int)nc, s);
		if (err != hipSuccess) {
			return hip_fail (err, "interleave launch");
		}
		// y[t] for t in [pos, pos + n_out) -> file frame t - lat
		const uint64_t skip = pos < lat ? std::min (lat - pos, n_out) : 0; // :963-966 off = latency on the first block(s)
		if (n_out > skip) {
			const uint64_t dst = pos + skip - lat;
			const uint64_t cnt = std::min (n_out - skip, n_frames - dst);
			HIP_TRY (hipMemcpyAsync (out + dst * nc, e->d_inter.p + skip * nc, cnt * nc * sizeof (float),
			                         hipMemcpyDeviceToHost, s));
		}
		HIP_TRY (hipStreamSynchronize (s));
		pos += n;
	}
	return PROT_OK;
}


} // extern "C"

