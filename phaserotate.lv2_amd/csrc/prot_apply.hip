// prot_apply.hip -- run geometry of a launch, the overlap-add / overlap-save launch (ola_launch), history, prot_apply_dev* (fp32 and fp16)
// (split from prot_capi.hip in round 6: same code; what the units share is in prot_engine.h)
#include "prot_engine.h"

namespace prot_host {

int pick_run_frames (const prot_engine* e, uint64_t n_out, int n_chn, int unit_frames, int slots_override)
{
	// One workgroup per run.  Long streams: fill every resident-workgroup slot of the
	// chip (CUs x 3, LDS-limited) a whole number of times so no partial last round is
	// left, with runs long enough (<= 96 pairs) that the halo re-computation stays ~2 %.
	// unit_frames: the transform block of the kernel that will run (default: this engine's own geometry)
	const int     pair  = unit_frames > 0 ? unit_frames : (int)e->info.run_align;
	const int64_t pairs = (int64_t)((n_out + pair - 1) / pair);
	int64_t       ppr   = 1;
	if (e->pairs_per_run_override > 0) {
		ppr = e->pairs_per_run_override;
	} else {
		const int64_t slots = std::max (1, slots_override > 0 ? slots_override : e->wg_slots);
		for (int k = 1; k <= 64; ++k) {
			const int64_t runs = std::max<int64_t> (1, slots * k / std::max (1, n_chn));
			ppr                = (pairs + runs - 1) / runs;
			if (ppr <= 96) {
				break;
			}
		}
	}
	ppr = std::min<int64_t> (std::max<int64_t> (ppr, 1), 1 << 20);
	return (int)std::min<int64_t> (ppr * pair, INT_MAX / 2);
}

// One-round launches of the polyphase kernel (every workgroup resident at once): run lengths by DISPATCH RANK, see
// olapoly_kernel.  `slots` = resident workgroups (CUs x 3 for fp32 samples, CUs x 4 for fp16 samples).
// measured (tools/ab_bench.py, PROT_RANK_W sweeps, headline / CLI 2048 / CLI 1024, ms per 2^27 frames, r2a kernel): equal
// runs 0.315-0.326 / 0.270 / 0.248-0.257; 1.3:1:0.75 0.312-0.318 / 0.249-0.252 / 0.233; 1.45:0.95:0.6 0.291-0.302 /
// 0.243-0.251 / 0.230-0.239; 1.6:0.9:0.5 0.291-0.293 / 0.258-0.265 / 0.247; 1.8:0.8:0.4 0.313-0.323 / 0.273-0.281 / 0.26
template <class P>
void rank_runs (P& p, int64_t blk, int n_chn, uint64_t n_out, int slots, int cus, bool half)
{
	if (cus <= 0 || slots % cus != 0) {
		return;
	}
	const int R = slots / cus;
	if (R < 2 || R > 4) {
		return;
	}
	// (the geometry test first: small launches -- every LV2 run() -- leave here without touching the environment)
	const int64_t runs = ((int64_t)n_out + p.run_frames - 1) / p.run_frames;
	if (runs * n_chn > slots || runs * n_chn <= (int64_t)(R - 1) * cus || p.run_frames / blk < 12) {
		return;
	}
	double      w[4] = { 1.45, 0.95, 0.6, 0.0 };
	if (R == 2) {
		w[0] = 1.2, w[1] = 0.8;
	}
	if (R == 4) {
		w[0] = 1.4, w[1] = 1.1, w[2] = 0.9, w[3] = 0.6; // fp16 headline / CLI 2048, ms per 2^27 frames: equal runs 0.247 / 0.238; 1.5:1.1:0.8:0.6 0.230 / 0.228;
		                                                 // 1.4:1.1:0.9:0.6 0.229 / 0.226; 1.6:1.1:0.8:0.5 0.227 / 0.235; 1.3:1.1:0.9:0.7 0.234 / 0.229 (PROT_RANK_W16)
	}
	const char* we = tuning_env (half ? "PROT_RANK_W16" : "PROT_RANK_W"); // tuning: "w0,w1,w2[,w3]" ("1,1,1": equal runs)
	if (we) {
		double v[4] = { 0, 0, 0, 0 };
		if (sscanf (we, "%lf,%lf,%lf,%lf", &v[0], &v[1], &v[2], &v[3]) >= R) {
			for (int k = 0; k < R; ++k) {
				w[k] = v[k];
			}
		}
	}
	bool equal = true;
	for (int k = 1; k < R; ++k) {
		equal = equal && w[k] == w[0];
	}
	if (equal) {
		return;
	}
	const int64_t total = ((int64_t)n_out + blk - 1) / blk; // blocks per channel
	double        wsum  = 0;
	for (int k = 0; k < R; ++k) {
		wsum += w[k];
	}
	int64_t nb[4] = { 0, 0, 0, 0 };
	for (int k = 0; k < R; ++k) {
		nb[k] = std::max<int64_t> (1, (int64_t)std::llround ((double)(p.run_frames / blk) * (double)R * w[k] / wsum));
	}
	// every channel must be covered: lengthen the LAST rank that occurs in the least-covered channel until it is (a channel
	// of three or more may consist of lower-rank workgroups only: lengthening rank R-1 alone would never cover it)
	for (int guard = 0;; ++guard) {
		int64_t worst = INT64_MAX;
		int     wy    = 0;
		for (int y = 0; y < n_chn; ++y) {
			int64_t cov = 0;
			for (int64_t x = 0; x < runs; ++x) {
				cov += nb[std::min<int64_t> (R - 1, (y * runs + x) / cus)];
			}
			if (cov < worst) {
				worst = cov;
				wy    = y;
			}
		}
		if (worst >= total) {
			break;
		}
		if (guard > (1 << 22)) { // cannot happen (every step raises the worst channel's coverage); equal runs if it does
			return;
		}
		const int64_t last_id = (int64_t)wy * runs + runs - 1;
		const int     rk      = (int)std::min<int64_t> (R - 1, last_id / cus);
		// all of that rank's runs in the channel grow together: jump by the whole deficit at once
		int64_t cnt = 0;
		for (int64_t x = 0; x < runs; ++x) {
			cnt += std::min<int64_t> (R - 1, ((int64_t)wy * runs + x) / cus) == rk;
		}
		nb[rk] += std::max<int64_t> (1, (total - worst + cnt - 1) / cnt);
	}
	p.rank_cu = cus;
	p.n_rank  = R;
	for (int k = 0; k < R; ++k) {
		p.rank_frames[k] = (int)(nb[k] * blk);
	}
}


int ola_launch (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_in, float* d_out,
                int64_t out_stride, uint64_t n_out, const float* d_rot, const float* d_rot_s,
                bool hilbert_only, bool stateless, int chn0, int n_chn, hipStream_t s, bool advance_hist,
                const float* h_rot, const uint32_t* pipe_wait_post, FusedSweep* fs, prot::SvServerBox* srv)
{
	prot::OlaParams p {};
	if (pipe_wait_post) { // pipelined stateful launch (prot_apply_dev): history handed over on the device
		p.hs_flag = e->d_pipe.p;
		p.hs_cnt  = e->d_pipe.p + 8 + (pipe_wait_post[1] & 3u);
		p.hs_wait = pipe_wait_post[0];
		p.hs_post = pipe_wait_post[1];
	}
	if (h_rot) { // steady rotation of up to kRotInline channels: by value in the kernel arguments (d_rot == nullptr)
		memcpy (p.rot_v, h_rot + 2 * chn0, 2 * sizeof (float) * (size_t)std::min (n_chn, prot::kRotInline));
	}
	p.in           = d_in + (int64_t)chn0 * in_stride;
	p.in_stride    = in_stride;
	p.n_in         = (int64_t)n_in;
	p.hist         = stateless ? nullptr : e->d_hist[e->hist_cur].p + (size_t)chn0 * e->info.history;
	p.hist_len     = (int)e->info.history;
	// all channels, something to launch: the first OLA launch also writes the next call's history
	// (other buffer of the pair) and the separate history kernel is skipped
	const bool fuse_hist = advance_hist && !stateless && n_out > 0 && chn0 == 0 && n_chn == (int)e->info.n_chn;
	p.hist_out     = fuse_hist ? e->d_hist[(e->hist_cur + 1) % prot_engine::kHist].p : nullptr;
	p.out          = d_out + (int64_t)chn0 * out_stride;
	p.out_stride   = out_stride;
	p.n_out        = (int64_t)n_out;
	p.tw           = e->d_tw.p;
	p.rot          = d_rot ? d_rot + 2 * chn0 : nullptr;
	p.rot_s        = d_rot_s ? d_rot_s + 2 * (int64_t)chn0 * (int64_t)n_out : nullptr;
	p.run_frames   = pick_run_frames (e, n_out, n_chn);
	p.dry_delay    = (int)e->info.dry_delay;
	p.hilbert_only = hilbert_only ? 1 : 0;
	// history bookkeeping of an advancing call: flip the buffer pair (fused) or run the history kernel
	auto finish = [&] () -> int {
		if (!advance_hist || stateless) {
			return PROT_OK;
		}
		if (fuse_hist) {
			e->hist_cur = (e->hist_cur + 1) % prot_engine::kHist;
			return PROT_OK;
		}
		return hist_advance (e, d_in, in_stride, n_in, chn0, n_chn, s);
	};
	if (srv && !(e->use_poly && e->use_save)) {
		return PROT_EUNSUPPORTED;
	}
	// The fused sweep (fs): the test rides in the (last) Hilbert launch -- one list per wave (4 per workgroup), long enough for
	// every frame of the workgroup's run to survive.  0: launch; 1: plan_only, nothing to launch; < 0: -status
	auto fs_prepare = [&] (int waves = 4) -> int { // waves per workgroup of the kernel that lists (8: the 8192-point kernel)
		const int64_t blk = (int64_t)e->info.run_align;
		if (fs->sample) {
			p.run_frames = (int)blk;
			p.run_stride = (int)(blk * fs->stride_blocks);
			p.rank_cu    = 0;
		}
		const int64_t unit   = p.run_stride > 0 ? p.run_stride : p.run_frames;
		const int64_t runs   = ((int64_t)n_out + unit - 1) / unit;
		int64_t       longest = p.run_frames;
		for (int k = 0; k < p.n_rank && p.rank_cu > 0; ++k) {
			longest = std::max<int64_t> (longest, p.rank_frames[k]);
		}
		fs->n_lists = (int)runs * waves;
		fs->cap     = ((longest / waves + 3) & ~(int64_t)3) + 4; // a wave's share of its workgroup's longest run
		if (fs->plan_only) {
			return 1;
		}
		// (the buffers were sized by the caller for BOTH passes before the first launch: growing one between the passes
		// would free memory a kernel of the pass before may still be reading)
		if (e->d_surv.cap < (size_t)fs->n_lists * fs->cap * n_chn + 2048 || e->d_surv_cnt.cap < 2 * (size_t)fs->n_lists * n_chn ||
		    e->d_desc.cap < ((size_t)fs->n_lists + (size_t)(n_out / 1024) + 8) * n_chn || e->d_nsteps.cap < (size_t)n_chn) {
			snprintf (g_err, sizeof (g_err), "fused sweep: list buffers not sized");
			return -PROT_EINVAL;
		}
		p.surv        = e->d_surv.p;
		p.surv_cap    = fs->cap;
		p.surv_cnt    = e->d_surv_cnt.p;
		p.sw_peaks    = fs->peaks + (size_t)chn0 * PROT_MAXSAMPLE;
		p.sw_angles   = fs->angles;
		p.sw_n_angles = fs->n_angles;
		p.sw_raw_slot = fs->raw_slot;
		p.sw_raw_tail = fs->raw_tail;
		p.sw_thr_zero = fs->sample ? 1 : 0;
		p.sw_t_begin  = fs->t_begin;
		p.sw_dry_from = fs->dry_from;
		return 0;
	};
	if (e->use_p2 && e->d_Hp.p != nullptr && e->info.kind == PROT_KIND_LV2) {
		// short launch of a direct-form LV2 engine: blocks per workgroup (k) against halo blocks -- direct (k + 2),
		// polyphase 1.07 (k + 1): polyphase below k ~ 13
		const int64_t k = p.run_frames / 4096;
		if (k < 12) {
			p.H          = e->d_Hp.p;
			p.conv_delay = (int)e->info.conv_delay;
			p.accumulate = 0;
			hipError_t err = prot::launch_ola_poly (p, 1, 8, n_chn, s);
			if (err != hipSuccess) {
				return hip_fail (err, "polyphase ola kernel launch (short launch)");
			}
			return finish ();
		}
	}
	if (e->use_save16 && pipe_wait_post == nullptr && srv == nullptr) {
		// CLI blksiz 16384: 16384-point blocks, 1024-thread workgroups, one per CU (equal runs: one workgroup per CU has no dispatch rank)
		const int blk16 = 2048 * e->sv16_rows;
		p.run_frames = pick_run_frames (e, n_out, n_chn, blk16, e->cus);
		p.rank_cu    = 0;
		p.sv         = e->d_sv16.p;
		p.sv_gain    = e->sv16_gain;
		p.sv_flags   = e->sv_flags;
		p.conv_delay = (int)e->info.conv_delay;
		p.accumulate = 0;
		if (fs) {
			if (!hilbert_only) {
				return PROT_EUNSUPPORTED;
			}
			const int st = fs_prepare (16);
			if (st != 0) {
				return st > 0 ? PROT_OK : -st;
			}
		}
		hipError_t err = prot::launch_ola_save16k (p, e->sv16_rows, n_chn, s);
		if (err != hipSuccess) {
			return hip_fail (err, "overlap-save (16384-point) ola kernel launch");
		}
		return finish ();
	}
	if (e->use_poly) {
		const uint32_t np = std::min<uint32_t> (2, e->info.partitions); // partitions of g per launch
		// (two workgroups per CU, two partitions in one launch: 1.2 : 0.8, -4...6 % at 8192 taps; the accumulating launches
		// of longer FIRs lose 6 % with it and keep equal runs)
		if (e->info.partitions == np) {
			rank_runs (p, (int64_t)e->info.run_align, n_chn, n_out, e->wg_slots, e->cus, false);
		}
		if (e->use_save8 && pipe_wait_post == nullptr && (e->sv8_always || (uint64_t)n_out * (uint64_t)n_chn >= ((uint64_t)1 << 24))) {
			// a long launch: 8192-point blocks, 512-thread workgroups, two per CU (run lengths 1.2 : 0.8 by dispatch rank)
			const int blk8 = 1024 * e->sv8_rows, slots8 = e->cus * 2;
			p.run_frames   = pick_run_frames (e, n_out, n_chn, blk8, slots8);
			p.rank_cu      = 0;
			rank_runs (p, (int64_t)blk8, n_chn, n_out, slots8, e->cus, false);
			p.sv         = e->d_sv8.p;
			p.sv_gain    = e->sv8_gain;
			p.sv_flags   = e->sv_flags;
			p.conv_delay = (int)e->info.conv_delay;
			p.accumulate = 0;
			if (fs) { // (CLI blksiz 8192: the fused sweep on 8192-point blocks, eight lists per workgroup)
				if ((e->sv8_rows != 8 && e->sv8_rows != 12) || !hilbert_only) {
					return PROT_EUNSUPPORTED;
				}
				const int st = fs_prepare (8);
				if (st != 0) {
					return st > 0 ? PROT_OK : -st;
				}
			}
			// Dynamic blocks (round 5): when the stream is long enough for every resident workgroup to transform several blocks, the
			// whole in-range blocks are CLAIMED one at a time instead of dealt out as runs (prot_olasave8k.hip: the static plan leaves
			// the CUs half empty for the last fifth of a launch).  Block b covers output frames [b BLK, (b + 1) BLK) and reads input
			// frames a0 - 1 ... a0 + 16383, a0 = b BLK - off: whole and in range for lo <= b < hi.
			if (!fs && !e->dyn_off && p.rot_s == nullptr && !(e->sv_flags & 1) && e->d_dyn.p != nullptr &&
			    ((reinterpret_cast<uintptr_t> (p.in) | reinterpret_cast<uintptr_t> (p.out)) & 7) == 0 && ((in_stride | out_stride) & 1) == 0) {
				const int64_t nt = 16 - e->sv8_rows, off = (int64_t)p.dry_delay + 1024 * (nt / 2) + ((nt & 1) ? 512 : 0);
				const int64_t lo = (1 + off + blk8 - 1) / blk8;
				const int64_t hi = std::min<int64_t> ((int64_t)n_out / blk8, (int64_t)n_in - 16384 + off >= 0 ? ((int64_t)n_in - 16384 + off) / blk8 + 1 : 0);
				const int     gx = std::max (1, slots8 / std::max (1, n_chn));
				if (hi - lo >= 2 * (int64_t)gx && hi - lo < ((int64_t)1 << 30) && ((int64_t)n_out + blk8 - 1) / blk8 - (hi - lo) <= 64) {
					p.dyn_ctr = e->d_dyn.p + (size_t)(e->dyn_seq++ % prot_engine::kDynSets) * e->info.n_chn * prot::kDynWords + (size_t)chn0 * prot::kDynWords;
					p.dyn_lo  = lo;
					p.dyn_nb  = hi - lo;
					p.rank_cu = 0;
					hipError_t err = prot::launch_ola_save8k_dyn (p, e->sv8_rows, n_chn, gx, s);
					if (err != hipSuccess) {
						return hip_fail (err, "overlap-save (8192-point, dynamic blocks) ola kernel launch");
					}
					return finish ();
				}
			}
			hipError_t err = prot::launch_ola_save8k (p, e->sv8_rows, n_chn, s);
			if (err != hipSuccess) {
				return hip_fail (err, "overlap-save (8192-point) ola kernel launch");
			}
			return finish ();
		}
		if (e->use_save) { // one partition of an antisymmetric g: overlap-save, no spectrum table, four workgroups per CU
			p.sv         = e->d_sv.p;
			p.sv_gain    = e->sv_gain;
			p.sv_flags   = e->sv_flags;
			p.conv_delay = (int)e->info.conv_delay;
			p.accumulate = 0;
			if (fs) {
				const int st = fs_prepare ();
				if (st != 0) {
					return st > 0 ? PROT_OK : -st;
				}
			}
			if (srv) { // the run() server takes the call: its parameters go into the mailbox, nothing is launched
				if (prot::post_ola_save_server (p, e->hopseg / 2, e->sv_rows, n_chn, srv) != hipSuccess) {
					return PROT_EUNSUPPORTED; // (more than one block per channel, ...: the caller launches it)
				}
				return finish ();
			}
			// a launch of fewer blocks than the device has CUs: every block is a lone workgroup's dependent chain -- the 512-thread
			// block walks it in half the time (prot_olasave512.hip; not bit-identical to the 256-thread block: the route depends on
			// the launch's shape only; pipelined launches hand their history over inside it as they do inside olasave_kernel)
			if (!fs && test_opt (PROT_TEST_NO_SHORT_CHAIN) == 0 && prot::ola_save512_takes (p, e->hopseg / 2, e->sv_rows, n_chn, e->cus)) {
				hipError_t err5 = prot::launch_ola_save512 (p, e->hopseg / 2, n_chn, s);
				if (err5 != hipSuccess) {
					return hip_fail (err5, "overlap-save (512-thread block) ola kernel launch");
				}
				return finish ();
			}
			hipError_t err = prot::launch_ola_save (p, e->hopseg / 2, e->sv_rows, n_chn, s);
			if (err != hipSuccess) {
				return hip_fail (err, "overlap-save ola kernel launch");
			}
			return finish ();
		}
		const uint32_t n_launch = e->info.partitions / np;
		if (fs) { // (long FIRs, CLI blksiz 16384 / 32768: the LAST accumulating launch tests instead of storing)
			if (!hilbert_only || !((n_launch >= 2 && np == 2) || (n_launch == 1 && np == 1 && e->hopseg == 28))) {
				return PROT_EUNSUPPORTED;
			}
			const int st = fs_prepare ();
			if (st != 0) {
				return st > 0 ? PROT_OK : -st;
			}
		}
		float2* const fs_surv = p.surv;
		for (uint32_t l = 0; l < n_launch; ++l) {
			p.H          = e->d_H.p + (size_t)(np * l) * e->info.fftlen;
			p.conv_delay = (int)(e->info.conv_delay + np * l * e->info.fftlen); // a partition of g spans 4096 frames
			p.accumulate = l > 0;
			p.surv       = l + 1 == n_launch ? fs_surv : nullptr;
			hipError_t err = prot::launch_ola_poly (p, (int)np, e->hopseg / 2, n_chn, s);
			if (err != hipSuccess) {
				return hip_fail (err, "polyphase ola kernel launch");
			}
			p.hist_out = nullptr;
		}
		return finish ();
	}
	if (e->use_p2) {
		for (uint32_t pair = 0; pair < e->info.partitions / 2; ++pair) {
			p.H          = e->d_H.p + (size_t)(2 * pair) * e->info.fftlen;
			p.conv_delay = (int)(e->info.conv_delay + 2 * pair * e->part_len);
			p.accumulate = pair > 0;
			hipError_t err = prot::launch_ola4096_p2 (p, n_chn, s);
			if (err != hipSuccess) {
				return hip_fail (err, "ola p2 kernel launch");
			}
			p.hist_out = nullptr;
		}
		return finish ();
	}
	for (uint32_t sp = 0; sp < e->info.partitions; ++sp) {
		p.H          = e->d_H.p + (size_t)sp * e->info.fftlen;
		p.conv_delay = (int)(e->info.conv_delay + sp * e->part_len);
		p.accumulate = sp > 0;
		hipError_t err = prot::launch_ola4096 (p, e->hopseg, n_chn, s);
		if (err != hipSuccess) {
			return hip_fail (err, "ola kernel launch");
		}
		p.hist_out = nullptr;
	}
	return finish ();
}

// advance the per-channel input history by one chunk (channels [chn0, chn0+n_chn))
int hist_advance (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_in, int chn0,
                  int n_chn, hipStream_t s)
{
	const int    H   = (int)e->info.history;
	const int    cur = e->hist_cur;
	const float* hin = e->d_hist[cur].p;
	float*       hout = e->d_hist[(cur + 1) % prot_engine::kHist].p;
	if (n_chn != (int)e->info.n_chn) {
		// untouched channels keep their history: carry it over to the other buffer
		HIP_TRY (hipMemcpyAsync (hout, hin, (size_t)H * e->info.n_chn * sizeof (float),
		                         hipMemcpyDeviceToDevice, s));
	}
	hipError_t err = prot::launch_hist_update (d_in + (int64_t)chn0 * in_stride, in_stride, (int64_t)n_in,
	                                           hin + (size_t)chn0 * H, hout + (size_t)chn0 * H, H, n_chn, s);
	if (err != hipSuccess) {
		return hip_fail (err, "hist update launch");
	}
	e->hist_cur = (cur + 1) % prot_engine::kHist;
	return PROT_OK;
}

// device calls read the input of neighbouring runs (halo blocks) while other workgroups store output:
// input and output must not share frames.  Exact per-channel interval test (skipped beyond 64 channels).
bool io_overlap_bytes (const void* in, int64_t in_stride, uint64_t n_in, const void* out, int64_t out_stride, uint64_t n_out, int nc,
                       size_t esz)
{
	if (nc > 64 || n_in == 0 || n_out == 0) {
		return false;
	}
	for (int ci = 0; ci < nc; ++ci) {
		const uintptr_t a0 = (uintptr_t)in + (uintptr_t)((int64_t)ci * in_stride * (int64_t)esz), a1 = a0 + n_in * esz;
		for (int co = 0; co < nc; ++co) {
			const uintptr_t b0 = (uintptr_t)out + (uintptr_t)((int64_t)co * out_stride * (int64_t)esz), b1 = b0 + n_out * esz;
			if (a0 < b1 && b0 < a1) {
				return true;
			}
		}
	}
	return false;
}
bool io_overlap (const float* in, int64_t in_stride, uint64_t n_in, const float* out, int64_t out_stride, uint64_t n_out, int nc)
{
	return io_overlap_bytes (in, in_stride, n_in, out, out_stride, n_out, nc, sizeof (float));
}

// pipelined launches in flight: everything that follows on stream s is ordered behind them
int join_pipe (prot_engine* e, hipStream_t s)
{
	if (e->pipe_dirty) {
		for (int i = 0; i < prot_engine::kPipe; ++i) {
			HIP_TRY (hipEventRecord (e->pipe_ev[i], e->pipe_stream[i]));
			if (s != e->pipe_stream[i]) {
				HIP_TRY (hipStreamWaitEvent (s, e->pipe_ev[i], 0));
			}
			if (s != e->stream && e->stream != e->pipe_stream[i]) {
				HIP_TRY (hipStreamWaitEvent (e->stream, e->pipe_ev[i], 0));
			}
		}
		e->pipe_dirty = false;
		e->pipe_q     = 0;
	}
	return PROT_OK;
}

// ---- the run() server: at most ONE resident kernel per process (a resident kernel holds its hardware queue, and the

} // namespace prot_host

extern "C" {

int prot_apply_dev (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_in, float* d_out,
                    int64_t out_stride, uint64_t n_out, const float* rot, uint32_t flags, void* stream)
{
	if (!e || !d_in || !d_out || (!rot && !(flags & PROT_F_HILBERT))) {
		return PROT_EINVAL;
	}
	if (io_overlap (d_in, in_stride, n_in, d_out, out_stride, n_out, (int)e->info.n_chn)) {
		snprintf (g_err, sizeof (g_err), "prot_apply_dev: d_out overlaps d_in (in-place processing is not supported on device streams)");
		return PROT_EINVAL;
	}
	HIP_TRY (hipSetDevice (e->device));
	const int   nc = (int)e->info.n_chn;
	const bool inl = rot && nc <= prot::kRotInline; // {ca, sa} per channel as kernel arguments: no copy in front of the launch
	// Pipelined route (PROT_CFG_PIPELINE engines, the engine's own streams): a stateful launch needs nothing of the launch
	// before it but the history, i.e. the tail of that launch's INPUT -- so consecutive launches rotate over three
	// streams and the kernel hands the history over on the device (prot_olasave.hip).  Only launches that are small against
	// the device (a waiting workgroup must never hold a slot the launch it waits for needs) and long enough that the new
	// history is a plain copy of input frames; everything else joins the two streams first (pick_stream).
	if (e->pipeline && stream == nullptr && inl && !(flags & (PROT_F_STATELESS | PROT_F_HILBERT)) && n_out > 0 && n_out <= n_in &&
	    n_in >= e->info.history &&
	    ((int64_t)n_out + e->info.run_align - 1) / (int64_t)e->info.run_align * nc <= e->wg_slots / (2 * prot_engine::kPipe)) {
		hipStream_t ps = e->pipe_stream[e->pipe_q % prot_engine::kPipe];
		uint32_t    wp[2];
		if (e->last_foreign) { // (this route bypasses pick_stream -- a call on a caller's stream, and a reset's zeroing behind it)
			if (mark_last (e)) {
				HIP_TRY (hipStreamWaitEvent (e->stream, e->ev_last, 0));
			}
			e->last_foreign   = false;
			e->last_stream_id = nullptr;
		}
		if (e->reset_pending) {
			if (e->reset_stream != e->stream) {
				HIP_TRY (hipStreamWaitEvent (e->stream, e->ev_reset, 0));
			}
			e->reset_pending = false;
		}
		if (!e->pipe_dirty) { // first of a series: behind whatever the engine's stream holds; its history came by stream order
			HIP_TRY (hipEventRecord (e->pipe_ev[prot_engine::kPipe], e->stream));
			for (int i = 1; i < prot_engine::kPipe; ++i) {
				HIP_TRY (hipStreamWaitEvent (e->pipe_stream[i], e->pipe_ev[prot_engine::kPipe], 0));
			}
			wp[0] = 0;
		} else {
			wp[0] = e->pipe_gen; // what the launch before this one publishes
		}
		const uint32_t gen_before = e->pipe_gen;
		if (++e->pipe_gen == 0u) {
			++e->pipe_gen; // 0 means "first of a series, wait for nobody" (hs_wait): never a generation's number
		}
		wp[1]         = e->pipe_gen;
		e->pipe_dirty = true;
		e->pipe_used  = true;
		e->pipe_q++;
		int rc = ola_launch (e, d_in, in_stride, n_in, d_out, out_stride, n_out, nullptr, nullptr, false, false, 0, nc, ps,
		                     /*advance_hist*/ true, rot, wp);
		if (rc == PROT_OK) {
			e->pos += n_in;
		} else {
			e->pipe_gen = gen_before; // nothing was launched: the generation must not be waited for by the next launch
		}
		return rc;
	}
	CallStream s (e, stream);
	if (rot && !inl) {
		int rcs = stage_small (e, e->d_rot.p, rot, 2 * (size_t)nc * sizeof (float), s);
		if (rcs) {
			return rcs;
		}
	}
	const bool stateless = flags & PROT_F_STATELESS;
	int rc = ola_launch (e, d_in, in_stride, n_in, d_out, out_stride, n_out, inl ? nullptr : e->d_rot.p, nullptr,
	                     flags & PROT_F_HILBERT, stateless, 0, nc, s, /*advance_hist*/ true, inl ? rot : nullptr);
	if (rc) {
		return rc;
	}
	if (!stateless) {
		e->pos += n_in;
	}
	return rc;
}

int prot_apply_dev_ramp (prot_engine* e, const float* d_in, int64_t in_stride, uint64_t n_in, float* d_out,
                         int64_t out_stride, uint64_t n_out, const float* d_rot, uint32_t flags, void* stream)
{
	if (!e || !d_in || !d_out || !d_rot) {
		return PROT_EINVAL;
	}
	if (io_overlap (d_in, in_stride, n_in, d_out, out_stride, n_out, (int)e->info.n_chn)) {
		snprintf (g_err, sizeof (g_err), "prot_apply_dev_ramp: d_out overlaps d_in (in-place processing is not supported on device streams)");
		return PROT_EINVAL;
	}
	HIP_TRY (hipSetDevice (e->device));
	CallStream  s (e, stream);
	const int   nc        = (int)e->info.n_chn;
	const bool  stateless = flags & PROT_F_STATELESS;
	int rc = ola_launch (e, d_in, in_stride, n_in, d_out, out_stride, n_out, nullptr, d_rot, false, stateless, 0, nc, s,
	                     /*advance_hist*/ true);
	if (rc) {
		return rc;
	}
	if (!stateless) {
		e->pos += n_in;
	}
	return rc;
}

// fp16 sample path (BASELINE config 5): half samples in HBM, packed-half butterflies, fp32 overlap-add + rotation.
// Polyphase engines carry history (its own, in half precision) and take per-frame rotations like the fp32 path; FIRs of more
// than 8192 taps (CLI blksiz 16384, 32768: four / eight partitions of the odd taps) take two / four launches of two partitions
// each, the later ones adding their share to the half-precision output the earlier ones stored -- one more rounding to half per
// launch (round 4: measured against the oracle inside the path's stated tolerance, tests/test_gpu_fp16.py; refused before).  The
// direct-form cross-check engines (PROT_CFG_*) only offer the stateless steady call.
static int apply_f16 (prot_engine* e, const void* d_in, int64_t in_stride, uint64_t n_in, void* d_out, int64_t out_stride,
                      uint64_t n_out, const float* rot, const float* d_rot_s, uint32_t flags, void* stream)
{
	if (!e || !d_in || !d_out || (!rot && !d_rot_s)) {
		return PROT_EINVAL;
	}
	const bool stateless = flags & PROT_F_STATELESS;
	const bool f16_poly  = e->use_poly && (e->info.partitions <= 2 || e->info.partitions % 2 == 0);
	const bool f16_p2    = e->use_p2 && e->info.partitions == 2;
	const bool f16_one   = !e->use_poly && !e->use_p2 && e->info.partitions == 1;
	if (!f16_poly && !f16_p2 && !f16_one) {
		snprintf (g_err, sizeof (g_err), "fp16 sample path: this engine form has no half-precision kernel");
		return PROT_EUNSUPPORTED;
	}
	if (!f16_poly && (!stateless || d_rot_s)) {
		snprintf (g_err, sizeof (g_err), "fp16 sample path: the direct-form engines offer the stateless, steady call only");
		return PROT_EUNSUPPORTED;
	}
	// (the same rule as prot_apply_dev: workgroups read neighbouring halo blocks while others store)
	if (io_overlap_bytes (d_in, in_stride, n_in, d_out, out_stride, n_out, (int)e->info.n_chn, 2)) {
		snprintf (g_err, sizeof (g_err), "prot_apply_dev_f16: d_out overlaps d_in (in-place processing is not supported on device streams)");
		return PROT_EINVAL;
	}
	HIP_TRY (hipSetDevice (e->device));
	CallStream  s (e, stream);
	const int   nc = (int)e->info.n_chn;
	const bool inl = rot && nc <= prot::kRotInline;
	if (rot && !inl) {
		int rcs = stage_small (e, e->d_rot.p, rot, 2 * (size_t)nc * sizeof (float), s);
		if (rcs) {
			return rcs;
		}
	}
	if (!stateless && n_out == 0) {
		// nothing to emit, input consumed all the same: the half-precision history advances by n_in frames
		if (n_in > 0) {
			hipError_t herr = prot::launch_hist_update_f16 ((const _Float16*)d_in, in_stride, (int64_t)n_in,
			                                                (const _Float16*)e->d_hist16[e->hist16_cur].p,
			                                                (_Float16*)e->d_hist16[1 - e->hist16_cur].p, (int)e->info.history, nc, s);
			if (herr != hipSuccess) {
				return hip_fail (herr, "fp16 history update launch");
			}
			e->hist16_cur = 1 - e->hist16_cur;
			e->pos += n_in;
		}
		return PROT_OK;
	}
	prot::OlaParamsH p {};
	p.in         = (const _Float16*)d_in;
	p.in_stride  = in_stride;
	p.n_in       = (int64_t)n_in;
	p.out        = (_Float16*)d_out;
	p.out_stride = out_stride;
	p.n_out      = (int64_t)n_out;
	p.H          = e->d_H.p;
	p.tw         = e->d_tw.p;
	p.rot        = inl ? nullptr : e->d_rot.p;
	if (inl) {
		memcpy (p.rot_v, rot, 2 * sizeof (float) * (size_t)nc);
	}
	p.rot_s      = d_rot_s;
	// fp16 samples, one partition: 104-128 VGPRs and 26 KiB of LDS per workgroup -- FOUR per CU (measured 0.269 -> 0.252 ms
	// at the headline shape with equal runs; the fp32 kernel cannot: 158 VGPRs, 53 KiB)
	static const bool slots_forced = tuning_env ("PROT_WG_PER_CU") != nullptr; // tuning
	// (two partitions per launch: 128-158 VGPRs, 34 KiB -- three per CU where the fp32 kernel has two; 0.367 -> 0.336 ms)
	const uint32_t np16    = std::min<uint32_t> (2, e->info.partitions);
	const uint32_t n_l16   = f16_poly ? e->info.partitions / np16 : 1;
	p.conv_delay = (int)e->info.conv_delay;
	p.dry_delay  = (int)e->info.dry_delay;
	// one partition, steady rotation: the radix-16 passes on the matrix core (prot_f16mfma.hip, round 6; three workgroups per CU)
	const bool mfma16 = f16_poly && n_l16 == 1 && np16 == 1 && test_opt (PROT_TEST_F16_NO_MFMA) == 0 && prot::mfma_f16_takes (p, e->hopseg / 2);
	const int slots16 = (!f16_poly || slots_forced) ? e->wg_slots : e->cus * (mfma16 ? prot::mfma_f16_wg_per_cu () : e->info.partitions == 1 ? 4 : 3);
	p.run_frames = pick_run_frames (e, n_out, nc, 0, slots16); // the engine's own transform block: the fp16 kernels share its form
	p.hist_len   = (int)e->info.history;
	const bool carry = !stateless && n_out > 0;
	if (!stateless) {
		p.hist     = (const _Float16*)e->d_hist16[e->hist16_cur].p;
		p.hist_out = carry ? (_Float16*)e->d_hist16[1 - e->hist16_cur].p : nullptr;
	}
	if (f16_poly && n_l16 == 1) {
		rank_runs (p, (int64_t)e->info.run_align, nc, n_out, slots16, e->cus, true);
	}
	hipError_t err = hipSuccess;
	if (f16_poly) {
		for (uint32_t l = 0; l < n_l16 && err == hipSuccess; ++l) {
			p.H          = e->d_H.p + (size_t)(np16 * l) * e->info.fftlen;
			p.conv_delay = (int)(e->info.conv_delay + np16 * l * e->info.fftlen); // a partition of g spans 4096 frames
			p.accumulate = l > 0;
			err          = mfma16 ? prot::launch_ola_poly_f16_mfma (p, e->hopseg / 2, nc, s) : prot::launch_ola_poly_f16 (p, (int)np16, e->hopseg / 2, nc, s);
			p.hist_out   = nullptr;
		}
	} else {
		err = f16_p2 ? prot::launch_ola4096_p2_f16 (p, nc, s) : prot::launch_ola4096_f16 (p, e->hopseg, nc, s);
	}
	if (err != hipSuccess) {
		return hip_fail (err, "fp16 ola kernel launch");
	}
	if (carry) {
		e->hist16_cur = 1 - e->hist16_cur;
		e->pos += n_in;
	}
	return PROT_OK;
}

int prot_apply_dev_f16 (prot_engine* e, const void* d_in, int64_t in_stride, uint64_t n_in, void* d_out,
                        int64_t out_stride, uint64_t n_out, const float* rot, uint32_t flags, void* stream)
{
	return rot ? apply_f16 (e, d_in, in_stride, n_in, d_out, out_stride, n_out, rot, nullptr, flags, stream) : PROT_EINVAL;
}

int prot_apply_dev_f16_ramp (prot_engine* e, const void* d_in, int64_t in_stride, uint64_t n_in, void* d_out,
                             int64_t out_stride, uint64_t n_out, const float* d_rot, uint32_t flags, void* stream)
{
	return d_rot ? apply_f16 (e, d_in, in_stride, n_in, d_out, out_stride, n_out, nullptr, d_rot, flags, stream) : PROT_EINVAL;
}

// seed_blocks > 0: ONLY the seeding pass of the fused route over about that many blocks of the call (nothing of the engine's state
// moves; a call the fused route does not take seeds nothing and returns PROT_OK)
} // extern "C"

