/*
 * prot_oracle.h -- CPU restatement of the phaserotate.lv2 hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may link or load it; the
 * product (phaserotate.lv2_amd/, the C-ABI library, the LV2 plugin and the
 * CLI) never does.
 *
 * PARITY STATUS: "parity unpinned" against reference *outputs*.
 *   The reference cannot be built in this image: src/phaserotate.c and
 *   cli/phase-rotate.cc need <fftw3.h>/libfftw3f, <sndfile.h>/libsndfile and
 *   the LV2 SDK headers, none of which are installed, and the reference ships
 *   no tests, golden vectors or fixtures.  FFTW (module fftw3f, version not
 *   pinned by the reference: Makefile:114-116, cli/Makefile:17-19) is
 *   restated here from its published definition: r2c/c2r are unnormalised
 *   DFTs of N/2+1 bins, c2r ignores Im(DC) and Im(Nyquist).
 *   What IS pinned: (1) cli/dsp_peak_calc.h compiles on its own and is built
 *   unmodified into oracle/_ref/libref_peak.so -- orc_compute_peak is checked
 *   bit-exactly against it; (2) first-principles known answers (closed-form
 *   FIR, latency values, float64 LTI identities) and the behaviours SURVEY.md
 *   recorded from the reference (Q1..Q10); see tests/test_oracle_*.py.
 *
 * Every function cites the reference file:line it restates
 * (paths relative to /root/reference).
 */
#ifndef PROT_ORACLE_H
#define PROT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_SUBSAMPLE 2                     /* cli/phase-rotate.cc:38 */
#define ORC_MAXSAMPLE (180 * ORC_SUBSAMPLE) /* cli/phase-rotate.cc:39 */
#define ORC_MAX_CHANNELS 2                  /* src/phaserotate.h:97   */

/* ---- FFTW-semantics transforms (third-party dependency, restated) ------- */
/* fftwf_plan_dft_r2c_1d + execute: out[k] = sum_j in[j] e^{-2 pi i jk/n},
 * k = 0..n/2, interleaved re,im.  n = 2^a * 3^b, even, >= 4. */
void orc_fft_r2c (int n, const float* in, float* out);
/* fftwf_plan_dft_c2r_1d + execute: out[j] = sum over the Hermitian-extended
 * spectrum, unnormalised; Im(in[0]) and Im(in[n/2]) are ignored. `in` is
 * preserved (FFTW destroys it; nothing in the reference relies on that). */
void orc_fft_c2r (int n, const float* in, float* out);
/* CPU-baseline leg only (bench.py): route the two calls above through the FFT library the reference links --
 * dlopen of libfftw3f.so.3, else of libmkl_rt.so (MKL's FFTW3 interface).  Returns 0 = none found (own FFT stays),
 * 1 = FFTW, 2 = MKL.  Off by default: tests and parity always run on the own FFT. */
int         orc_fft_use_external (int on);
const char* orc_fft_backend_name (void);
/* same transforms in double (tolerance accounting only) */
void orc_fft_r2c_f64 (int n, const double* in, double* out);
void orc_fft_c2r_f64 (int n, const double* in, double* out);

/* ---- peak (cli/dsp_peak_calc.h:311-325, cli/phase-rotate.cc:85-96) ------ */
float orc_compute_peak (const float* buf, uint32_t n_samples, float current);

/* ---- SinCosLut (cli/phase-rotate.cc:41-74) ------------------------------ */
void orc_sincos_lut (int a, float* s, float* c);

/* ---- FIR design --------------------------------------------------------- */
/* Hilbert FIR of n taps: spectrum j(-1)^k -> c2r_n -> * (0.5/n)(1-cos 2 pi i/n)
 * (src/phaserotate.c:374-391 with n = firlen; cli/phase-rotate.cc:144-161
 * with n = _parsiz).  Window in double, product rounded to float as there. */
void orc_fir_design (int n, float* fir);
/* closed form of the same filter in double */
void orc_fir_closed_form (int n, double* fir);

/* ---- CLI engine: PhaseRotateProc (cli/phase-rotate.cc:125-245) ---------- */
typedef struct orc_cli_proc orc_cli_proc;
orc_cli_proc* orc_cli_proc_new (uint32_t blksiz);
void          orc_cli_proc_free (orc_cli_proc*);
uint32_t      orc_cli_proc_parsiz (const orc_cli_proc*);
void orc_cli_proc_hilbert (orc_cli_proc*, const float* in, float* out, float* o_out); /* :181-212 */
void orc_cli_proc_rotate (orc_cli_proc*, const float* in, float* out, int angle);    /* :214-225 */
void orc_cli_proc_process (orc_cli_proc*, const float* in, float* out, float* o_out, int angle); /* :227-232 */
const float* orc_cli_proc_ffir (const orc_cli_proc*); /* _ffir_data, parsiz+1 complex */

/* ---- CLI engine: PhaseRotate (cli/phase-rotate.cc:251-485) -------------- */
typedef struct orc_cli orc_cli;
orc_cli* orc_cli_new (uint32_t blksiz, uint32_t n_chn);
void     orc_cli_free (orc_cli*);
void     orc_cli_reset (orc_cli*);                                   /* :355-366 */
void     orc_cli_analyze (orc_cli*, const float* in, int ang_start, int ang_end,
                          int ang_stride, int chn, int start);      /* :431-444, :388-429 */
void     orc_cli_apply (orc_cli*, float* buf, const int* angles);    /* :467-485, :446-465 */
float    orc_cli_peak (const orc_cli*, int c, int a);                /* :275-285 */
float    orc_cli_peak_all (const orc_cli*, int a);                   /* :287-299 */
/* analyze_file on an in-memory interleaved float file (cli/phase-rotate.cc:565-587) */
void     orc_cli_analyze_file (orc_cli*, const float* data, uint64_t n_frames,
                               int ang_start, int ang_end, int ang_stride, int chn);

/* main()'s minimum search (cli/phase-rotate.cc:784-929) on an in-memory file.
 * Outputs per channel: min_angle (half-degree index, possibly negative after
 * the phase-distance unwrapping), found (0 = "cannot find min."), r_zro, r_min.
 * Returns 0. */
int orc_cli_find_min (uint32_t blksiz, uint32_t n_chn, const float* data, uint64_t n_frames,
                      int stride, int link_chn,
                      int* min_angle, int* found, float* r_zro, float* r_min,
                      float* coarse_table /* optional [n_chn][360] after the coarse pass */);

/* the apply loop of main() (cli/phase-rotate.cc:950-1003) on an in-memory
 * file whose length is a whole number of blocks (Q3: the reference's tail
 * handling is broken otherwise).  out receives n_frames frames. mono-safe:
 * for n_chn > 1 the reference writes &buf[off] with off in samples, not
 * frames (:985); this restatement reproduces that only through
 * orc_cli_apply -- callers doing file parity should use n_chn = 1 or compare
 * per-block orc_cli_apply output. */
void orc_cli_apply_file_mono (uint32_t blksiz, const float* data, uint64_t n_frames,
                              int angle, float* out);

/* ---- LV2 engine (src/phaserotate.c) ------------------------------------- */
typedef struct orc_lv2 orc_lv2;
orc_lv2* orc_lv2_new (double rate, uint32_t n_chn);      /* instantiate :225-407 (DSP part) */
void     orc_lv2_free (orc_lv2*);
void     orc_lv2_activate (orc_lv2*);                    /* :511-520 */
uint32_t orc_lv2_latency (const orc_lv2*);               /* :297 */
void     orc_lv2_info (const orc_lv2*, uint32_t* fftlen, uint32_t* firlen, uint32_t* parsiz,
                       uint32_t* n_segm);
/* run() with both atom ports connected and an empty control sequence
 * (:774-852): in-place semantics, angle in degrees per channel.
 * lvl[c][0..1] (optional) receive lvl_in / lvl_out of process_channel. */
void     orc_lv2_run (orc_lv2*, const float* const* in, float* const* out,
                      uint32_t n_samples, const float* angle_deg, float (*lvl)[2]);
/* meter state after the last run (momentary/peak/diff), for the a7 row */
void     orc_lv2_meters (const orc_lv2*, uint32_t chn, float* momentary2, float* peak2, float* diff2);

#ifdef __cplusplus
}
#endif
#endif
