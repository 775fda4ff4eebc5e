/*
 * ref_peak_shim.c -- harness that compiles the reference's OWN
 * cli/dsp_peak_calc.h (unmodified, read in place from /root/reference) into
 * oracle/_ref/libref_peak.so.  This header is the only part of the hot path
 * that builds without libraries the image lacks (no FFTW / libsndfile / LV2).
 * With -mavx on x86_64 the AVX variant (cli/dsp_peak_calc.h:143-228) is the
 * one selected, exactly as cli/Makefile:34 does.
 *
 * TEST INFRASTRUCTURE: used to pin orc_compute_peak and, through it, the
 * abs-max reduction of the HIP sweep kernel.  Never shipped, never copied.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#include "dsp_peak_calc.h" /* -I/root/reference/cli */

float
ref_dsp_compute_peak (const float* src, uint32_t n_samples, float current)
{
	return dsp_compute_peak (src, n_samples, current);
}
