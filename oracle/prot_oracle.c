/*
 * prot_oracle.c -- CPU restatement of the phaserotate.lv2 hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- see prot_oracle.h for who may use it and for
 * the parity status ("parity unpinned" against reference outputs; the
 * reference is unbuildable in this image).
 *
 * Citations are file:line into /root/reference.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* M_PI, sincosf */
#endif
#include "prot_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* FFT: FFTW r2c / c2r semantics, own radix-2 implementation                 */
/* ------------------------------------------------------------------------ */

#define ORC_MAX_PLANS 32

typedef struct {
	int     m;   /* complex length (2^a * 3^b) */
	int     pow2;
	float*  twf; /* m (cos, -sin) pairs, exp(-2 pi i k/m), full circle */
	double* twd;
	int*    rev; /* bit reversal, power-of-two lengths only */
} orc_plan;

static orc_plan        g_plans[ORC_MAX_PLANS];
static int             g_n_plans = 0;
static pthread_mutex_t g_plan_lock = PTHREAD_MUTEX_INITIALIZER;

static const orc_plan*
get_plan (int m)
{
	pthread_mutex_lock (&g_plan_lock);
	for (int i = 0; i < g_n_plans; ++i) {
		if (g_plans[i].m == m) {
			pthread_mutex_unlock (&g_plan_lock);
			return &g_plans[i];
		}
	}
	if (g_n_plans == ORC_MAX_PLANS) {
		abort ();
	}
	orc_plan* p = &g_plans[g_n_plans];
	p->m        = m;
	p->pow2     = (m & (m - 1)) == 0;
	p->twf      = (float*)malloc (sizeof (float) * 2 * (size_t)m);
	p->twd      = (double*)malloc (sizeof (double) * 2 * (size_t)m);
	p->rev      = (int*)calloc ((size_t)m, sizeof (int));
	for (int k = 0; k < m; ++k) {
		const double a  = -2.0 * M_PI * (double)k / (double)m;
		p->twd[2 * k]     = cos (a);
		p->twd[2 * k + 1] = sin (a);
		p->twf[2 * k]     = (float)p->twd[2 * k];
		p->twf[2 * k + 1] = (float)p->twd[2 * k + 1];
	}
	if (p->pow2) {
		int bits = 0;
		while ((1 << bits) < m) {
			++bits;
		}
		for (int i = 0; i < m; ++i) {
			int r = 0;
			for (int b = 0; b < bits; ++b) {
				if (i & (1 << b)) {
					r |= 1 << (bits - 1 - b);
				}
			}
			p->rev[i] = r;
		}
	}
	++g_n_plans;
	pthread_mutex_unlock (&g_plan_lock);
	return p;
}

/* in-place complex FFT of length m, sign = -1 forward / +1 backward,
 * unnormalised.  Power-of-two lengths: iterative radix-2.  Lengths with a
 * factor 3 (the LV2 48 kHz FIR design is a 3072-point c2r,
 * src/phaserotate.c:364): one decimation-in-time radix-3 split per factor. */
#define DEFINE_CFFT(NAME, T, TW)                                              \
	static void NAME (const orc_plan* p, T* z, int sign)                      \
	{                                                                         \
		const int m = p->m;                                                   \
		if (!p->pow2) {                                                       \
			const int       q  = m / 3;                                       \
			const orc_plan* pq = get_plan (q);                                \
			T*              y  = (T*)malloc (sizeof (T) * 2 * (size_t)m);     \
			for (int j = 0; j < 3; ++j) {                                     \
				for (int i = 0; i < q; ++i) {                                 \
					y[2 * (j * q + i)]     = z[2 * (3 * i + j)];              \
					y[2 * (j * q + i) + 1] = z[2 * (3 * i + j) + 1];          \
				}                                                             \
				NAME (pq, &y[2 * j * q], sign);                               \
			}                                                                 \
			for (int k = 0; k < m; ++k) {                                     \
				const int kq = k % q;                                         \
				T         sr = y[2 * kq], si = y[2 * kq + 1];                 \
				for (int j = 1; j < 3; ++j) {                                 \
					const int e  = (int)(((long long)j * k) % m);             \
					const T   wr = p->TW[2 * e];                              \
					const T   wi = sign < 0 ? p->TW[2 * e + 1] : -p->TW[2 * e + 1]; \
					const T   ar = y[2 * (j * q + kq)], ai = y[2 * (j * q + kq) + 1]; \
					sr += ar * wr - ai * wi;                                  \
					si += ar * wi + ai * wr;                                  \
				}                                                             \
				z[2 * k]     = sr;                                            \
				z[2 * k + 1] = si;                                            \
			}                                                                 \
			free (y);                                                         \
			return;                                                           \
		}                                                                     \
		for (int i = 0; i < m; ++i) {                                         \
			const int r = p->rev[i];                                          \
			if (r > i) {                                                      \
				T t0 = z[2 * i], t1 = z[2 * i + 1];                           \
				z[2 * i]     = z[2 * r];                                      \
				z[2 * i + 1] = z[2 * r + 1];                                  \
				z[2 * r]     = t0;                                            \
				z[2 * r + 1] = t1;                                            \
			}                                                                 \
		}                                                                     \
		for (int len = 2; len <= m; len <<= 1) {                              \
			const int half = len >> 1;                                        \
			const int step = m / len;                                         \
			for (int i = 0; i < m; i += len) {                                \
				for (int k = 0; k < half; ++k) {                              \
					const T wr = p->TW[2 * k * step];                         \
					const T wi = sign < 0 ? p->TW[2 * k * step + 1]           \
					                      : -p->TW[2 * k * step + 1];         \
					T* a = &z[2 * (i + k)];                                   \
					T* b = &z[2 * (i + k + half)];                            \
					const T tr = b[0] * wr - b[1] * wi;                       \
					const T ti = b[0] * wi + b[1] * wr;                       \
					b[0] = a[0] - tr;                                         \
					b[1] = a[1] - ti;                                         \
					a[0] += tr;                                               \
					a[1] += ti;                                               \
				}                                                             \
			}                                                                 \
		}                                                                     \
	}

DEFINE_CFFT (cfft_f32, float, twf)
DEFINE_CFFT (cfft_f64, double, twd)

/* r2c of length n through a complex FFT of length n/2 (standard even/odd split) */
#define DEFINE_R2C(NAME, T, CFFT, TW)                                         \
	void NAME (int n, const T* in, T* out)                                    \
	{                                                                         \
		const int       m  = n / 2;                                           \
		const orc_plan* pm = get_plan (m);                                    \
		const orc_plan* pn = get_plan (n);                                    \
		T*              z  = (T*)malloc (sizeof (T) * (size_t)n);             \
		memcpy (z, in, sizeof (T) * (size_t)n);                               \
		CFFT (pm, z, -1);                                                     \
		for (int k = 0; k <= m; ++k) {                                        \
			const int ka = k % m, kb = (m - k) % m;                           \
			const T   ar = z[2 * ka], ai = z[2 * ka + 1];                     \
			const T   br = z[2 * kb], bi = -z[2 * kb + 1]; /* conj */         \
			const T   er = (T)0.5 * (ar + br), ei = (T)0.5 * (ai + bi);       \
			/* O = (A - B) / (2i) */                                          \
			const T   or_ = (T)0.5 * (ai - bi), oi = (T)-0.5 * (ar - br);     \
			T wr, wi;                                                         \
			if (k < m) {                                                      \
				wr = pn->TW[2 * k];                                           \
				wi = pn->TW[2 * k + 1];                                       \
			} else {                                                          \
				wr = (T)-1;                                                   \
				wi = (T)0;                                                    \
			}                                                                 \
			out[2 * k]     = er + (or_ * wr - oi * wi);                       \
			out[2 * k + 1] = ei + (or_ * wi + oi * wr);                       \
		}                                                                     \
		free (z);                                                             \
	}

DEFINE_R2C (orc_fft_r2c_own, float, cfft_f32, twf)
DEFINE_R2C (orc_fft_r2c_f64, double, cfft_f64, twd)

#define DEFINE_C2R(NAME, T, CFFT, TW)                                         \
	void NAME (int n, const T* in, T* out)                                    \
	{                                                                         \
		const int       m  = n / 2;                                           \
		const orc_plan* pm = get_plan (m);                                    \
		const orc_plan* pn = get_plan (n);                                    \
		T*              z  = (T*)malloc (sizeof (T) * (size_t)n);             \
		for (int k = 0; k < m; ++k) {                                         \
			T ar = in[2 * k], ai = in[2 * k + 1];                             \
			T br = in[2 * (m - k)], bi = -in[2 * (m - k) + 1]; /* conj */     \
			if (k == 0) {                                                     \
				ai = 0; /* Im(DC), Im(Nyquist) ignored */                     \
				bi = 0;                                                       \
			}                                                                 \
			const T er = ar + br, ei = ai + bi; /* 2E */                      \
			const T dr = ar - br, di = ai - bi; /* 2(A-B)/2 */                \
			const T wr = pn->TW[2 * k], wi = -pn->TW[2 * k + 1]; /* e^{+} */  \
			const T or_ = dr * wr - di * wi, oi = dr * wi + di * wr;          \
			/* Z = E + i O, scaled by 2 to give the unnormalised c2r */       \
			z[2 * k]     = er - oi;                                           \
			z[2 * k + 1] = ei + or_;                                          \
		}                                                                     \
		CFFT (pm, z, +1);                                                     \
		memcpy (out, z, sizeof (T) * (size_t)n);                              \
		free (z);                                                             \
	}

DEFINE_C2R (orc_fft_c2r_own, float, cfft_f32, twf)
DEFINE_C2R (orc_fft_c2r_f64, double, cfft_f64, twd)

/* ------------------------------------------------------------------------ */
/* External FFT behind the same two calls (CPU baseline): a real   */
/* libfftw3f.so.3 if the host has one, else MKL's FFTW3 interface (libmkl_rt) */
/* -- the library the reference would link (src/phaserotate.c:648,657,        */
/* cli/phase-rotate.cc:193,205).  Resolved with dlopen/dlsym against          */
/* prototypes written out here (FFTW's documented API); never used unless     */
/* orc_fft_use_external (1) is called: tests and parity run on the own FFT.   */
#include <dlfcn.h>
#include <pthread.h>
typedef struct orc_ext_plan_s* orc_ext_plan;
static struct {
	void*        lib;
	int          kind; /* 0 none, 1 fftw3f, 2 mkl */
	int          on;
	orc_ext_plan (*plan_r2c) (int, float*, float (*)[2], unsigned);
	orc_ext_plan (*plan_c2r) (int, float (*)[2], float*, unsigned);
	void (*exec_r2c) (orc_ext_plan, float*, float (*)[2]);
	void (*exec_c2r) (orc_ext_plan, float (*)[2], float*);
	struct { int n; orc_ext_plan r2c, c2r; } plans[16];
	int             n_plans;
	pthread_mutex_t lock;
} g_ext = { .lock = PTHREAD_MUTEX_INITIALIZER };

int orc_fft_use_external (int on)
{
	pthread_mutex_lock (&g_ext.lock);
	if (on && !g_ext.lib) {
		static const char* const names[][2] = { { "libfftw3f.so.3", "1" }, { "libfftw3f.so", "1" }, { "libmkl_rt.so", "2" },
			                                    { "libmkl_rt.so.2", "2" }, { "/opt/conda/lib/libmkl_rt.so", "2" }, { "/opt/conda/lib/libmkl_rt.so.2", "2" } };
		for (unsigned i = 0; i < sizeof (names) / sizeof (names[0]) && !g_ext.lib; ++i) {
			void* l = dlopen (names[i][0], RTLD_NOW | RTLD_LOCAL);
			if (!l) {
				continue;
			}
			g_ext.plan_r2c = (orc_ext_plan (*) (int, float*, float (*)[2], unsigned))dlsym (l, "fftwf_plan_dft_r2c_1d");
			g_ext.plan_c2r = (orc_ext_plan (*) (int, float (*)[2], float*, unsigned))dlsym (l, "fftwf_plan_dft_c2r_1d");
			g_ext.exec_r2c = (void (*) (orc_ext_plan, float*, float (*)[2]))dlsym (l, "fftwf_execute_dft_r2c");
			g_ext.exec_c2r = (void (*) (orc_ext_plan, float (*)[2], float*))dlsym (l, "fftwf_execute_dft_c2r");
			if (g_ext.plan_r2c && g_ext.plan_c2r && g_ext.exec_r2c && g_ext.exec_c2r) {
				g_ext.lib  = l;
				g_ext.kind = names[i][1][0] - '0';
			} else {
				dlclose (l);
			}
		}
	}
	g_ext.on = on && g_ext.lib;
	const int k = g_ext.on ? g_ext.kind : 0;
	pthread_mutex_unlock (&g_ext.lock);
	return k;
}

const char* orc_fft_backend_name (void)
{
	return !g_ext.on ? "own radix-2 FFT (oracle/prot_oracle.c)" : g_ext.kind == 1 ? "libfftw3f.so.3" : "libmkl_rt.so (MKL's FFTW3 interface)";
}

/* plans are made once per length, with FFTW_ESTIMATE | FFTW_UNALIGNED | FFTW_PRESERVE_INPUT, like the reference's
 * FFTW_ESTIMATE plans (src/phaserotate.c:361-364, cli/phase-rotate.cc:137-141); execution is the thread-safe
 * new-array form */
static int ext_plans (int n, orc_ext_plan* r2c, orc_ext_plan* c2r)
{
	pthread_mutex_lock (&g_ext.lock);
	for (int i = 0; i < g_ext.n_plans; ++i) {
		if (g_ext.plans[i].n == n) {
			*r2c = g_ext.plans[i].r2c;
			*c2r = g_ext.plans[i].c2r;
			pthread_mutex_unlock (&g_ext.lock);
			return 1;
		}
	}
	int ok = 0;
	if (g_ext.n_plans < 16) {
		float* a            = (float*)calloc ((size_t)n + 2, sizeof (float));
		float (*b)[2]       = (float (*)[2])calloc ((size_t)n / 2 + 1, sizeof (float[2]));
		const unsigned flag = (1u << 6) | (1u << 1) | (1u << 4);
		orc_ext_plan p1 = g_ext.plan_r2c (n, a, b, flag), p2 = g_ext.plan_c2r (n, b, a, flag);
		free (a);
		free (b);
		if (p1 && p2) {
			g_ext.plans[g_ext.n_plans].n   = n;
			g_ext.plans[g_ext.n_plans].r2c = *r2c = p1;
			g_ext.plans[g_ext.n_plans].c2r = *c2r = p2;
			++g_ext.n_plans;
			ok = 1;
		}
	}
	pthread_mutex_unlock (&g_ext.lock);
	return ok;
}

void orc_fft_r2c (int n, const float* in, float* out)
{
	orc_ext_plan p1, p2;
	if (g_ext.on && ext_plans (n, &p1, &p2)) {
		g_ext.exec_r2c (p1, (float*)in, (float (*)[2])out); /* r2c preserves its input */
		return;
	}
	orc_fft_r2c_own (n, in, out);
}

void orc_fft_c2r (int n, const float* in, float* out)
{
	orc_ext_plan p1, p2;
	if (g_ext.on && ext_plans (n, &p1, &p2)) {
		g_ext.exec_c2r (p2, (float (*)[2])in, out);
		return;
	}
	orc_fft_c2r_own (n, in, out);
}

/* ------------------------------------------------------------------------ */
/* peak: scalar variant cli/dsp_peak_calc.h:311-325 (the AVX variant :143-228 */
/* computes the same exact value: max and |.| do not round)                  */
/* ------------------------------------------------------------------------ */

float
orc_compute_peak (const float* buf, uint32_t n_samples, float current)
{
	for (uint32_t i = 0; i < n_samples; ++i) {
		const float x = fabsf (buf[i]);
		if (x > current) {
			current = x;
		}
	}
	return current;
}

/* cli/phase-rotate.cc:98-121 */
static float
calc_rotated_peak (const float* b0, const float* b1, uint32_t n_samples, float pk, int angle)
{
	float sa, ca;
	orc_sincos_lut (angle, &sa, &ca);
	for (uint32_t i = 0; i < n_samples; ++i) {
		/* the reference is built without -mfma: two products, one sum */
		volatile float p0 = ca * b0[i];
		volatile float p1 = sa * b1[i];
		const float    x  = p0 + p1;
		pk                = fmaxf (pk, fabsf (x));
	}
	return pk;
}

/* ------------------------------------------------------------------------ */
/* SinCosLut cli/phase-rotate.cc:41-74                                       */
/* ------------------------------------------------------------------------ */

static float          g_lut_sin[ORC_MAXSAMPLE];
static float          g_lut_cos[ORC_MAXSAMPLE];
static pthread_once_t g_lut_once = PTHREAD_ONCE_INIT;

static void
lut_init (void)
{
	float mp = 2.f * M_PI / ORC_SUBSAMPLE / -360.0; /* :46 */
	for (int i = 0; i < ORC_MAXSAMPLE; ++i) {
		sincosf (mp * i, &g_lut_sin[i], &g_lut_cos[i]); /* :52 */
	}
}

void
orc_sincos_lut (int a, float* s, float* c)
{
	pthread_once (&g_lut_once, lut_init);
	*s = g_lut_sin[a];
	*c = g_lut_cos[a];
}

/* ------------------------------------------------------------------------ */
/* FIR design                                                                */
/* ------------------------------------------------------------------------ */

void
orc_fir_design (int n, float* fir)
{
	/* src/phaserotate.c:374-380, cli/phase-rotate.cc:148-156 */
	const int half = n / 2;
	float*    spec = (float*)calloc ((size_t)(half + 1) * 2, sizeof (float));
	for (int i = 0; i <= half; ++i) {
		spec[2 * i]     = 0;
		spec[2 * i + 1] = (i & 1) ? -1.f : 1.f;
	}
	orc_fft_c2r (n, spec, fir);
	free (spec);
	/* src/phaserotate.c:386-391, cli/phase-rotate.cc:158-161 */
	const double fnorm = 0.5 / n;
	const double flen  = 1.0 / n;
	for (int i = 0; i < n; ++i) {
		fir[i] *= fnorm * (1 - cos (2.0 * M_PI * i * flen));
	}
}

void
orc_fir_closed_form (int n, double* fir)
{
	for (int i = 0; i < n; ++i) {
		const int m = i - n / 2;
		if ((m & 1) == 0) {
			fir[i] = 0;
		} else {
			const double a = M_PI * (double)m / (double)n;
			fir[i] = -(2.0 / n) * (cos (a) / sin (a)) * 0.5 * (1.0 - cos (2.0 * M_PI * (double)i / (double)n));
		}
	}
}

/* ------------------------------------------------------------------------ */
/* CLI: PhaseRotateProc cli/phase-rotate.cc:125-245                          */
/* ------------------------------------------------------------------------ */

struct orc_cli_proc {
	uint32_t fftlen, parsiz, firlen;
	float    norm;
	float*   time_data; /* fftlen */
	float*   freq_data; /* (parsiz+1) complex */
	float*   ffir_data; /* (parsiz+1) complex */
};

orc_cli_proc*
orc_cli_proc_new (uint32_t blksiz)
{
	orc_cli_proc* p = (orc_cli_proc*)calloc (1, sizeof (orc_cli_proc));
	p->fftlen       = blksiz * 2; /* :129 */
	p->parsiz       = blksiz;     /* :130 */
	p->firlen       = blksiz / 2; /* :131 */
	p->time_data    = (float*)calloc (p->fftlen, sizeof (float));
	p->freq_data    = (float*)calloc ((size_t)(p->parsiz + 1) * 2, sizeof (float));
	p->ffir_data    = (float*)calloc ((size_t)(p->parsiz + 1) * 2, sizeof (float));
	p->norm         = 0.5 / p->parsiz; /* :142 */

	float* fir = (float*)calloc (p->fftlen, sizeof (float));
	orc_fir_design ((int)p->parsiz, fir); /* :144-161 (FIR length == _parsiz) */
	memset (fir + p->parsiz, 0, p->parsiz * sizeof (float)); /* :163 */
	orc_fft_r2c ((int)p->fftlen, fir, p->ffir_data);          /* :164 */
	free (fir);
	return p;
}

void
orc_cli_proc_free (orc_cli_proc* p)
{
	if (!p) {
		return;
	}
	free (p->time_data);
	free (p->freq_data);
	free (p->ffir_data);
	free (p);
}

uint32_t
orc_cli_proc_parsiz (const orc_cli_proc* p)
{
	return p->parsiz;
}

const float*
orc_cli_proc_ffir (const orc_cli_proc* p)
{
	return p->ffir_data;
}

void
orc_cli_proc_hilbert (orc_cli_proc* p, const float* in, float* out, float* o_out)
{
	const uint32_t parsiz = p->parsiz;
	memcpy (out, o_out, sizeof (float) * parsiz);                  /* :187 */
	memcpy (p->time_data, in + parsiz, sizeof (float) * parsiz);   /* :190 */
	memset (p->time_data + parsiz, 0, sizeof (float) * parsiz);    /* :191 */
	orc_fft_r2c ((int)p->fftlen, p->time_data, p->freq_data);      /* :193 */
	const float  norm = p->norm;
	const float* ffir = p->ffir_data;
	float*       fd   = p->freq_data;
	for (uint32_t i = 0; i <= parsiz; ++i) { /* :199-203 */
		const float re = fd[2 * i];
		fd[2 * i]      = norm * (re * ffir[2 * i] - fd[2 * i + 1] * ffir[2 * i + 1]);
		fd[2 * i + 1]  = norm * (re * ffir[2 * i + 1] + fd[2 * i + 1] * ffir[2 * i]);
	}
	orc_fft_c2r ((int)p->fftlen, fd, p->time_data); /* :205 */
	for (uint32_t i = 0; i < parsiz; ++i) {          /* :208-210 */
		out[i] += p->time_data[i];
	}
	memcpy (o_out, p->time_data + parsiz, sizeof (float) * parsiz); /* :211 */
}

void
orc_cli_proc_rotate (orc_cli_proc* p, const float* in, float* out, int angle)
{
	float sa, ca;
	orc_sincos_lut (angle, &sa, &ca);
	const float* fin = &in[p->firlen]; /* :220 */
	for (uint32_t i = 0; i < p->parsiz; ++i) {
		volatile float p0 = ca * fin[i];
		volatile float p1 = sa * out[i];
		out[i]            = p0 + p1; /* :223 */
	}
}

void
orc_cli_proc_process (orc_cli_proc* p, const float* in, float* out, float* o_out, int angle)
{
	orc_cli_proc_hilbert (p, in, out, o_out);
	orc_cli_proc_rotate (p, in, out, angle);
}

/* ------------------------------------------------------------------------ */
/* CLI: PhaseRotate cli/phase-rotate.cc:251-485                              */
/* (one std::thread per channel there; serial here -- the threads share      */
/* nothing but read-only input, so the results are identical)                */
/* ------------------------------------------------------------------------ */

struct orc_cli {
	uint32_t       n_chn, parsiz;
	orc_cli_proc** proc;
	float**        buf_out;
	float**        buf_old;
	float**        buf_olp;
	float**        peak;
};

orc_cli*
orc_cli_new (uint32_t blksiz, uint32_t n_chn)
{
	orc_cli* s = (orc_cli*)calloc (1, sizeof (orc_cli));
	s->n_chn   = n_chn;
	s->parsiz  = blksiz;
	s->proc    = (orc_cli_proc**)calloc (n_chn, sizeof (void*));
	s->buf_out = (float**)calloc (n_chn, sizeof (float*));
	s->buf_old = (float**)calloc (n_chn, sizeof (float*));
	s->buf_olp = (float**)calloc (n_chn, sizeof (float*));
	s->peak    = (float**)calloc (n_chn, sizeof (float*));
	for (uint32_t c = 0; c < n_chn; ++c) {
		s->proc[c]    = orc_cli_proc_new (blksiz);
		s->buf_out[c] = (float*)calloc (blksiz, sizeof (float));
		s->buf_old[c] = (float*)calloc (blksiz, sizeof (float));
		s->buf_olp[c] = (float*)calloc (blksiz, sizeof (float));
		s->peak[c]    = (float*)calloc (ORC_MAXSAMPLE, sizeof (float));
	}
	return s;
}

void
orc_cli_free (orc_cli* s)
{
	if (!s) {
		return;
	}
	for (uint32_t c = 0; c < s->n_chn; ++c) {
		orc_cli_proc_free (s->proc[c]);
		free (s->buf_out[c]);
		free (s->buf_old[c]);
		free (s->buf_olp[c]);
		free (s->peak[c]);
	}
	free (s->proc);
	free (s->buf_out);
	free (s->buf_old);
	free (s->buf_olp);
	free (s->peak);
	free (s);
}

void
orc_cli_reset (orc_cli* s)
{
	for (uint32_t c = 0; c < s->n_chn; ++c) {
		memset (s->buf_old[c], 0, s->parsiz * sizeof (float));
		memset (s->buf_olp[c], 0, s->parsiz * sizeof (float));
		for (uint32_t a = 0; a < ORC_MAXSAMPLE; ++a) {
			s->peak[c][a] = 0;
		}
	}
}

/* thr_process cli/phase-rotate.cc:388-429 */
static void
cli_thr_process (orc_cli* s, const float* p_in, int ang_start, int ang_end, int ang_stride, int c, int start)
{
	const uint32_t parsiz = s->parsiz;
	const uint32_t fftlen = 2 * parsiz;
	const uint32_t firlen = parsiz / 2;

	float* tdc = (float*)malloc (sizeof (float) * fftlen);
	float* hil = (float*)malloc (sizeof (float) * parsiz);

	memcpy (tdc, s->buf_old[c], parsiz * sizeof (float)); /* :398 */
	for (uint32_t i = 0; i < parsiz; ++i) {
		tdc[i + parsiz] = p_in[c + i * s->n_chn]; /* :400 */
	}
	memcpy (s->buf_old[c], &tdc[parsiz], parsiz * sizeof (float)); /* :404 */

	orc_cli_proc_hilbert (s->proc[c], tdc, hil, s->buf_olp[c]); /* :407 */

	int angle = ang_start;
	while (angle <= ang_end) { /* :410 */
		int a = (angle + ORC_MAXSAMPLE) % ORC_MAXSAMPLE;
		if (angle == 0) {
			s->peak[c][a] = orc_compute_peak (s->buf_old[c], parsiz, s->peak[c][a]); /* :414 */
		} else {
			memcpy (s->buf_out[c], hil, parsiz * sizeof (float)); /* :416 */
			if (start) {
				s->peak[c][a] = calc_rotated_peak (&tdc[firlen], &s->buf_out[c][firlen], firlen, s->peak[c][a], a); /* :419 */
			} else {
				s->peak[c][a] = calc_rotated_peak (&tdc[firlen], s->buf_out[c], parsiz, s->peak[c][a], a); /* :421 */
			}
		}
		angle += ang_stride;
		if (angle >= ang_end) { /* :425 */
			break;
		}
	}
	free (tdc);
	free (hil);
}

void
orc_cli_analyze (orc_cli* s, const float* in, int ang_start, int ang_end, int ang_stride, int chn, int start)
{
	uint32_t c0 = (chn < 0) ? 0 : (uint32_t)chn; /* :434-435 */
	uint32_t c1 = (chn < 0) ? s->n_chn : (uint32_t)chn + 1;
	for (uint32_t c = c0; c < c1; ++c) {
		cli_thr_process (s, in, ang_start, ang_end, ang_stride, (int)c, start);
	}
}

/* thr_apply + apply cli/phase-rotate.cc:446-485 */
void
orc_cli_apply (orc_cli* s, float* buf, const int* angles)
{
	const uint32_t parsiz = s->parsiz;
	float*         tdc    = (float*)malloc (sizeof (float) * 2 * parsiz);
	for (uint32_t c = 0; c < s->n_chn; ++c) {
		memcpy (tdc, s->buf_old[c], parsiz * sizeof (float)); /* :455 */
		for (uint32_t i = 0; i < parsiz; ++i) {
			tdc[i + parsiz] = buf[c + i * s->n_chn]; /* :458 */
		}
		memcpy (s->buf_old[c], &tdc[parsiz], parsiz * sizeof (float)); /* :461 */
		int a = (angles[c] + ORC_MAXSAMPLE) % ORC_MAXSAMPLE;           /* :463 */
		orc_cli_proc_process (s->proc[c], tdc, s->buf_out[c], s->buf_olp[c], a);
	}
	for (uint32_t c = 0; c < s->n_chn; ++c) { /* :478-484 */
		for (uint32_t i = 0; i < parsiz; ++i) {
			buf[c + i * s->n_chn] = s->buf_out[c][i];
		}
	}
	free (tdc);
}

float
orc_cli_peak_all (const orc_cli* s, int a)
{
	float p = 0;
	if (a < 0) {
		a += ORC_MAXSAMPLE;
	}
	a = a % ORC_MAXSAMPLE;
	for (uint32_t c = 0; c < s->n_chn; ++c) {
		p = fmaxf (p, s->peak[c][a]);
	}
	return p;
}

float
orc_cli_peak (const orc_cli* s, int c, int a)
{
	if (c < 0 || (uint32_t)c > s->n_chn) { /* :278 (sic: '>' not '>=') */
		return orc_cli_peak_all (s, a);
	}
	if (a < 0) {
		a += ORC_MAXSAMPLE;
	}
	return s->peak[c][a % ORC_MAXSAMPLE];
}

/* analyze_file cli/phase-rotate.cc:565-587 with sf_readf_float replaced by a
 * cursor into an in-memory interleaved float file */
void
orc_cli_analyze_file (orc_cli* s, const float* data, uint64_t n_frames, int ang_start, int ang_end, int ang_stride, int chn)
{
	const uint32_t blksiz = s->parsiz;
	const uint32_t n_chn  = s->n_chn;
	float*         buf    = (float*)malloc (sizeof (float) * blksiz * n_chn);
	uint64_t       pos    = 0;
	int            start  = 1;
	for (;;) {
		uint64_t n = n_frames - pos;
		if (n > blksiz) {
			n = blksiz;
		}
		if (n == 0) { /* :574 */
			break;
		}
		memcpy (buf, data + pos * n_chn, sizeof (float) * n * n_chn);
		pos += n;
		if (n < blksiz) { /* :577-580 */
			memset (&buf[n_chn * n], 0, sizeof (float) * n_chn * (blksiz - n));
		}
		orc_cli_analyze (s, buf, ang_start, ang_end, ang_stride, chn, start); /* :581 */
		start = 0;
	}
	memset (buf, 0, blksiz * n_chn * sizeof (float)); /* :585 */
	orc_cli_analyze (s, buf, ang_start, ang_end, ang_stride, chn, 0);
	free (buf);
}

/* main() minimum search cli/phase-rotate.cc:784-929 */
int
orc_cli_find_min (uint32_t blksiz, uint32_t n_chn, const float* data, uint64_t n_frames,
                  int stride, int link_chn,
                  int* min_angle, int* found, float* r_zro, float* r_min,
                  float* coarse_table)
{
	orc_cli* pr = orc_cli_new (blksiz, n_chn);

	orc_cli_analyze_file (pr, data, n_frames, 0, ORC_MAXSAMPLE, stride, -1); /* :784 */

	if (coarse_table) {
		for (uint32_t c = 0; c < n_chn; ++c) {
			memcpy (&coarse_table[c * ORC_MAXSAMPLE], pr->peak[c], sizeof (float) * ORC_MAXSAMPLE);
		}
	}

	/* std::map<int, std::vector<int>> mins -- keys are in [0, 360), iterate ascending */
	int* mins_n = (int*)calloc (ORC_MAXSAMPLE, sizeof (int));
	int* mins_c = (int*)calloc ((size_t)ORC_MAXSAMPLE * n_chn, sizeof (int));
	float* p_min = (float*)malloc (sizeof (float) * n_chn);

	for (uint32_t c = 0; c < n_chn; ++c) {
		float c_min = INFINITY;
		float c_max = 0;
		float range;

		/* the reference leaves p_min[c]/min_angle[c] uninitialised when
		 * range == 0 (:836-839) and later reads them (:909): undefined
		 * behaviour there; defined here as "not found". */
		p_min[c]     = INFINITY;
		min_angle[c] = 0;
		r_min[c]     = 0;
		r_zro[c]     = orc_cli_peak (pr, (int)c, 0); /* :828 */

		for (int a = 0; a < ORC_MAXSAMPLE; a += stride) { /* :830-833 */
			const float p = orc_cli_peak (pr, link_chn ? -1 : (int)c, a);
			c_min         = fminf (c_min, p);
			c_max         = fmaxf (c_max, p);
		}
		range = c_max - c_min;
		if (range == 0) {
			mins_c[0 * n_chn + mins_n[0]++] = (int)c; /* :837 */
			continue;
		}
		if (stride > 1) {
			range *= .07f; /* :841 (float *= double literal, result float) */
			p_min[c] = INFINITY;
		} else {
			range    = 0;
			p_min[c] = c_min;
		}
		for (int a = 0; a < ORC_MAXSAMPLE; a += stride) { /* :848-856 */
			const float p = orc_cli_peak (pr, link_chn ? -1 : (int)c, a);
			if (p <= c_min + range) {
				mins_c[a * n_chn + mins_n[a]++] = (int)c;
			}
		}
	}

	if (stride == 1) { /* :859-865 */
		for (int a = 0; a < ORC_MAXSAMPLE; ++a) {
			for (int k = 0; k < mins_n[a]; ++k) {
				const int cn  = mins_c[a * n_chn + k];
				min_angle[cn] = a;
				r_min[cn]     = orc_cli_peak (pr, cn, a);
			}
		}
	} else {
		const int stride_2 = (stride + 1) / 2; /* :868 */
		for (int ma = 0; ma < ORC_MAXSAMPLE; ++ma) {
			if (mins_n[ma] == 0) {
				continue;
			}
			orc_cli_reset (pr); /* :876 */
			orc_cli_analyze_file (pr, data, n_frames, ma - stride_2, ma + stride_2 + 1, 1,
			                      mins_n[ma] > 1 ? -1 : mins_c[ma * n_chn]); /* :880 */
			for (int k = 0; k < mins_n[ma]; ++k) {
				const int cn = mins_c[ma * n_chn + k];
				for (int a = ma - stride_2; a < ma + stride_2 + 1; ++a) { /* :883-889 */
					const float p = orc_cli_peak (pr, link_chn ? -1 : cn, a);
					if (p <= p_min[cn]) {
						p_min[cn]     = p;
						r_min[cn]     = orc_cli_peak (pr, cn, a);
						min_angle[cn] = (a + ORC_MAXSAMPLE) % ORC_MAXSAMPLE;
					}
				}
			}
		}
	}

	/* collect results :905-929 */
	float avg_rotate = 0;
	int   avg_count  = 0;
	for (uint32_t c = 0; c < n_chn; ++c) {
		if (p_min[c] != INFINITY) {
			avg_rotate += min_angle[c];
			++avg_count;
		}
	}
	avg_rotate /= avg_count;
	const float avg_dist = ORC_MAXSAMPLE / (float)avg_count;

	for (uint32_t c = 0; c < n_chn; ++c) {
		if (p_min[c] == INFINITY) {
			found[c]     = 0;
			min_angle[c] = 0;
		} else {
			found[c] = 1;
			if (min_angle[c] > 90 * ORC_SUBSAMPLE && fabsf (min_angle[c] - avg_rotate) > avg_dist) {
				min_angle[c] -= ORC_MAXSAMPLE;
			} else if (avg_rotate > 90 * ORC_SUBSAMPLE) {
				min_angle[c] -= ORC_MAXSAMPLE;
			}
		}
	}

	free (mins_n);
	free (mins_c);
	free (p_min);
	orc_cli_free (pr);
	return 0;
}

/* apply loop of main() cli/phase-rotate.cc:950-1003, mono, whole-block file */
void
orc_cli_apply_file_mono (uint32_t blksiz, const float* data, uint64_t n_frames, int angle, float* out)
{
	orc_cli*       pr      = orc_cli_new (blksiz, 1);
	const uint32_t latency = blksiz / 2; /* :963 */
	float*         buf     = (float*)malloc (sizeof (float) * blksiz);
	uint64_t       pos = 0, wpos = 0;
	uint32_t       off = latency;
	while (pos < n_frames) {
		uint64_t n = n_frames - pos;
		if (n > blksiz) {
			n = blksiz;
		}
		memcpy (buf, data + pos, sizeof (float) * n);
		if (n < blksiz) {
			memset (buf + n, 0, sizeof (float) * (blksiz - n));
		}
		pos += n;
		orc_cli_apply (pr, buf, &angle); /* :981 */
		n -= off;
		memcpy (out + wpos, &buf[off], sizeof (float) * n); /* :985 */
		wpos += n;
		off = 0;
	}
	memset (buf, 0, blksiz * sizeof (float)); /* :995 */
	orc_cli_apply (pr, buf, &angle);
	uint64_t n = n_frames - wpos;
	if (n > latency) {
		n = latency;
	}
	memcpy (out + wpos, buf, sizeof (float) * n); /* :998 */
	free (buf);
	orc_cli_free (pr);
}

/* ------------------------------------------------------------------------ */
/* LV2 engine src/phaserotate.c                                              */
/* ------------------------------------------------------------------------ */

typedef struct {
	float    angle, sa, ca;
	float*   buf_dly;
	float*   buf_src;
	float*   buf_out;
	uint32_t offset;
	int32_t  overlap;
	float*   time_data;
	int      mtr_reset_delay;
	int      mtr_holdcnt[2];
	float    mtr_momentary[2];
	float    mtr_peak[2];
	float    mtr_diff[2];
} lv2_channel;

struct orc_lv2 {
	uint32_t n_chn;
	float    rate;
	uint32_t fftlen, firlen, parsiz, firlat, n_segm, latency;
	float    interp_th, interp_nm;
	uint32_t mtr_holdtme, mtr_fpp;
	float    mtr_falloff;
	float*   freq_data;
	float*   freq_sum;
	float**  freq_fir;
	lv2_channel channel[ORC_MAX_CHANNELS];
};

static void
lv2_sin_cos (float angle, float* s, float* c)
{
	static const float twopi = 2 * M_PI; /* :125 */
	sincosf (angle * twopi, s, c);       /* :131 */
}

orc_lv2*
orc_lv2_new (double rate, uint32_t n_chn)
{
	if (n_chn < 1 || n_chn > ORC_MAX_CHANNELS) {
		return NULL;
	}
	orc_lv2* self = (orc_lv2*)calloc (1, sizeof (orc_lv2));
	self->n_chn   = n_chn;
	if (rate < 64000) { /* :278-289 */
		self->fftlen = 512;
		self->firlen = 3072;
	} else if (rate < 128000) {
		self->fftlen = 1024;
		self->firlen = 4096;
	} else {
		self->fftlen = 2048;
		self->firlen = 8192;
	}
	self->rate      = rate; /* :291-297 */
	self->parsiz    = self->fftlen / 2;
	self->firlat    = self->firlen / 2;
	self->n_segm    = self->firlen / self->parsiz;
	self->interp_th = self->parsiz * 1e-6f;
	self->interp_nm = 1.f / self->parsiz;
	self->latency   = self->parsiz + self->firlat;

	self->mtr_holdtme = 0.5 * rate + 0.5f; /* :303-305 */
	self->mtr_falloff = 0;
	self->mtr_fpp     = 0;

	self->freq_data = (float*)calloc ((size_t)(self->parsiz + 1) * 2, sizeof (float));
	self->freq_sum  = (float*)calloc ((size_t)(self->parsiz + 1) * 2, sizeof (float));

	for (uint32_t chn = 0; chn < n_chn; ++chn) { /* channel_init :144-167 */
		lv2_channel* c = &self->channel[chn];
		c->angle       = 0;
		c->overlap     = 0;
		c->mtr_diff[0] = c->mtr_diff[1] = 1;
		lv2_sin_cos (c->angle, &c->sa, &c->ca);
		c->buf_dly   = (float*)calloc (self->latency, sizeof (float));
		c->buf_src   = (float*)calloc (self->firlen, sizeof (float));
		c->buf_out   = (float*)calloc (self->parsiz, sizeof (float));
		c->time_data = (float*)calloc (self->fftlen, sizeof (float));
	}

	self->freq_fir = (float**)calloc (self->n_segm, sizeof (float*));
	for (uint32_t i = 0; i < self->n_segm; ++i) {
		self->freq_fir[i] = (float*)calloc ((size_t)(self->parsiz + 1) * 2, sizeof (float));
	}

	float* fir = (float*)calloc (self->firlen, sizeof (float));
	orc_fir_design ((int)self->firlen, fir); /* :374-391 */

	const float norm      = 0.5 / self->parsiz; /* :393 */
	float*      time_data = self->channel[0].time_data;
	memset (&time_data[self->parsiz], 0, sizeof (float) * self->parsiz);
	for (uint32_t s = 0; s < self->n_segm; ++s) { /* :396-401 */
		for (uint32_t i = 0; i < self->parsiz; ++i) {
			time_data[i] = norm * fir[s * self->parsiz + i];
		}
		orc_fft_r2c ((int)self->fftlen, time_data, self->freq_fir[s]);
	}
	free (fir);
	/* instantiate leaves channel[0].time_data dirty (last FIR segment); the
	 * host must call activate (channel_reset :169-177) before run, as LV2
	 * requires. */
	return self;
}

void
orc_lv2_free (orc_lv2* self)
{
	if (!self) {
		return;
	}
	for (uint32_t chn = 0; chn < self->n_chn; ++chn) {
		lv2_channel* c = &self->channel[chn];
		free (c->buf_dly);
		free (c->buf_src);
		free (c->buf_out);
		free (c->time_data);
	}
	for (uint32_t i = 0; i < self->n_segm; ++i) {
		free (self->freq_fir[i]);
	}
	free (self->freq_fir);
	free (self->freq_data);
	free (self->freq_sum);
	free (self);
}

static void
lv2_meter_reset_peaks (lv2_channel* c) /* :489-495 */
{
	c->mtr_peak[0] = c->mtr_peak[1] = 0;
	c->mtr_diff[0] = c->mtr_diff[1] = 1;
	c->mtr_momentary[0] = c->mtr_momentary[1] = 0;
}

void
orc_lv2_activate (orc_lv2* self)
{
	for (uint32_t chn = 0; chn < self->n_chn; ++chn) {
		lv2_channel* c = &self->channel[chn];
		memset (c->buf_src, 0, self->firlen * sizeof (float)); /* channel_reset :169-177 */
		memset (c->buf_out, 0, self->parsiz * sizeof (float));
		memset (c->time_data, 0, self->fftlen * sizeof (float));
		c->overlap = 0;
		c->offset  = 0;
		lv2_meter_reset_peaks (c);
		c->mtr_reset_delay = (int)self->latency; /* :518 */
	}
}

uint32_t
orc_lv2_latency (const orc_lv2* self)
{
	return self->latency;
}

void
orc_lv2_info (const orc_lv2* self, uint32_t* fftlen, uint32_t* firlen, uint32_t* parsiz, uint32_t* n_segm)
{
	*fftlen = self->fftlen;
	*firlen = self->firlen;
	*parsiz = self->parsiz;
	*n_segm = self->n_segm;
}

static float
lv2_meter_proc (const orc_lv2* self, lv2_channel* c, float meter_peak, int m) /* :451-470 */
{
	if (!isfinite (meter_peak)) {
		meter_peak = 0;
	}
	if (meter_peak > c->mtr_peak[m]) {
		c->mtr_peak[m] = meter_peak;
	}
	if (meter_peak > c->mtr_momentary[m]) {
		c->mtr_momentary[m] = meter_peak;
		c->mtr_holdcnt[m]   = (int)self->mtr_holdtme;
	} else if (c->mtr_holdcnt[m] > 0) {
		c->mtr_holdcnt[m] -= (int)self->mtr_fpp;
	} else {
		c->mtr_momentary[m] *= self->mtr_falloff;
		c->mtr_momentary[m] += 1e-20f;
	}
	return meter_peak;
}

static float
lv2_meter_run (const orc_lv2* self, lv2_channel* c, const float* d, uint32_t n_samples, int m) /* :472-487 */
{
	float meter_peak = 0;
	for (uint32_t i = 0; i < n_samples; ++i) {
		float v = fabsf (d[i]);
		if (v > meter_peak) {
			meter_peak = v;
		}
	}
	return lv2_meter_proc (self, c, meter_peak, m);
}

static void
lv2_meter_delayed_reset (lv2_channel* c, uint32_t n_samples, uint32_t reset) /* :497-509 */
{
	if (c->mtr_reset_delay > 0) {
		c->mtr_diff[0]      = 1;
		c->mtr_diff[1]      = 1;
		c->mtr_momentary[1] = 0;
		c->mtr_reset_delay -= (int)n_samples;
	}
	if (reset > 0) {
		c->mtr_reset_delay = (int)(reset + n_samples);
	}
}

/* process_channel :538-772 (the ui_active forge block :741-771 excluded) */
static void
lv2_process_channel (orc_lv2* self, uint32_t chn, float* p_out, float p_angle, uint32_t n_samples, float* lvl)
{
	lv2_channel* c = &self->channel[chn];

	const uint32_t parsiz  = self->parsiz;
	const uint32_t firlen  = self->firlen;
	const uint32_t firlat  = self->firlat;
	const uint32_t n_segm  = self->n_segm;
	const uint32_t latency = self->latency;

	float* freq_data = self->freq_data;
	float* freq_sum  = self->freq_sum;

	float*   time_data = c->time_data;
	uint32_t offset    = c->offset;
	int32_t  overlap   = c->overlap;
	float    angle     = c->angle;
	uint32_t remain    = n_samples;

	float* iobuf        = p_out;
	float  target_angle = p_angle / -360.f; /* :564 */
	if (target_angle < -.5f) {
		target_angle = -.5f;
	}
	if (target_angle > 0.5f) {
		target_angle = 0.5f;
	}

	/* meter delayed input :573-609 (incl. the Q6 shift bug :586-588) */
	float lvl_in = 0;
	if (n_samples < latency) {
		lvl_in                 = lv2_meter_run (self, c, c->buf_dly, n_samples, 0);
		const uint32_t ovl = latency - n_samples;
		if (ovl <= n_samples) {
			memcpy (c->buf_dly, &c->buf_dly[n_samples], ovl * sizeof (float));
		} else {
			memcpy (c->buf_dly, &c->buf_dly[n_samples], n_samples * sizeof (float));
			for (uint32_t i = 0; i < ovl - n_samples; ++i) {
				c->buf_dly[n_samples] = c->buf_dly[n_samples + i]; /* sic :587 */
			}
		}
		memcpy (&c->buf_dly[ovl], iobuf, n_samples * sizeof (float));
	} else {
		float meter_peak = 0;
		for (uint32_t i = 0; i < latency; ++i) {
			float v = fabsf (c->buf_dly[i]);
			if (v > meter_peak) {
				meter_peak = v;
			}
		}
		uint32_t rem = n_samples - latency;
		for (uint32_t i = 0; i < rem; ++i) {
			float v = fabsf (iobuf[i]);
			if (v > meter_peak) {
				meter_peak = v;
			}
		}
		memcpy (c->buf_dly, &iobuf[rem], latency * sizeof (float));
		lvl_in = lv2_meter_proc (self, c, meter_peak, 0);
	}

	lv2_meter_delayed_reset (c, n_samples, target_angle != angle ? latency : 0); /* :611 */

	while (remain > 0) { /* :615 */
		uint32_t ns = parsiz - offset;
		if (ns > remain) {
			ns = remain;
		}
		memcpy (&c->buf_src[offset + overlap], iobuf, sizeof (float) * ns); /* :622 */
		memcpy (iobuf, &c->buf_out[offset], sizeof (float) * ns);           /* :623 */
		iobuf += ns;
		offset += ns;
		remain -= ns;

		if (offset == parsiz) { /* :629 */
			offset = 0;
			memcpy (c->buf_out, &time_data[parsiz], sizeof (float) * parsiz);   /* :633 */
			memset (&time_data[parsiz], 0, sizeof (float) * parsiz);            /* :636 */
			memset (freq_sum, 0, sizeof (float) * 2 * (parsiz + 1));            /* :637 */

			int32_t olp = overlap;
			for (uint32_t s = 0; s < n_segm; ++s) { /* :640-655 */
				memcpy (time_data, &c->buf_src[olp], sizeof (float) * parsiz);
				olp -= (int32_t)parsiz;
				if (olp < 0) {
					olp += (int32_t)firlen;
				}
				const float* ffir = self->freq_fir[s];
				orc_fft_r2c ((int)self->fftlen, time_data, freq_data);
				for (uint32_t i = 0; i <= parsiz; ++i) {
					freq_sum[2 * i] += freq_data[2 * i] * ffir[2 * i] - freq_data[2 * i + 1] * ffir[2 * i + 1];
					freq_sum[2 * i + 1] += freq_data[2 * i] * ffir[2 * i + 1] + freq_data[2 * i + 1] * ffir[2 * i];
				}
			}
			orc_fft_c2r ((int)self->fftlen, freq_sum, time_data); /* :657 */
			for (uint32_t i = 0; i < parsiz; ++i) {               /* :660-662 */
				c->buf_out[i] += time_data[i];
			}

			int32_t off = overlap - (int32_t)firlat; /* :665-668 */
			if (off < 0) {
				off += (int32_t)firlen;
			}
			float* in  = &c->buf_src[off];
			float* out = c->buf_out;

			if (target_angle != angle) { /* :673-709 */
				float da = (target_angle - angle);
				if (fabs (da) > 0.5) {
					if (da < 0) {
						da += 1.f;
					} else {
						da -= 1.f;
					}
				}
				da *= self->interp_nm;
				int         final  = 0;
				const float thresh = self->interp_th;
				if (da > thresh) {
					da = thresh;
				} else if (da < -thresh) {
					da = -thresh;
				} else {
					final = 1;
				}
				for (uint32_t i = 0; i < parsiz; ++i) {
					float ca, sa;
					lv2_sin_cos (angle, &sa, &ca);
					volatile float p0 = ca * in[i];
					volatile float p1 = sa * out[i];
					out[i]            = p0 + p1;
					angle += da;
				}
				if (final) {
					angle = target_angle;
				}
				if (angle == target_angle) {
					lv2_sin_cos (angle, &c->sa, &c->ca);
				}
			} else { /* :710-717 */
				const float sa = c->sa;
				const float ca = c->ca;
				for (uint32_t i = 0; i < parsiz; ++i) {
					volatile float p0 = ca * in[i];
					volatile float p1 = sa * out[i];
					out[i]            = p0 + p1;
				}
			}
			overlap = (int32_t)(((uint32_t)overlap + parsiz) % firlen); /* :719 */
		}
	}
	c->overlap = overlap; /* :723-725 */
	c->offset  = offset;
	c->angle   = angle;

	float lvl_out = lv2_meter_run (self, c, p_out, n_samples, 1); /* :728 */

	float lvl_diff = 1.0; /* :730-739 */
	if (c->mtr_momentary[0] > 0.001f && c->mtr_momentary[1] > 0.001f) {
		lvl_diff = c->mtr_momentary[1] / c->mtr_momentary[0];
		if (lvl_diff < c->mtr_diff[0]) {
			c->mtr_diff[0] = lvl_diff;
		}
		if (lvl_diff > c->mtr_diff[1]) {
			c->mtr_diff[1] = lvl_diff;
		}
	}
	if (lvl) {
		lvl[0] = lvl_in;
		lvl[1] = lvl_out;
	}
}

void
orc_lv2_run (orc_lv2* self, const float* const* in, float* const* out, uint32_t n_samples, const float* angle_deg, float (*lvl)[2])
{
	for (uint32_t chn = 0; chn < self->n_chn; ++chn) { /* :780-785 */
		if (in[chn] != out[chn]) {
			memcpy (out[chn], in[chn], sizeof (float) * n_samples);
		}
	}
	if (self->mtr_fpp != n_samples) { /* :833-838 */
		const float fall  = 15.0f;
		const float tme   = (float)n_samples / self->rate;
		self->mtr_falloff = powf (10.0f, -0.05f * fall * tme);
		self->mtr_fpp     = n_samples;
	}
	for (uint32_t chn = 0; chn < self->n_chn; ++chn) { /* :841-843 */
		lv2_process_channel (self, chn, out[chn], angle_deg[chn], n_samples, lvl ? lvl[chn] : NULL);
	}
}

void
orc_lv2_meters (const orc_lv2* self, uint32_t chn, float* momentary2, float* peak2, float* diff2)
{
	const lv2_channel* c = &self->channel[chn];
	for (int m = 0; m < 2; ++m) {
		momentary2[m] = c->mtr_momentary[m];
		peak2[m]      = c->mtr_peak[m];
		diff2[m]      = c->mtr_diff[m];
	}
}
