// prot_kernels_f16.hip -- the fp16 sample path of the 4096-point OLA
// phase-rotate (BASELINE config 5): half-precision samples in HBM (2 B in,
// 2 B out per frame), packed-half radix-16 butterflies (one complex value =
// one VGPR, v_pk_*_f16 = one instruction per complex add), fp32 accumulate for
// the overlap-add tail and the rotation.  Same decomposition, LDS exchange
// pattern and thread<->frame map as ola4096_kernel (prot_kernels.hip).
// Stateless, steady rotation only: it is the throughput/tolerance variant of
// the roofline path, not a second implementation of the plugin semantics.
#include "prot_fft.h"
#include "prot_kernels.h"


namespace prot {

#define PROT_DEVH static __device__ __forceinline__

PROT_DEVH h2 mk (float x, float y) { return pk16::mk (x, y); }
template <int DIR> PROT_DEVH h2 cmulh (h2 a, h2 w) { return pk16::cmul<DIR> (a, w); }
template <int DIR> PROT_DEVH void fft16h (h2 (&v)[16]) { pkt_fft16<pk16, DIR> (v); }

// LDS exchange strides in 4-byte slots: ds_read/write_b32 use 32 banks of 4 B; a wave is
// served in two 32-lane groups.
constexpr int kH1 = 272; // pass1<->pass2: slot = k0*kH1 + t; reads see 16 n0 x 2 k0 -> banks n0 + 16 (k0 & 1)
constexpr int kH2 = 258; // pass2<->pass3: slot = n0*kH2 + k0 + 16*k1 -> banks 2 n0 + (k0 & 1)

PROT_DEVH float ldh (const _Float16* __restrict__ in, int64_t n_in, int64_t i)
{
	return (i >= 0 && i < n_in) ? (float)in[i] : 0.f;
}

template <int HOPSEG>
__global__ __launch_bounds__ (256, 4) void ola4096h_kernel (const OlaParamsH p)
{
	constexpr int HOP  = HOPSEG * 256;
	constexpr int NS   = 16 / HOPSEG;
	constexpr int NH   = NS - 1;
	constexpr int NACC = (NS - 1) * HOPSEG;

	__shared__ h2 lds[16 * kH1];
	__shared__ h2 twm[256];
	__shared__ h2 Hs[2049];

	const int     t  = threadIdx.x;
	const int     c  = blockIdx.y;
	const int64_t r0 = (int64_t)blockIdx.x * p.run_frames;
	if (r0 >= p.n_out) {
		return;
	}
	const int64_t r1 = (r0 + p.run_frames < p.n_out) ? r0 + p.run_frames : p.n_out;

	const _Float16* __restrict__ in = p.in + (int64_t)c * p.in_stride;
	_Float16* __restrict__ out      = p.out + (int64_t)c * p.out_stride;
	const int64_t n_in            = p.n_in;
	const int     Dc = p.conv_delay, Dd = p.dry_delay;

	h2 tw1[16];
#pragma unroll
	for (int k = 1; k < 16; ++k) {
		const float2 w = p.tw[(t * k) & 4095];
		tw1[k]         = mk (w.x, w.y);
	}
	{
		const float2 w = p.tw[16 * (((t & 15) * (t >> 4)) & 255)];
		twm[t]         = mk (w.x, w.y);
	}
	for (int i = t; i <= 2048; i += 256) {
		const float2 hh = p.H[i]; // |H|/4096 = 2.4e-4: a normal fp16 number
		Hs[i]           = mk (hh.x, hh.y);
	}
	const float ca = p.rot ? p.rot[2 * c] : p.rot_v[2 * c], sa = p.rot ? p.rot[2 * c + 1] : p.rot_v[2 * c + 1];

	float acc[NACC];
#pragma unroll
	for (int i = 0; i < NACC; ++i) {
		acc[i] = 0.f;
	}
	const int tl = t & 15, th = t >> 4;
	const int n_pairs = (NH + 1) / 2 + (int)((r1 - r0 + 2 * HOP - 1) / (2 * HOP));
	int64_t   sA      = r0 - (int64_t)(NH + 1) * HOP;
	__syncthreads ();

	for (int it = 0; it < n_pairs; ++it, sA += 2 * HOP) {
		h2 v[16];
		{
			const int64_t iA   = sA - Dc;
			const bool    fast = (iA >= 0) && (iA + 2 * HOP <= n_in);
			if (fast) {
				const _Float16* __restrict__ q = in + iA + t;
#pragma unroll
				for (int k = 0; k < HOPSEG; ++k) {
					v[k] = h2{ q[256 * k], q[HOP + 256 * k] };
				}
			} else {
#pragma unroll
				for (int k = 0; k < HOPSEG; ++k) {
					v[k] = mk (ldh (in, n_in, iA + t + 256 * k), ldh (in, n_in, iA + HOP + t + 256 * k));
				}
			}
			if (it == 0) {
#pragma unroll
				for (int k = 0; k < HOPSEG; ++k) {
					v[k].x = (_Float16)0;
				}
			}
#pragma unroll
			for (int k = HOPSEG; k < 16; ++k) {
				v[k] = h2{ (_Float16)0, (_Float16)0 };
			}
		}
		// inputs |x| <= 1: the 1024/2048-term sums stay far below the fp16 maximum (65504)
		pkt_fft16_fwd_pruned<pk16, HOPSEG> (v);
#pragma unroll
		for (int k = 1; k < 16; ++k) {
			v[k] = cmulh<1> (v[k], tw1[k]);
		}
		{
			h2* w = &lds[t];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				w[k * kH1] = v[k];
			}
			__syncthreads ();
			const h2* r = &lds[th * kH1 + tl];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				v[k] = r[16 * k];
			}
		}
		fft16h<1> (v);
#pragma unroll
		for (int k = 1; k < 16; ++k) {
			v[k] = cmulh<1> (v[k], twm[k * 16 + tl]);
		}
		__syncthreads ();
		{
			h2* w = &lds[tl * kH2 + th];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				w[16 * k] = v[k];
			}
			__syncthreads ();
			const h2* r = &lds[t];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				v[k] = r[k * kH2];
			}
		}
		fft16h<1> (v);
		// x H / 4096
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			v[k] = cmulh<1> (v[k], Hs[t + 256 * k]);
		}
#pragma unroll
		for (int k = 8; k < 16; ++k) {
			v[k] = cmulh<-1> (v[k], Hs[256 * (16 - k) - t]);
		}
		fft16h<-1> (v);
#pragma unroll
		for (int k = 1; k < 16; ++k) {
			v[k] = cmulh<-1> (v[k], twm[th * 16 + k]);
		}
		{
			h2* w = &lds[t]; // same slots this thread just read (n0*kH2 + t)
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				w[k * kH2] = v[k];
			}
			__syncthreads ();
			const h2* r = &lds[tl * kH2 + th];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				v[k] = r[16 * k];
			}
		}
		fft16h<-1> (v);
		__syncthreads ();
		{
			h2* w = &lds[th * kH1 + tl];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				w[16 * k] = v[k];
			}
			__syncthreads ();
			const h2* r = &lds[t];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				v[k] = r[k * kH1];
			}
		}
#pragma unroll
		for (int k = 1; k < 16; ++k) {
			v[k] = cmulh<-1> (v[k], tw1[k]);
		}
		fft16h<-1> (v);

		// overlap-add and rotation in fp32
		float u[2 * HOPSEG];
#pragma unroll
		for (int j = 0; j < HOPSEG; ++j) {
			u[j]          = acc[j] + (float)v[j].x;
			u[HOPSEG + j] = (float)v[HOPSEG + j].x + (float)v[j].y;
			if (NS > 2) {
				u[HOPSEG + j] += acc[HOPSEG + j];
			}
		}
#pragma unroll
		for (int q = 0; q < NS - 1; ++q) {
#pragma unroll
			for (int j = 0; j < HOPSEG; ++j) {
				float a = (float)v[(q + 1) * HOPSEG + j].y;
				if (q + 2 < NS) {
					a += (float)v[(q + 2) * HOPSEG + j].x;
				}
				if (q + 2 <= NS - 2) {
					a += acc[(q + 2) * HOPSEG + j];
				}
				acc[q * HOPSEG + j] = a;
			}
		}
		if (sA >= r0) {
#pragma unroll
			for (int k = 0; k < 2 * HOPSEG; ++k) {
				const int64_t pos = sA + t + 256 * k;
				if (pos < r1) {
					const float dry = ldh (in, n_in, pos - Dd);
					out[pos]        = (_Float16)(ca * dry + sa * u[k]);
				}
			}
		}
	}
}

hipError_t launch_ola4096_f16 (const OlaParamsH& p, int hopseg, int n_chn, hipStream_t s)
{
	if (p.n_out <= 0) {
		return hipSuccess;
	}
	const int64_t runs = (p.n_out + p.run_frames - 1) / p.run_frames;
	dim3          grid ((unsigned)runs, (unsigned)n_chn), block (256);
	if (hopseg == 4) {
		hipLaunchKernelGGL ((ola4096h_kernel<4>), grid, block, 0, s, p);
	} else if (hopseg == 8) {
		hipLaunchKernelGGL ((ola4096h_kernel<8>), grid, block, 0, s, p);
	} else {
		return hipErrorInvalidValue;
	}
	return hipGetLastError ();
}

} // namespace prot
