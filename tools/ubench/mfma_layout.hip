// operand layout check of v_mfma_f32_32x32x16_f16 on gfx950 (what prot_f16mfma.hip relies on):
//   A (32 x 16): lane l holds A[l & 31][8 (l >> 5) + i], i = 0..7;  B (16 x 32): lane l holds B[8 (l >> 5) + i][l & 31];
//   D (32 x 32): lane l, register r holds D[8 (r >> 2) + 4 (l >> 5) + (r & 3)][l & 31].
// hipcc --offload-arch=gfx950 -O2 -o mfma_layout mfma_layout.hip && ./mfma_layout   -> "layout ok"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__ ((ext_vector_type (16)));
typedef _Float16 h8 __attribute__ ((ext_vector_type (8)));
__global__ void k (const float* A, const float* B, float* D)
{
	const int l = threadIdx.x;
	h8 a, b;
	for (int i = 0; i < 8; ++i) {
		a[i] = (_Float16)A[(l & 31) * 16 + 8 * (l >> 5) + i];
		b[i] = (_Float16)B[(8 * (l >> 5) + i) * 32 + (l & 31)];
	}
	f32x16 z;
	for (int r = 0; r < 16; ++r) {
		z[r] = 0.f;
	}
	f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16 (a, b, z, 0, 0, 0);
	for (int r = 0; r < 16; ++r) {
		D[(8 * (r >> 2) + 4 * (l >> 5) + (r & 3)) * 32 + (l & 31)] = d[r];
	}
}
int main ()
{
	float hA[32 * 16], hB[16 * 32], hD[32 * 32], ref[32 * 32];
	srand (1);
	for (int i = 0; i < 512; ++i) {
		hA[i] = (float)(rand () % 9 - 4);
		hB[i] = (float)(rand () % 9 - 4);
	}
	for (int m = 0; m < 32; ++m) {
		for (int n = 0; n < 32; ++n) {
			float s = 0;
			for (int kk = 0; kk < 16; ++kk) {
				s += hA[m * 16 + kk] * hB[kk * 32 + n];
			}
			ref[m * 32 + n] = s;
		}
	}
	float *dA, *dB, *dD;
	hipMalloc (&dA, sizeof hA); hipMalloc (&dB, sizeof hB); hipMalloc (&dD, sizeof hD);
	hipMemcpy (dA, hA, sizeof hA, hipMemcpyHostToDevice);
	hipMemcpy (dB, hB, sizeof hB, hipMemcpyHostToDevice);
	hipLaunchKernelGGL (k, dim3 (1), dim3 (64), 0, 0, dA, dB, dD);
	hipMemcpy (hD, dD, sizeof hD, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int i = 0; i < 1024; ++i) {
		bad += hD[i] != ref[i];
	}
	printf (bad ? "layout WRONG: %d of 1024 entries differ\n" : "layout ok\n", bad);
	return bad != 0;
}
