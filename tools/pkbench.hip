// pkbench.hip -- issue rate of packed (v_pk_fma_f32) against scalar (v_fma_f32) FP32 on gfx950, all SIMDs busy.
// Answers whether carrying two signals per lane through packed instructions can shorten a VALU-bound kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float pk2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096, NACC = 8;
__global__ void k_scalar(float *out, float a, float b)
{
	float acc[2 * NACC];
	for (int i = 0; i < 2 * NACC; i++) acc[i] = threadIdx.x + i;
	for (int it = 0; it < ITER; it++)
#pragma unroll
		for (int i = 0; i < 2 * NACC; i++) acc[i] = __builtin_fmaf(acc[i], a, b);
	float s = 0; for (int i = 0; i < 2 * NACC; i++) s += acc[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_packed(float *out, float a, float b)
{
	pk2 acc[NACC];
	for (int i = 0; i < NACC; i++) acc[i] = pk2{(float)threadIdx.x + i, (float)threadIdx.x - i};
	const pk2 av = {a, a}, bv = {b, b};
	for (int it = 0; it < ITER; it++)
#pragma unroll
		for (int i = 0; i < NACC; i++) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
	float s = 0; for (int i = 0; i < NACC; i++) s += acc[i].x + acc[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
	float *d; CHK(hipMalloc(&d, 4 * 256 * 16 * 256));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	for (int waves = 1; waves <= 4; waves *= 2) {
		for (int mode = 0; mode < 2; mode++) {
			float best = 1e9;
			for (int rep = 0; rep < 5; rep++) {
				CHK(hipEventRecord(e0));
				if (mode == 0) hipLaunchKernelGGL(k_scalar, dim3(256 * 4), dim3(256 * waves), 0, 0, d, 1.0001f, 0.5f);
				else hipLaunchKernelGGL(k_packed, dim3(256 * 4), dim3(256 * waves), 0, 0, d, 1.0001f, 0.5f);
				CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
				float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
			}
			const double flop = 2.0 * 2 * NACC * ITER * 256.0 * waves * 256 * 4;
			printf("%s %d wave(s)/SIMD per workgroup: %.3f ms, %.1f TFLOP/s\n", mode ? "v_pk_fma_f32" : "v_fma_f32   ", waves, best, flop / best / 1e9);
		}
	}
	return 0;
}
