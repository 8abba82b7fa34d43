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
// the same with fewer independent chains per wave (NCH) and add / mul instead of fma: what a butterfly's dependency chains see
template <int NCH, int OP, bool PACKED>
__global__ void k_chain(float *out, float a, float b)
{
	typedef float pk2v __attribute__((ext_vector_type(2)));
	if constexpr (PACKED) {
		pk2v acc[NCH];
		for (int i = 0; i < NCH; i++) acc[i] = pk2v{(float)threadIdx.x + i, (float)threadIdx.x - i};
		const pk2v av = {a, a}, bv = {b, b};
		for (int it = 0; it < ITER; it++)
#pragma unroll
			for (int i = 0; i < NCH; i++) acc[i] = OP == 0 ? __builtin_elementwise_fma(acc[i], av, bv) : OP == 1 ? acc[i] + bv : acc[i] * av;
		float s = 0; for (int i = 0; i < NCH; i++) s += acc[i].x + acc[i].y;
		out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	} else {
		float acc[2 * NCH];
		for (int i = 0; i < 2 * NCH; i++) acc[i] = threadIdx.x + i;
		for (int it = 0; it < ITER; it++)
#pragma unroll
			for (int i = 0; i < 2 * NCH; i++) acc[i] = OP == 0 ? __builtin_fmaf(acc[i], a, b) : OP == 1 ? acc[i] + b : acc[i] * a;
		float s = 0; for (int i = 0; i < 2 * NCH; i++) s += acc[i];
		out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	}
}
template <int NCH, int OP>
static int chain_case(float *d, hipEvent_t e0, hipEvent_t e1, int waves)
{
	static const char *opn[] = {"fma", "add", "mul"};
	float t[2];
	for (int mode = 0; mode < 2; mode++) {
		float best = 1e9;
		for (int rep = 0; rep < 4; rep++) {
			CHK(hipEventRecord(e0));
			if (mode == 0) hipLaunchKernelGGL((k_chain<NCH, OP, false>), dim3(256 * 4), dim3(256 * waves), 0, 0, d, 1.0001f, 0.5f);
			else hipLaunchKernelGGL((k_chain<NCH, OP, true>), dim3(256 * 4), dim3(256 * waves), 0, 0, d, 1.0001f, 0.5f);
			CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
			float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
		}
		t[mode] = best;
	}
	// element-operations per second: 2 NCH per lane per iteration either way
	const double ops = 2.0 * NCH * ITER * 256.0 * waves * 256 * 4;
	printf("%s, %d independent chains of pairs, %d wave(s)/SIMD: scalar %.3f ms (%.1f Gop/s), packed %.3f ms (%.1f Gop/s), packed/scalar speed %.2f\n",
	       opn[OP], NCH, waves, t[0], ops / t[0] / 1e6, t[1], ops / t[1] / 1e6, t[0] / t[1]);
	return 0;
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
	for (int waves = 1; waves <= 4; waves *= 4) {
		if (chain_case<1, 0>(d, e0, e1, waves) || chain_case<2, 0>(d, e0, e1, waves) || chain_case<4, 0>(d, e0, e1, waves)) return 1;
		if (chain_case<1, 1>(d, e0, e1, waves) || chain_case<4, 1>(d, e0, e1, waves)) return 1;
		if (chain_case<1, 2>(d, e0, e1, waves) || chain_case<4, 2>(d, e0, e1, waves)) return 1;
	}
	return 0;
}
