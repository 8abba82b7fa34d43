// mfmapeak.hip -- what v_mfma_f32_32x32x2_f32 delivers with nothing else in the way: every SIMD of the chip issuing MFMAs back to back out
// of registers (no LDS, no memory), W waves per SIMD, for a few milliseconds -- the ceiling a product kernel is priced against at the
// clock the chip actually holds under that load.  s_memtime / wall_clock64 give the shader clock of the run.
//   hipcc --offload-arch=gfx950 -O3 tools/mfmapeak.hip -o tools/mfmapeak && tools/mfmapeak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) peak(float *out, int iters, unsigned long long *clk)
{
	f32x16 acc[NACC];
	for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
	float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
	const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int u = 0; u < 16; u++)
#pragma unroll
			for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
	}
	const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
	float s = 0.f;
	for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int NACC> static void run(int wgs_per_cu, int iters)
{
	int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
	int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
	const int wgs = cus * wgs_per_cu;
	float *out; unsigned long long *clk, h[2];
	hipMalloc(&out, (size_t)wgs * 256 * 4); hipMalloc(&clk, 16);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int rep = 0; rep < 3; rep++) {
		hipEventRecord(e0);
		hipLaunchKernelGGL(peak<NACC>, dim3(wgs), dim3(256), 0, 0, out, iters, clk);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
		const double flops = (double)wgs * 4 * iters * 16 * NACC * 4096.0;
		printf("acc sets %d, %d workgroup(s) of 4 waves per CU (%d waves per SIMD): %.3f ms  %.1f TF   shader clock %.0f MHz (%llu cycles in %llu ticks of %d kHz)\n",
		       NACC, wgs_per_cu, wgs_per_cu, ms, flops / ms / 1e9, (double)h[0] / ((double)h[1] / wall_khz) / 1e3, h[0], h[1], wall_khz);
	}
	hipFree(out); hipFree(clk);
}

int main()
{
	run<4>(1, 4000);
	run<4>(2, 2000);
	run<6>(2, 1400);
	run<1>(2, 8000);
	return 0;
}
