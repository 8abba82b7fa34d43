// membench.hip -- calibration microbenchmarks for the memory access shapes the DCT passes use
// (not part of the product): what bandwidth can a pass of each shape reach on this MI355X when it
// does no arithmetic?  Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void copy_linear(float4 *dst, const float4 *src, size_t n4)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
		float4 v = src[i]; v.x += 1.f; dst[i] = v;
	}
}

// one workgroup per row of `rowlen` floats: load all to LDS, barrier, store back
__global__ void row_lds(float *buf, int rowlen)
{
	extern __shared__ float4 lds4[];
	float4 *row = reinterpret_cast<float4 *>(buf + (size_t)blockIdx.x * rowlen);
	for (int i = threadIdx.x; i < rowlen / 4; i += blockDim.x) lds4[i] = row[i];
	__syncthreads();
	for (int i = threadIdx.x; i < rowlen / 4; i += blockDim.x) { float4 v = lds4[i]; v.x += 1.f; row[i] = v; }
}

// column tile: K floats x H rows, row pitch `pitch` floats; VEC = floats per lane per access
template <int VEC>
__global__ void col_tile(float *buf, int H, int K, long long pitch, int remap)
{
	extern __shared__ float ldsf[];
	int t = blockIdx.x;
	if (remap) { int n = gridDim.x; int per = n / 8; if (n % 8 == 0) t = (blockIdx.x % 8) * per + blockIdx.x / 8; }
	float *base = buf + (size_t)t * K;
	const int per_row = K / VEC;
	for (int it = threadIdx.x; it < H * per_row; it += blockDim.x) {
		int y = it / per_row, j = it - y * per_row;
		const float *p = base + (size_t)y * pitch + j * VEC;
		if (VEC == 4) reinterpret_cast<float4 *>(ldsf)[it] = *reinterpret_cast<const float4 *>(p);
		else if (VEC == 2) reinterpret_cast<float2 *>(ldsf)[it] = *reinterpret_cast<const float2 *>(p);
		else ldsf[it] = *p;
	}
	__syncthreads();
	for (int it = threadIdx.x; it < H * per_row; it += blockDim.x) {
		int y = it / per_row, j = it - y * per_row;
		float *p = base + (size_t)y * pitch + j * VEC;
		if (VEC == 4) { float4 v = reinterpret_cast<float4 *>(ldsf)[it]; v.x += 1.f; *reinterpret_cast<float4 *>(p) = v; }
		else if (VEC == 2) { float2 v = reinterpret_cast<float2 *>(ldsf)[it]; v.x += 1.f; *reinterpret_cast<float2 *>(p) = v; }
		else *p = ldsf[it] + 1.f;
	}
}

template <class F>
static double time_ms(F f, int iters)
{
	hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
	for (int i = 0; i < 5; i++) f();
	CHK(hipEventRecord(a));
	for (int i = 0; i < iters; i++) f();
	CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
	float ms; CHK(hipEventElapsedTime(&ms, a, b));
	return ms / iters;
}

int main()
{
	const int H = 2160, W = 3840, C = 3;
	const size_t n = (size_t)H * W * C;
	float *frame, *big;
	CHK(hipMalloc(&frame, n * 4));
	const size_t nbig = (size_t)512 << 20;   // 2 GiB of floats
	CHK(hipMalloc(&big, nbig * 4));
	CHK(hipMemset(frame, 0, n * 4)); CHK(hipMemset(big, 0, nbig * 4));
	printf("shape,bytes_moved_MB,ms,GBps\n");
	auto report = [&](const char *name, double bytes, double ms) { printf("%s,%.1f,%.4f,%.1f\n", name, bytes / 1e6, ms, bytes / ms / 1e6); fflush(stdout); };
	{
		double ms = time_ms([&] { hipLaunchKernelGGL(copy_linear, dim3(2048), dim3(256), 0, 0, (float4 *)frame, (const float4 *)frame, n / 4); }, 50);
		report("linear inplace 99.5MB (cache-resident)", 2.0 * n * 4, ms);
		ms = time_ms([&] { hipLaunchKernelGGL(copy_linear, dim3(2048), dim3(256), 0, 0, (float4 *)big, (const float4 *)big, nbig / 4); }, 10);
		report("linear inplace 2GiB (HBM)", 2.0 * nbig * 4, ms);
		ms = time_ms([&] { hipLaunchKernelGGL(copy_linear, dim3(2048), dim3(256), 0, 0, (float4 *)big, (const float4 *)(big + nbig / 2), nbig / 8); }, 10);
		report("linear copy 1GiB->1GiB (HBM)", 1.0 * nbig * 4, ms);
	}
	{
		const int rowlen = W * C;
		CHK(hipFuncSetAttribute((const void *)row_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		for (int thr : {256, 512}) {
			double ms = time_ms([&] { hipLaunchKernelGGL(row_lds, dim3(H), dim3(thr), rowlen * 4, 0, frame, rowlen); }, 50);
			char nm[128]; snprintf(nm, sizeof nm, "row via LDS 46KB/WG thr=%d (frame)", thr); report(nm, 2.0 * n * 4, ms);
		}
		// many frames (HBM-resident working set): 20 frames = 2 GB
		double ms = time_ms([&] { hipLaunchKernelGGL(row_lds, dim3(H * 20), dim3(256), rowlen * 4, 0, big, rowlen); }, 5);
		report("row via LDS 46KB/WG thr=256 (20 frames, HBM)", 2.0 * n * 4 * 20, ms);
	}
	{
		const long long pitch = (long long)W * C;
		CHK(hipFuncSetAttribute((const void *)col_tile<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		CHK(hipFuncSetAttribute((const void *)col_tile<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
		for (int remap = 0; remap < 2; remap++)
			for (int K : {4, 8, 16}) {
				for (int thr : {256, 512, 1024}) {
					int ntiles = (int)(pitch / K);
					double ms = time_ms([&] { hipLaunchKernelGGL(col_tile<4>, dim3(ntiles), dim3(thr), (size_t)H * K * 4, 0, frame, H, K, pitch, remap); }, 30);
					char nm[128]; snprintf(nm, sizeof nm, "col tile K=%d float4 thr=%d remap=%d (frame)", K, thr, remap); report(nm, 2.0 * n * 4, ms);
				}
			}
		for (int K : {8, 16}) {
			int ntiles = (int)(pitch / K);
			double ms = time_ms([&] { hipLaunchKernelGGL(col_tile<2>, dim3(ntiles), dim3(512), (size_t)H * K * 4, 0, frame, H, K, pitch, 1); }, 30);
			char nm[128]; snprintf(nm, sizeof nm, "col tile K=%d float2 thr=512 remap=1 (frame)", K); report(nm, 2.0 * n * 4, ms);
		}
		// HBM-resident: 20 frames stacked => treat as one image with 20x rows? keep pitch, tiles over the first frame only but H*20 rows is too much LDS;
		// instead run the K=16 tile kernel over 20 different frames back to back (working set 2 GB)
		{
			int K = 16, ntiles = (int)(pitch / K);
			double ms = time_ms([&] { for (int f = 0; f < 20; f++) hipLaunchKernelGGL(col_tile<4>, dim3(ntiles), dim3(512), (size_t)H * K * 4, 0, big + (size_t)f * n, H, K, pitch, 1); }, 3);
			report("col tile K=16 float4 thr=512 remap=1 (20 frames, HBM)", 2.0 * n * 4 * 20, ms);
		}
	}
	return 0;
}
