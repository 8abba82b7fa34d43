// wgcost.hip -- what a launch of large workgroups costs before it does anything: W workgroups of T threads with L bytes of LDS that
// (a) return at once, (b) read one dword per thread and return, (c) read R dwords per thread, spread like the owner ids of a column tile.
//   hipcc --offload-arch=gfx950 -O3 tools/wgcost.hip -o tools/wgcost && tools/wgcost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#pragma clang diagnostic ignored "-Wunused-value"
template <int MODE, int T>
__global__ void __launch_bounds__(T) k(const uint32_t *ids, uint8_t *flags, int rows, int pitch, uint32_t want)
{
	extern __shared__ unsigned char lds[];
	if (MODE == 0) return;
	const int tid = threadIdx.x;
	bool hit = false;
	if (MODE == 1) hit = ids[(size_t)blockIdx.x * T + tid] == want;
	if (MODE == 2 || MODE == 3) {
		// tile b: 16 floats = 6 pixel ids of every row; thread -> (row = tid / 4 + i * T / 4, lane quarter)
		int b = blockIdx.x;
		if (MODE == 3) { const int g = 6, xcd = b & 7, slot = b >> 3; b = ((slot / g) * 8 + xcd) * g + slot % g; }      // groups of 6 neighbours on one XCD
		const int col = (b % (pitch / 6)) * 6 + (tid & 3);
		for (int r = tid >> 2; r < rows; r += T >> 2) hit |= ids[(size_t)r * pitch + col] == want;
	}
	if (MODE == 4 || MODE == 5) {
		// the half-tile REDFT01 pass as it is: tile t1 of 2 x 1440 (h = t1 / 1440), items q = tid / 4 + i * 256 < 1080: rows k = 2 q + h and N - k, lane
		// quarter jp: floats 16 t + 4 jp .. + 3 -> the ids of the first and of the last
		int b = blockIdx.x;
		if (MODE == 5) { const int g = 6, xcd = b & 7, slot = b >> 3; b = ((slot / g) * 8 + xcd) * g + slot % g; }
		const int h = b / 1440, t = b - h * 1440, jp = tid & 3, N = 4320;
		for (int q = tid >> 2; q < 1080; q += T >> 2) {
			if (h && q >= 540) break;
			const int k = 2 * q + h, km = k ? N - k : 0;
			const unsigned o = 16 * t + 4 * jp;
			hit |= ids[(size_t)k * pitch + o / 3] == want; hit |= ids[(size_t)k * pitch + (o + 3) / 3] == want;
			hit |= ids[(size_t)km * pitch + o / 3] == want; hit |= ids[(size_t)km * pitch + (o + 3) / 3] == want;
		}
	}
	const int nz = __syncthreads_or(hit);
	if (tid == 0) flags[blockIdx.x] = nz != 0;
}
template <int MODE, int T> static void run(const char *what, int wgs, size_t lds, const uint32_t *ids, uint8_t *flags, int rows, int pitch)
{
	hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<MODE, T>), dim3(wgs), dim3(T), lds, 0, ids, flags, rows, pitch, 0xffffffffu);
	hipEventRecord(e0);
	for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<MODE, T>), dim3(wgs), dim3(T), lds, 0, ids, flags, rows, pitch, 0xffffffffu);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	printf("%-44s %5d workgroups x %4d threads, %6zu B LDS: %7.1f us\n", what, wgs, T, lds, ms / 20 * 1e3);
}
int main()
{
	const int rows = 4320, pitch = 7680;
	uint32_t *ids; uint8_t *flags;
	hipMalloc(&ids, (size_t)rows * pitch * 4); hipMemset(ids, 0, (size_t)rows * pitch * 4); hipMalloc(&flags, 1 << 16);
	run<0, 1024>("returns at once", 2880, 138 * 1024 + 512, ids, flags, rows, pitch);
	run<0, 1024>("returns at once", 2880, 64 * 1024, ids, flags, rows, pitch);
	run<0, 1024>("returns at once", 2880, 0, ids, flags, rows, pitch);
	run<0, 512>("returns at once", 5760, 69 * 1024, ids, flags, rows, pitch);
	run<0, 256>("returns at once", 11520, 0, ids, flags, rows, pitch);
	run<1, 1024>("one dword per thread, barrier, flag", 2880, 138 * 1024 + 512, ids, flags, rows, pitch);
	run<1, 1024>("one dword per thread, barrier, flag", 2880, 0, ids, flags, rows, pitch);
	run<2, 1024>("ids of a 4320-row tile (half the rows)", 2880, 138 * 1024 + 512, ids, flags, rows / 2, pitch);
	run<2, 1024>("ids of a 4320-row tile (half the rows)", 2880, 0, ids, flags, rows / 2, pitch);
	run<2, 512>("ids of a 4320-row tile (half the rows)", 2880, 69 * 1024, ids, flags, rows / 2, pitch);
	run<3, 1024>("the same, 6 neighbouring tiles per XCD", 2880, 138 * 1024 + 512, ids, flags, rows / 2, pitch);
	run<3, 1024>("the same, 6 neighbouring tiles per XCD", 2880, 0, ids, flags, rows / 2, pitch);
	run<4, 1024>("ids as the half-tile REDFT01 pass reads them", 2880, 138 * 1024 + 512, ids, flags, rows, pitch);
	run<5, 1024>("the same, 6 neighbouring tiles per XCD", 2880, 138 * 1024 + 512, ids, flags, rows, pitch);
	return 0;
}
