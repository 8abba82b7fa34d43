// membench2.hip -- round-2 calibration of the MI355X memory system for the access shapes of the DCT passes
// (not part of the product).  Answers, with no arithmetic in the way:
//   * what do plain streams reach (read-only / write-only / copy), Infinity-Cache-resident (one 99.5 MB frame reused)
//     and HBM-resident (20 distinct frames), as a function of loads in flight per lane and grid size?
//   * what does a strided tile of (rows x SEG floats) reach, for the tile shapes a two-level column split could use:
//     2160x8 (32-B segments), 1080x16 (64 B), 540x32 (128 B), 270x64 (256 B) -- read side and write side separately
//   * register staging vs LDS-DMA (global_load_lds_dwordx4) for the tile load
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/membench2.hip -o tools/membench2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// ---- plain streams: each workgroup owns a contiguous chunk; U float4 in flight per lane ----------------------
template <int U, int MODE>   // MODE 0 copy, 1 read-only, 2 write-only, 3 copy with nontemporal stores, 4 copy nt loads+stores
__global__ void __launch_bounds__(256) stream_k(f4 *dst, const f4 *src, size_t n4, float *sink)
{
	const size_t per = (size_t)256 * U;
	f4 acc = {0, 0, 0, 0};
	for (size_t base = (size_t)blockIdx.x * per; base < n4; base += (size_t)gridDim.x * per) {
		f4 v[U];
		if (MODE != 2) {
#pragma unroll
			for (int u = 0; u < U; u++) {
				const size_t i = base + (size_t)u * 256 + threadIdx.x;
				if (MODE == 4) v[u] = __builtin_nontemporal_load(src + i); else v[u] = src[i];
			}
		} else {
#pragma unroll
			for (int u = 0; u < U; u++) v[u] = f4{1.f, 2.f, 3.f, (float)u};
		}
		if (MODE == 1) {
#pragma unroll
			for (int u = 0; u < U; u++) acc += v[u];
		} else {
#pragma unroll
			for (int u = 0; u < U; u++) {
				const size_t i = base + (size_t)u * 256 + threadIdx.x;
				f4 w = v[u]; w.x += 1.f;
				if (MODE >= 3) __builtin_nontemporal_store(w, dst + i); else dst[i] = w;
			}
		}
	}
	if (MODE == 1 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

// ---- strided tiles: ROWS x SEG floats, row pitch `pitch` floats, staged through LDS -----------------------------
// MODE 0: load tile -> LDS -> barrier -> store tile (in place); 1: load only; 2: store only
// DMA 1: the load goes global -> LDS directly (global_load_lds_dwordx4, 64 lanes x 16 B = 1 KiB of LDS per instruction)
__device__ __forceinline__ int xcd_remap(int bid, int n)
{
	const int per = n >> 3, full = per << 3;
	if (bid >= full) return bid;
	return (bid & 7) * per + (bid >> 3);
}

template <int ROWS, int SEG, int T, int MODE, int DMA>
__global__ void __launch_bounds__(T) tile_k(float *buf, long long pitch, int tiles_per_row, int ntiles, long long frame_stride, float *sink)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	f4 *lds = reinterpret_cast<f4 *>(lds_raw);
	constexpr int Q = SEG / 4;               // float4 per row segment
	constexpr int ITEMS = ROWS * Q;
	constexpr int ROUNDS = (ITEMS + T - 1) / T;
	const int frame = blockIdx.x / ntiles;
	int t = blockIdx.x - frame * ntiles;
	t = xcd_remap(t, ntiles);
	const int rb = t / tiles_per_row, cb = t - rb * tiles_per_row;   // row block, column block
	float *base = buf + (long long)frame * frame_stride + (long long)rb * ROWS * pitch + (long long)cb * SEG;
	const int tid = threadIdx.x;
	if (MODE != 2) {
		if (DMA) {
#pragma unroll
			for (int r = 0; r < ROUNDS; r++) {
				const int it = tid + r * T;
				if (it < ITEMS) {
					const int y = it / Q, j = it - y * Q;
					// LDS destination of a wave instruction = M0 base + lane * 16: lanes of one wave are consecutive items
					__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (long long)y * pitch + 4 * j),
					                                 (__attribute__((address_space(3))) void *)(lds + (it & ~63)), 16, 0, 0);
				}
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		} else {
			f4 v[ROUNDS];
#pragma unroll
			for (int r = 0; r < ROUNDS; r++) {
				const int it = tid + r * T;
				if (it < ITEMS) { const int y = it / Q, j = it - y * Q; v[r] = *reinterpret_cast<const f4 *>(base + (long long)y * pitch + 4 * j); }
			}
#pragma unroll
			for (int r = 0; r < ROUNDS; r++) { const int it = tid + r * T; if (it < ITEMS) lds[it] = v[r]; }
		}
	} else {
#pragma unroll
		for (int r = 0; r < ROUNDS; r++) { const int it = tid + r * T; if (it < ITEMS) lds[it] = f4{1.f, 2.f, 3.f, (float)it}; }
	}
	__syncthreads();
	if (MODE != 1) {
#pragma unroll
		for (int r = 0; r < ROUNDS; r++) {
			const int it = tid + r * T;
			if (it < ITEMS) {
				const int y = it / Q, j = it - y * Q;
				f4 w = lds[it]; w.x += 1.f;
				*reinterpret_cast<f4 *>(base + (long long)y * pitch + 4 * j) = w;
			}
		}
	} else {
		f4 a = lds[tid];
		if (a.x == 12345.678f) *sink = a.y;
	}
}

template <class F>
static double time_us(F f, int iters)
{
	hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
	for (int i = 0; i < 3; i++) f(i);
	CHK(hipEventRecord(a));
	for (int i = 0; i < iters; i++) f(i);
	CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
	float ms; CHK(hipEventElapsedTime(&ms, a, b));
	CHK(hipGetLastError());
	CHK(hipEventDestroy(a)); CHK(hipEventDestroy(b));
	return ms * 1000.0 / iters;
}

static const int H = 2160, W = 3840, C = 3;
static const size_t NF = (size_t)H * W * C;       // floats per frame
static float *g_buf, *g_sink;
static const int NFR = 20;

static void report(const char *name, const char *where, double bytes, double us)
{
	printf("%-58s,%-6s,%9.1f,%8.2f,%8.1f\n", name, where, bytes / 1e6, us, bytes / us / 1e6);
	fflush(stdout);
}

template <int U, int MODE>
static void run_stream(const char *name, int blocks_per_cu)
{
	const int grid = 256 * blocks_per_cu;
	const double mult = (MODE == 1 || MODE == 2) ? 1.0 : 2.0;
	char nm[128];
	// Infinity-Cache-resident: the same frame (copy: in place)
	snprintf(nm, sizeof nm, "%s U=%d grid=256x%d", name, U, blocks_per_cu);
	double us = time_us([&](int) { hipLaunchKernelGGL((stream_k<U, MODE>), dim3(grid), dim3(256), 0, 0, (f4 *)g_buf, (const f4 *)g_buf, NF / 4, g_sink); }, 40);
	report(nm, "1frame", mult * NF * 4, us);
	// HBM-resident: rotate over 20 frames
	us = time_us([&](int i) { float *p = g_buf + (size_t)(i % NFR) * NF; hipLaunchKernelGGL((stream_k<U, MODE>), dim3(grid), dim3(256), 0, 0, (f4 *)p, (const f4 *)p, NF / 4, g_sink); }, 40);
	report(nm, "20fr", mult * NF * 4, us);
	// one long launch over all 20 frames
	us = time_us([&](int) { hipLaunchKernelGGL((stream_k<U, MODE>), dim3(grid), dim3(256), 0, 0, (f4 *)g_buf, (const f4 *)g_buf, NF / 4 * NFR, g_sink); }, 4);
	report(nm, "2GB", mult * NF * 4 * NFR, us);
}

template <int ROWS, int SEG, int T, int MODE, int DMA>
static void run_tile(const char *name)
{
	const long long pitch = (long long)W * C;
	const int tiles_per_row = (int)(pitch / SEG), ntiles = tiles_per_row * (H / ROWS);
	const size_t lds = (size_t)ROWS * SEG * 4;
	CHK(hipFuncSetAttribute((const void *)tile_k<ROWS, SEG, T, MODE, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	int occ = 0; CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)tile_k<ROWS, SEG, T, MODE, DMA>, T, lds));
	const double mult = MODE == 0 ? 2.0 : 1.0;
	char nm[160];
	snprintf(nm, sizeof nm, "%s %dx%d T=%d %s%s occ=%d", name, ROWS, SEG, T, MODE == 0 ? "rw" : MODE == 1 ? "r" : "w", DMA ? " dma" : "", occ);
	double us = time_us([&](int) { hipLaunchKernelGGL((tile_k<ROWS, SEG, T, MODE, DMA>), dim3(ntiles), dim3(T), lds, 0, g_buf, pitch, tiles_per_row, ntiles, (long long)NF, g_sink); }, 40);
	report(nm, "1frame", mult * NF * 4, us);
	us = time_us([&](int i) { float *p = g_buf + (size_t)(i % NFR) * NF; hipLaunchKernelGGL((tile_k<ROWS, SEG, T, MODE, DMA>), dim3(ntiles), dim3(T), lds, 0, p, pitch, tiles_per_row, ntiles, (long long)NF, g_sink); }, 40);
	report(nm, "20fr", mult * NF * 4, us);
	// 2 frames per launch (a batch of frames in one grid)
	us = time_us([&](int) { hipLaunchKernelGGL((tile_k<ROWS, SEG, T, MODE, DMA>), dim3(ntiles * 2), dim3(T), lds, 0, g_buf, pitch, tiles_per_row, ntiles, (long long)NF, g_sink); }, 40);
	report(nm, "2fr/l", mult * NF * 4 * 2, us);
}

int main(int argc, char **argv)
{
	const char *only = argc > 1 ? argv[1] : "";
	CHK(hipMalloc(&g_buf, NF * 4 * NFR));
	CHK(hipMalloc(&g_sink, 64));
	{
		std::vector<float> h(NF);
		for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f;
		for (int f = 0; f < NFR; f++) CHK(hipMemcpy(g_buf + (size_t)f * NF, h.data(), NF * 4, hipMemcpyHostToDevice));
	}
	printf("shape,where,MB_moved,us,GBps\n");
	if (!*only || !strcmp(only, "stream")) {
		run_stream<1, 0>("copy", 8); run_stream<4, 0>("copy", 8); run_stream<8, 0>("copy", 8); run_stream<8, 0>("copy", 4); run_stream<16, 0>("copy", 4);
		run_stream<8, 0>("copy", 16); run_stream<8, 0>("copy", 32);
		run_stream<8, 3>("copy nt-store", 8); run_stream<8, 4>("copy nt-both", 8);
		run_stream<4, 1>("read", 8); run_stream<8, 1>("read", 8); run_stream<16, 1>("read", 4); run_stream<8, 1>("read", 32);
		run_stream<4, 2>("write", 8); run_stream<8, 2>("write", 8); run_stream<8, 2>("write", 32);
	}
	if (!*only || !strcmp(only, "tile")) {
		// register staging, read+write
		run_tile<2160, 8, 512, 0, 0>("tile"); run_tile<1080, 16, 512, 0, 0>("tile"); run_tile<540, 32, 512, 0, 0>("tile"); run_tile<270, 64, 512, 0, 0>("tile");
		run_tile<2160, 8, 256, 0, 0>("tile"); run_tile<1080, 16, 256, 0, 0>("tile"); run_tile<540, 32, 256, 0, 0>("tile");
		run_tile<2160, 16, 1024, 0, 0>("tile"); run_tile<2160, 4, 256, 0, 0>("tile"); run_tile<1080, 8, 256, 0, 0>("tile"); run_tile<540, 16, 256, 0, 0>("tile");
		run_tile<720, 24, 512, 0, 0>("tile"); run_tile<135, 128, 512, 0, 0>("tile");
		// read side / write side alone
		run_tile<2160, 8, 512, 1, 0>("tile"); run_tile<2160, 8, 512, 2, 0>("tile");
		run_tile<1080, 16, 512, 1, 0>("tile"); run_tile<1080, 16, 512, 2, 0>("tile");
		run_tile<540, 32, 512, 1, 0>("tile"); run_tile<540, 32, 512, 2, 0>("tile");
		// LDS-DMA loads
		run_tile<2160, 8, 512, 0, 1>("tile"); run_tile<2160, 8, 512, 1, 1>("tile"); run_tile<1080, 16, 512, 0, 1>("tile"); run_tile<540, 32, 512, 0, 1>("tile");
		run_tile<2160, 8, 256, 0, 1>("tile");
	}
	return 0;
}
