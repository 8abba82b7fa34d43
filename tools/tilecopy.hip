// tilecopy.hip -- the memory floor of the 8K column passes' access shape: a 7680 x 4320 x 3 float frame (398 MB, HBM-resident) copied in place in
// column TILES of R rows x K floats by workgroups of T threads (lane = 16 bytes), no LDS, no arithmetic, as many workgroups per CU as fit.
//   hipcc -O3 --offload-arch=gfx950 tools/tilecopy.hip -o tools/tilecopy && tools/tilecopy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// tile t covers floats [t*K, t*K+K) of every row; SPLIT: rows y = 2n (half 0) or N-1-2n (half 1) as the half-tile kernels read them
template <int K, int T, int ROUNDS, bool SPLIT>
__global__ void __launch_bounds__(T) tile_copy(const float4 *in, float4 *out, int rows, int pitch4, int ntiles)
{
	constexpr int NP = K / 4;
	int b = blockIdx.x;
	const int per = ntiles >> 3, full = per << 3;
	if (b < full) b = (b & 7) * per + (b >> 3);          // neighbouring tiles on one XCD (xcd_remap)
	const int h = SPLIT ? b / (ntiles / 2) : 0, t = SPLIT ? b - h * (ntiles / 2) : b;
	const int R = SPLIT ? rows / 2 : rows;
	float4 v[ROUNDS];
#pragma unroll
	for (int i = 0; i < ROUNDS; i++) {
		const int it = threadIdx.x + i * T, n = it / NP, jp = it - n * NP;
		if (n < R) { const int y = SPLIT ? (h ? rows - 1 - 2 * n : 2 * n) : n; v[i] = in[(size_t)y * pitch4 + t * NP + jp]; }
	}
#pragma unroll
	for (int i = 0; i < ROUNDS; i++) {
		const int it = threadIdx.x + i * T, n = it / NP, jp = it - n * NP;
		if (n < R) { const int y = SPLIT ? (h ? rows - 1 - 2 * n : 2 * n) : n; float4 w = v[i]; w.x += 1.f; out[(size_t)y * pitch4 + t * NP + jp] = w; }
	}
}
template <int K, int T, int ROUNDS, bool SPLIT>
static void run(const char *what, float *x, int rows, int width)
{
	const int ntiles = (SPLIT ? 2 : 1) * width / K;
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto go = [&]() { hipLaunchKernelGGL((tile_copy<K, T, ROUNDS, SPLIT>), dim3(ntiles), dim3(T), 0, 0, (const float4 *)x, (float4 *)x, rows, width / 4, ntiles); };
	for (int i = 0; i < 3; i++) go();
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0, 0));
	for (int i = 0; i < 20; i++) go();
	CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	const double us = ms * 1e3 / 20;
	printf("%-72s %8.1f us  %5.2f TB/s\n", what, us, 2.0 * rows * width * 4 / us / 1e6);
}
int main()
{
	const int W = 7680 * 3, H = 4320;
	float *x; CK(hipMalloc(&x, (size_t)W * H * 4)); CK(hipMemset(x, 0, (size_t)W * H * 4));
	// ROUNDS = ceil(R * K / 4 / T)
	run<16, 1024, 9, true>("half tiles 2160 x 16 floats (64-B segments), 1024 threads [today]", x, H, W);
	run<16, 512, 17, true>("half tiles 2160 x 16, 512 threads", x, H, W);
	run<8, 512, 9, true>("half tiles 2160 x 8 floats (32-B segments), 512 threads", x, H, W);
	run<8, 1024, 5, true>("half tiles 2160 x 8, 1024 threads", x, H, W);
	run<32, 1024, 17, true>("half tiles 2160 x 32 floats (128-B segments), 1024 threads", x, H, W);
	run<8, 1024, 9, false>("whole tiles 4320 x 8 (32-B segments), 1024 threads", x, H, W);
	run<16, 1024, 17, false>("whole tiles 4320 x 16, 1024 threads", x, H, W);
	run<64, 1024, 34, true>("half tiles 2160 x 64 floats (256-B segments), 1024 threads", x, H, W);
	return 0;
}
