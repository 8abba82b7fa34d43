// kbench.hip -- experiment harness (not part of the product): times variants of the specialised
// passes (thread count, radices, tile width, resident workgroups per CU) and ablations on 4K frames.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -Idspfun_amd/csrc -Itools tools/kbench.hip -o tools/kbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "dct_spec.h"
using namespace dspfft;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ int g_stagger;   // initial delay (x 64*64 cycles) per 256-block group (experiment)
// workgroup wrappers (same control flow as backend_hip.hip); ABL=1 skips the FFT phases
template <class S, int KIND, int ABL, int WPE>
__global__ void __launch_bounds__(S::T, WPE) row_k(const PassArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	cf *planes = reinterpret_cast<cf *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	row_base(a, blockIdx.x, bin, bout);
	{ const int grp = blockIdx.x >> 8; if (g_stagger > 0 && grp >= 1 && grp <= 2) for (int i = 0; i < g_stagger * grp; i++) __builtin_amdgcn_s_sleep(64); }
	S::template prefetch<KIND>(a, bin, tid, st);
	S::template phase<KIND, 0>(a, planes, bout, tid, st);
	__syncthreads();
	static_for<1, S::NPH>([&](auto ph) {
		if constexpr (ABL == 0 || ph == S::NPH - 1) {
			S::template phase<KIND, ph>(a, planes, bout, tid, st);
			if constexpr (ph + 1 < S::NPH) __syncthreads();
		}
	});
}
template <class S, int KIND, int ABL, int WPE>
__global__ void __launch_bounds__(S::T, WPE) col_k(const PassArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	typename S::V *buf = reinterpret_cast<typename S::V *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	S::base(a, blockIdx.x, bin, bout);
	{ const int grp = blockIdx.x >> 8; if (g_stagger > 0 && grp == 1) for (int i = 0; i < g_stagger; i++) __builtin_amdgcn_s_sleep(64); }
	S::template prefetch<KIND>(a, bin, tid, st);
	S::template phase<KIND, 0>(a, buf, bout, tid, st);
	__syncthreads();
	static_for<1, S::NPH>([&](auto ph) {
		if constexpr (ABL == 0 || ph == S::NPH - 1) {
			S::template phase<KIND, ph>(a, buf, bout, tid, st);
			if constexpr (ph + 1 < S::NPH) __syncthreads();
		}
	});
}

static const int H = 2160, W = 3840, C = 3;
static float *g_frame;
static int g_frames = 1;

struct Tables { cf *T, *Wt; };
static Tables make_tables(int N, int L)
{
	std::vector<cf> T(N + 1), Wv(L);
	for (int j = 0; j <= N; j++) T[j] = cmk((float)cos(M_PI * j / (2.0 * N)), (float)-sin(M_PI * j / (2.0 * N)));
	for (int t = 0; t < L; t++) Wv[t] = cmk((float)cos(2 * M_PI * t / L), (float)-sin(2 * M_PI * t / L));
	Tables r;
	CHK(hipMalloc(&r.T, T.size() * 8)); CHK(hipMalloc(&r.Wt, Wv.size() * 8));
	CHK(hipMemcpy(r.T, T.data(), T.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(r.Wt, Wv.data(), Wv.size() * 8, hipMemcpyHostToDevice));
	return r;
}

template <class F>
static double time_us(F f, int iters = 30)
{
	hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
	for (int i = 0; i < 3; i++) f();
	CHK(hipEventRecord(a));
	for (int i = 0; i < iters; i++) f();
	CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
	float ms; CHK(hipEventElapsedTime(&ms, a, b));
	CHK(hipGetLastError());
	return ms * 1000.0 / iters;
}

template <class S, int KIND, int ABL, int WPE>
static void run_row(const char *name)
{
	static Tables tb = make_tables(S::N, S::L);
	PassArgs a; memset((void *)&a, 0, sizeof a);
	a.in = g_frame; a.out = g_frame; a.N = S::N; a.kind = KIND; a.C = S::C; a.nb0 = H; a.nb1 = g_frames; a.sb0_in = a.sb0_out = (long long)W * C;
	a.sb1_in = a.sb1_out = (long long)H * W * C;
	a.T = tb.T; a.W = tb.Wt; a.scale = 1.f / 7680.f; a.in_scale0 = a.out_scale0 = 1.f;
	const int NWORK = H * g_frames;
	CHK(hipFuncSetAttribute((const void *)row_k<S, KIND, ABL, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
	int occ = 0; CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)row_k<S, KIND, ABL, WPE>, S::T, S::LDS));
	printf("%-40s kind=%d abl=%d wpe=%d occ=%d |", name, KIND, ABL, WPE, occ);
	for (int wpc : {0}) {
		{ int sg = wpc; CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_stagger), &sg, sizeof sg)); }
		int grid = NWORK;
		double us = time_us([&] { hipLaunchKernelGGL((row_k<S, KIND, ABL, WPE>), dim3(grid), dim3(S::T), S::LDS, 0, a); }) / g_frames;
		printf(" stg%d:%6.1fus", wpc, us);
	}
	printf("\n"); fflush(stdout);
}
template <class S, int KIND, int ABL, int WPE>
static void run_col(const char *name)
{
	static Tables tb = make_tables(S::N, S::N);
	PassArgs a; memset((void *)&a, 0, sizeof a);
	a.in = g_frame; a.out = g_frame; a.N = S::N; a.kind = KIND; a.K = S::K; a.B = S::B; a.ninner = W * C; a.ntiles = W * C / S::K;
	a.es_in = a.es_out = (long long)W * C; a.nb0 = g_frames; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)H * W * C;
	a.T = tb.T; a.W = tb.Wt; a.scale = 1.f / 4320.f; a.in_scale0 = a.out_scale0 = 1.f;
	const int NWORK = a.ntiles * g_frames;
	CHK(hipFuncSetAttribute((const void *)col_k<S, KIND, ABL, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
	int occ = 0; CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)col_k<S, KIND, ABL, WPE>, S::T, S::LDS));
	printf("%-40s kind=%d abl=%d wpe=%d occ=%d |", name, KIND, ABL, WPE, occ);
	for (int wpc : {0}) {
		{ int sg = wpc; CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_stagger), &sg, sizeof sg)); }
		int grid = NWORK;
		double us = time_us([&] { hipLaunchKernelGGL((col_k<S, KIND, ABL, WPE>), dim3(grid), dim3(S::T), S::LDS, 0, a); }) / g_frames;
		printf(" stg%d:%6.1fus", wpc, us);
	}
	printf("\n"); fflush(stdout);
}

#define ROW(T, WPE, ...) run_row<RowSpec<3840, 3, T, __VA_ARGS__>, KIND_REDFT10, 0, WPE>("ROW T=" #T " R=" #__VA_ARGS__)
#define ROWA(T, WPE, ...) run_row<RowSpec<3840, 3, T, __VA_ARGS__>, KIND_REDFT10, 1, WPE>("ROW T=" #T " R=" #__VA_ARGS__)
#define ROW3(T, WPE, ...) run_row<RowSpec<3840, 3, T, __VA_ARGS__>, KIND_REDFT01, 0, WPE>("ROW T=" #T " R=" #__VA_ARGS__)
#define COL(K, T, WPE, ...) run_col<ColSpec<2160, K, T, __VA_ARGS__>, KIND_REDFT10, 0, WPE>("COL K=" #K " T=" #T " R=" #__VA_ARGS__)
#define COLA(K, T, WPE, ...) run_col<ColSpec<2160, K, T, __VA_ARGS__>, KIND_REDFT10, 1, WPE>("COL K=" #K " T=" #T " R=" #__VA_ARGS__)
#define COL3A(K, T, WPE, ...) run_col<ColSpec<2160, K, T, __VA_ARGS__>, KIND_REDFT01, 1, WPE>("COL K=" #K " T=" #T " R=" #__VA_ARGS__)
#define ROW3A(T, WPE, ...) run_row<RowSpec<3840, 3, T, __VA_ARGS__>, KIND_REDFT01, 1, WPE>("ROW T=" #T " R=" #__VA_ARGS__)
#define COL3(K, T, WPE, ...) run_col<ColSpec<2160, K, T, __VA_ARGS__>, KIND_REDFT01, 0, WPE>("COL K=" #K " T=" #T " R=" #__VA_ARGS__)

int main()
{
	g_frames = getenv("FRAMES") ? atoi(getenv("FRAMES")) : 1;
	CHK(hipMalloc(&g_frame, (size_t)H * W * C * 4 * g_frames));
	{
		std::vector<float> h((size_t)H * W * C * g_frames);
		for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f;
		CHK(hipMemcpy(g_frame, h.data(), h.size() * 4, hipMemcpyHostToDevice));
	}
	printf("frames per launch: %d (times are per frame; wpc0 = one workgroup per item, wpcN = 256*N persistent workgroups)\n", g_frames);
#include "kbench_variants.inc"
	return 0;
}
