// kbench.hip -- experiment harness (not part of the product): times variants of the specialised
// passes (thread count, radices, tile width) and ablations on one 3840x2160x3 frame.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idspfun_amd/csrc tools/kbench.hip -o tools/kbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include "dct_spec.h"
using namespace dspfft;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ABL: 0 full, 1 skip the FFT stages (load/pre + post only), 2 load + store phases only without LDS barriers in between
template <class S, int KIND, int ABL>
__global__ void __launch_bounds__(S::T) row_k(const PassArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	cf *planes = reinterpret_cast<cf *>(lds);
	long long bin, bout;
	row_base(a, blockIdx.x, bin, bout);
	const int tid = threadIdx.x;
	typename S::State st;
	static_for<0, S::NPH>([&](auto ph) {
		if constexpr (ABL == 0 || ph == 0 || ph == S::NPH - 1) {
			S::template phase<KIND, ph>(a, planes, bin, bout, tid, st);
			if constexpr (ph + 1 < S::NPH) __syncthreads();
		}
	});
}
template <class S, int KIND, int ABL>
__global__ void __launch_bounds__(S::T) col_k(const PassArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	float4 *buf = reinterpret_cast<float4 *>(lds);
	long long bin, bout;
	S::base(a, blockIdx.x, bin, bout);
	const int tid = threadIdx.x;
	typename S::State st;
	static_for<0, S::NPH>([&](auto ph) {
		if constexpr (ABL == 0 || ph == 0 || ph == S::NPH - 1) {
			S::template phase<KIND, ph>(a, buf, bin, bout, tid, st);
			if constexpr (ph + 1 < S::NPH) __syncthreads();
		}
	});
}

// diagnostic build: s_memtime after every phase (wave 0 of each workgroup), shares only -- never quote its run time
template <class S, int KIND>
__global__ void __launch_bounds__(S::T) row_stamp(const PassArgs a, unsigned long long *stamps)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	cf *planes = reinterpret_cast<cf *>(lds);
	long long bin, bout;
	row_base(a, blockIdx.x, bin, bout);
	const int tid = threadIdx.x;
	typename S::State st;
	unsigned long long t[S::NPH + 1];
	t[0] = __builtin_amdgcn_s_memtime();
	static_for<0, S::NPH>([&](auto ph) {
		S::template phase<KIND, ph>(a, planes, bin, bout, tid, st);
		if constexpr (ph + 1 < S::NPH) __syncthreads();
		else __builtin_amdgcn_s_waitcnt(0);
		t[ph + 1] = __builtin_amdgcn_s_memtime();
	});
	if (tid == 0) for (int i = 0; i <= S::NPH; i++) stamps[(size_t)blockIdx.x * 16 + i] = t[i];
}
template <class S, int KIND>
__global__ void __launch_bounds__(S::T) col_stamp(const PassArgs a, unsigned long long *stamps)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	float4 *buf = reinterpret_cast<float4 *>(lds);
	long long bin, bout;
	S::base(a, blockIdx.x, bin, bout);
	const int tid = threadIdx.x;
	typename S::State st;
	unsigned long long t[S::NPH + 1];
	t[0] = __builtin_amdgcn_s_memtime();
	static_for<0, S::NPH>([&](auto ph) {
		S::template phase<KIND, ph>(a, buf, bin, bout, tid, st);
		if constexpr (ph + 1 < S::NPH) __syncthreads();
		else __builtin_amdgcn_s_waitcnt(0);
		t[ph + 1] = __builtin_amdgcn_s_memtime();
	});
	if (tid == 0) for (int i = 0; i <= S::NPH; i++) stamps[(size_t)blockIdx.x * 16 + i] = t[i];
}

static void report_stamps(unsigned long long *d, int nwg, int nph)
{
	std::vector<unsigned long long> h((size_t)nwg * 16);
	CHK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
	std::vector<double> avg(nph, 0.0);
	unsigned long long t0 = ~0ull, t1 = 0;
	double life = 0;
	for (int w = 0; w < nwg; w++) {
		for (int i = 0; i < nph; i++) avg[i] += (double)(h[(size_t)w * 16 + i + 1] - h[(size_t)w * 16 + i]);
		life += (double)(h[(size_t)w * 16 + nph] - h[(size_t)w * 16]);
		if (h[(size_t)w * 16] < t0) t0 = h[(size_t)w * 16];
		if (h[(size_t)w * 16 + nph] > t1) t1 = h[(size_t)w * 16 + nph];
	}
	printf("   stamps (s_memtime ticks @100MHz?): kernel span %.0f, mean WG life %.0f; phases:", (double)(t1 - t0), life / nwg);
	for (int i = 0; i < nph; i++) printf(" %.0f", avg[i] / nwg);
	printf("\n");
}

static const int H = 2160, W = 3840, C = 3;
static float *g_frame;

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

template <class S, int KIND, int ABL>
static void run_row(const char *name)
{
	static Tables tb = make_tables(S::N, S::L);
	PassArgs a; memset((void *)&a, 0, sizeof a);
	a.in = g_frame; a.out = g_frame; a.N = S::N; a.kind = KIND; a.C = S::C; a.nb0 = H; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)W * C;
	a.T = tb.T; a.W = tb.Wt; a.scale = 1.f / 7680.f; a.in_scale0 = a.out_scale0 = 1.f;
	CHK(hipFuncSetAttribute((const void *)row_k<S, KIND, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
	double us = time_us([&] { hipLaunchKernelGGL((row_k<S, KIND, ABL>), dim3(H), dim3(S::T), S::LDS, 0, a); });
	printf("%-52s kind=%d abl=%d  %8.1f us  %7.1f GB/s(2x99.5MB)\n", name, KIND, ABL, us, 2.0 * H * W * C * 4 / us / 1e3); fflush(stdout);
	if (ABL == 0 && getenv("STAMPS")) {
		unsigned long long *d; CHK(hipMalloc(&d, (size_t)H * 16 * 8));
		CHK(hipFuncSetAttribute((const void *)row_stamp<S, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
		hipLaunchKernelGGL((row_stamp<S, KIND>), dim3(H), dim3(S::T), S::LDS, 0, a, d);
		CHK(hipDeviceSynchronize());
		report_stamps(d, H, S::NPH); CHK(hipFree(d));
	}
}
template <class S, int KIND, int ABL>
static void run_col(const char *name)
{
	static Tables tb = make_tables(S::N, S::N);
	PassArgs a; memset((void *)&a, 0, sizeof a);
	a.in = g_frame; a.out = g_frame; a.N = S::N; a.kind = KIND; a.K = S::K; a.B = S::B; a.ninner = W * C; a.ntiles = W * C / S::K;
	a.es_in = a.es_out = (long long)W * C; a.nb0 = a.nb1 = 1;
	a.T = tb.T; a.W = tb.Wt; a.scale = 1.f / 4320.f; a.in_scale0 = a.out_scale0 = 1.f;
	CHK(hipFuncSetAttribute((const void *)col_k<S, KIND, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
	double us = time_us([&] { hipLaunchKernelGGL((col_k<S, KIND, ABL>), dim3(a.ntiles), dim3(S::T), S::LDS, 0, a); });
	printf("%-52s kind=%d abl=%d  %8.1f us  %7.1f GB/s(2x99.5MB)\n", name, KIND, ABL, us, 2.0 * H * W * C * 4 / us / 1e3); fflush(stdout);
	if (ABL == 0 && getenv("STAMPS")) {
		unsigned long long *d; CHK(hipMalloc(&d, (size_t)a.ntiles * 16 * 8));
		CHK(hipFuncSetAttribute((const void *)col_stamp<S, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
		hipLaunchKernelGGL((col_stamp<S, KIND>), dim3(a.ntiles), dim3(S::T), S::LDS, 0, a, d);
		CHK(hipDeviceSynchronize());
		report_stamps(d, a.ntiles, S::NPH); CHK(hipFree(d));
	}
}

#define ROW(T, ...) run_row<RowSpec<3840, 3, T, __VA_ARGS__>, KIND_REDFT10, 0>("ROW 3840x3 T=" #T " R=" #__VA_ARGS__)
#define ROWA(T, ABL, ...) run_row<RowSpec<3840, 3, T, __VA_ARGS__>, KIND_REDFT10, ABL>("ROW 3840x3 T=" #T " R=" #__VA_ARGS__)
#define ROW3(T, ...) run_row<RowSpec<3840, 3, T, __VA_ARGS__>, KIND_REDFT01, 0>("ROW 3840x3 T=" #T " R=" #__VA_ARGS__)
#define COL(K, T, ...) run_col<ColSpec<2160, K, T, __VA_ARGS__>, KIND_REDFT10, 0>("COL 2160 K=" #K " T=" #T " R=" #__VA_ARGS__)
#define COLA(K, T, ABL, ...) run_col<ColSpec<2160, K, T, __VA_ARGS__>, KIND_REDFT10, ABL>("COL 2160 K=" #K " T=" #T " R=" #__VA_ARGS__)
#define COL3(K, T, ...) run_col<ColSpec<2160, K, T, __VA_ARGS__>, KIND_REDFT01, 0>("COL 2160 K=" #K " T=" #T " R=" #__VA_ARGS__)

int main()
{
	CHK(hipMalloc(&g_frame, (size_t)H * W * C * 4));
	{
		std::vector<float> h((size_t)H * W * C);
		for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f;
		CHK(hipMemcpy(g_frame, h.data(), h.size() * 4, hipMemcpyHostToDevice));
	}
#include "kbench_variants.inc"
	return 0;
}
