// f64bench.hip -- radix / thread-count variants of the double-precision 4K passes (not part of the product): each pass alone on one
// 3840x2160x3 double frame, 30 launches, HIP events.  Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -Idspfun_amd/csrc
// tools/f64bench.hip -o tools/f64bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "dct_spec.h"
using namespace dspfft;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef cx<double> cd;

template <class S, int KIND, bool ROWK>
__global__ void __launch_bounds__(S::T, S::WPE) pass_k(const PassArgsD a)
{
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	if constexpr (ROWK) row_base(a, blockIdx.x, bin, bout); else S::base(a, blockIdx.x, bin, bout);
	auto *buf = reinterpret_cast<std::conditional_t<ROWK, cd, SigVec<double, 1>> *>(lds);
	S::template prefetch<KIND>(a, bin, tid, st);
	S::template phase<KIND, 0>(a, buf, bout, tid, st);
	__syncthreads();
	static_for<1, S::NPH>([&](auto ph) {
		S::template phase<KIND, ph>(a, buf, bout, tid, st);
		if constexpr (ph + 1 < S::NPH) __syncthreads();
	});
}

static const int H = 2160, W = 3840, C = 3;
static const size_t NF = (size_t)H * W * C;
static double *g_buf;
struct Tables { cd *T, *Wt; };
static Tables make_tables(int N, int L)
{
	std::vector<cd> T(N + 1), Wv(L);
	for (int j = 0; j <= N; j++) T[j] = cmk<double>(cos(M_PI * j / (2.0 * N)), -sin(M_PI * j / (2.0 * N)));
	for (int t = 0; t < L; t++) Wv[t] = cmk<double>(cos(2 * M_PI * t / L), -sin(2 * M_PI * t / L));
	Tables r;
	CHK(hipMalloc(&r.T, T.size() * 16)); CHK(hipMalloc(&r.Wt, Wv.size() * 16));
	CHK(hipMemcpy(r.T, T.data(), T.size() * 16, hipMemcpyHostToDevice)); CHK(hipMemcpy(r.Wt, Wv.data(), Wv.size() * 16, hipMemcpyHostToDevice));
	return r;
}
static Tables g_trow, g_tcol;

template <class S, int KIND, bool ROWK>
static void bench(const char *name)
{
	PassArgsD a; memset((void *)&a, 0, sizeof a);
	a.in = a.out = g_buf;
	a.in_scale0 = a.out_scale0 = 1.0;
	a.kind = KIND;
	a.scale = KIND == KIND_REDFT10 ? 1.0 : (ROWK ? 1.0 / (2.0 * W) : 1.0 / (2.0 * H));
	int grid;
	if constexpr (ROWK) { a.N = W; a.C = C; a.nb0 = H; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)W * C; a.T = g_trow.T; a.W = g_trow.Wt; grid = H; }
	else { a.N = H; a.K = S::K; a.B = S::B; a.ninner = W * C; a.ntiles = W * C / S::K; a.es_in = a.es_out = (long long)W * C; a.nb0 = 1; a.nb1 = 1; a.T = g_tcol.T; a.W = g_tcol.Wt; grid = a.ntiles; }
	if (S::LDS > 160 * 1024) { printf("%-46s kind %d: LDS %zu too large\n", name, KIND, (size_t)S::LDS); return; }
	CHK(hipFuncSetAttribute((const void *)pass_k<S, KIND, ROWK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
	for (int i = 0; i < 3; i++) hipLaunchKernelGGL((pass_k<S, KIND, ROWK>), dim3(grid), dim3(S::T), S::LDS, 0, a);
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	CHK(hipEventRecord(e0, 0));
	for (int i = 0; i < 30; i++) hipLaunchKernelGGL((pass_k<S, KIND, ROWK>), dim3(grid), dim3(S::T), S::LDS, 0, a);
	CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
	float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
	CHK(hipGetLastError());
	printf("%-46s kind %d: %6.1f us  (LDS %zu, %d threads)\n", name, KIND, ms * 1000 / 30, (size_t)S::LDS, S::T);
}
#define ROWV(T_, ...) { typedef RowSpecT<double, 3840, 3, T_, __VA_ARGS__> S; bench<S, 0, true>("row " #T_ " " #__VA_ARGS__); bench<S, 1, true>("row " #T_ " " #__VA_ARGS__); }
#define COLV(K_, T_, ...) { typedef ColSpecT<double, 2160, K_, T_, __VA_ARGS__> S; bench<S, 0, false>("col K" #K_ " " #T_ " " #__VA_ARGS__); bench<S, 1, false>("col K" #K_ " " #T_ " " #__VA_ARGS__); }

int main()
{
	CHK(hipMalloc(&g_buf, NF * 8));
	{ std::vector<double> h(NF); for (size_t i = 0; i < NF; i++) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0; CHK(hipMemcpy(g_buf, h.data(), NF * 8, hipMemcpyHostToDevice)); }
	g_trow = make_tables(W, W / 2); g_tcol = make_tables(H, H);
	for (int round = 0; round < 2; round++) {
		ROWV(1024, 12, 10, 16)
		ROWV(1024, 8, 8, 6, 5)
		ROWV(1024, 6, 8, 8, 5)
		ROWV(1024, 4, 4, 8, 15)
		ROWV(1024, 8, 16, 15)
		ROWV(1024, 4, 4, 4, 6, 5)
		ROWV(512, 12, 10, 16)
		COLV(4, 512, 12, 12, 15)
		COLV(4, 512, 8, 6, 9, 5)
		COLV(4, 512, 6, 8, 9, 5)
		COLV(4, 768, 6, 6, 6, 10)
		COLV(4, 512, 6, 6, 6, 10)
		COLV(4, 512, 10, 12, 18)
		COLV(4, 1024, 6, 6, 6, 10)
		COLV(2, 256, 12, 12, 15)
		printf("--\n");
	}
	return 0;
}
