// wavebench.hip -- experiment harness (not part of the product): ONE WAVEFRONT PER ROW against the product's choice for planar rows.
// north_star asks for "rows staged in LDS with one-wavefront-per-row shuffle butterflies"; the product runs motion's planar 1920-sample
// rows (30 floats per lane for one wave) with 128 threads = two waves per row (spec_list.h).  Variants here, same RowSpec phases:
//   T = 64   one wavefront per row: no workgroup barrier is needed (the __syncthreads of a one-wave workgroup is a wave-local wait),
//            the exchange between butterfly stages goes through the wave's own LDS rows
//   T = 128  the product (two waves per row)        T = 256  four waves per row
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -ffp-contract=on -std=c++17 -Idspfun_amd/csrc tools/wavebench.hip -o tools/wavebench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "dct_spec.h"
using namespace dspfft;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <class S, int KIND>
__global__ void __launch_bounds__(S::T) row_k(const PassArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	cf *planes = reinterpret_cast<cf *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	row_base(a, blockIdx.x, bin, bout);
	S::template prefetch<KIND>(a, bin, tid, st);
	static_for<0, S::NPH>([&](auto ph) {
		S::template phase<KIND, ph>(a, planes, bout, tid, st);
		if constexpr (ph + 1 < S::NPH) __syncthreads();
	});
}

static const int N = 1920, LINES = 1080 * 64;
static float *g_buf;
static cf *g_T, *g_W;

template <class S, int KIND>
static void run(const char *name)
{
	PassArgs a; memset((void *)&a, 0, sizeof a);
	a.in = g_buf; a.out = g_buf; a.N = N; a.kind = KIND; a.C = S::C;
	a.T = g_T; a.W = g_W; a.scale = 1.f / 4000.f; a.in_scale0 = a.out_scale0 = 1.f;
	const int LINES = ::LINES / S::C;
	a.nb0 = LINES; a.nb1 = 1; a.sb0_in = a.sb0_out = N * S::C; a.sb1_in = a.sb1_out = 0;
	auto kern = row_k<S, KIND>;
	CHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
	int occ = 0; CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)kern, S::T, S::LDS));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	float best = 1e9;
	for (int rep = 0; rep < 6; rep++) {
		CHK(hipEventRecord(e0));
		hipLaunchKernelGGL(kern, dim3(LINES), dim3(S::T), S::LDS, 0, a);
		CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
	}
	CHK(hipGetLastError());
	const double bytes = 2.0 * LINES * N * S::C * 4;
	printf("%-44s kind=%d workgroups/CU=%2d waves/CU=%2d lds=%5zu | %7.1f us = %5.2f TB/s (read + write)\n", name, KIND, occ, occ * S::T / 64, S::LDS, best * 1e3, bytes / best / 1e9);
}

int main()
{
	CHK(hipMalloc(&g_buf, (size_t)LINES * N * 4));
	{
		std::vector<float> h((size_t)LINES * N);
		for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f;
		CHK(hipMemcpy(g_buf, h.data(), h.size() * 4, hipMemcpyHostToDevice));
		const int L = N / 2;
		std::vector<cf> T(N + 1), Wv(L);
		for (int j = 0; j <= N; j++) T[j] = cmk((float)cos(M_PI * j / (2.0 * N)), (float)-sin(M_PI * j / (2.0 * N)));
		for (int t = 0; t < L; t++) Wv[t] = cmk((float)cos(2 * M_PI * t / L), (float)-sin(2 * M_PI * t / L));
		CHK(hipMalloc(&g_T, T.size() * 8)); CHK(hipMalloc(&g_W, Wv.size() * 8));
		CHK(hipMemcpy(g_T, T.data(), T.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(g_W, Wv.data(), Wv.size() * 8, hipMemcpyHostToDevice));
	}
	printf("planar rows of %d samples, %d rows (%.0f MB), in place\n", N, LINES, (double)LINES * N * 4 / 1e6);
#define BOTH(T, ...) run<RowSpec<1920, 1, T, __VA_ARGS__>, KIND_REDFT10>("T=" #T " radices " #__VA_ARGS__); run<RowSpec<1920, 1, T, __VA_ARGS__>, KIND_REDFT01>("T=" #T " radices " #__VA_ARGS__);
	BOTH(64, 4, 16, 15)
	BOTH(64, 8, 8, 15)
	BOTH(64, 64, 15)
	BOTH(128, 4, 16, 15)
	BOTH(256, 4, 16, 15)
	// RGB rows of 1920 pixels (zoom C3, 1080p frames): product = T 256, radices 4, 15, 16 (720 butterflies in the first stage: three rounds)
#define BOTH3(T, ...) run<RowSpec<1920, 3, T, __VA_ARGS__>, KIND_REDFT10>("RGB T=" #T " radices " #__VA_ARGS__); run<RowSpec<1920, 3, T, __VA_ARGS__>, KIND_REDFT01>("RGB T=" #T " radices " #__VA_ARGS__);
	BOTH3(256, 4, 15, 16)
	BOTH3(384, 8, 8, 15)
	BOTH3(256, 8, 8, 15)
	BOTH3(384, 6, 10, 16)
	BOTH3(256, 12, 5, 16)
	BOTH3(256, 16, 4, 15)
	BOTH3(512, 8, 8, 15)
	return 0;
}
