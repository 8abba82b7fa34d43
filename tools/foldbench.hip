// foldbench.hip -- the folded row kernel (dct_fold.h, spec_kernels.h row_fold_kernel) on an 8K frame's lines, alone: total time per pass
// and the clock behind every barrier (per-phase durations, averaged over the workgroups), for tuning without rebuilding the library.
//
//   hipcc -std=c++17 -O3 -fno-slp-vectorize -ffp-contract=on --offload-arch=gfx950 -Idspfun_amd/csrc tools/foldbench.hip -o tools/foldbench
//   tools/foldbench [lines=4320] [T=512|768|...]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ unsigned long long *g_stamps;
#define DSP_FOLD_STAMP(i) do { if (g_stamps && threadIdx.x == 0) g_stamps[(size_t)blockIdx.x * 32 + (i)] = clock64(); } while (0)
#include "spec_kernels.h"

using namespace dspfft;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

#ifndef FB_SPEC
#define FB_SPEC RowFoldT<7680, 3, 512, 12, 10, 16>
#endif
typedef FB_SPEC S;

static std::vector<cf> tables()
{
	const long double pi = 3.14159265358979323846264338327950288L;
	const int H = S::H, M = S::M;
	std::vector<cf> t;
	for (int k = 0; k <= H; k++) t.push_back(cmk<float>((float)cosl(pi * k / (2.0L * H)), (float)-sinl(pi * k / (2.0L * H))));
	for (int k = 0; k < M; k++) t.push_back(cmk<float>((float)cosl(2 * pi * k / M), (float)-sinl(2 * pi * k / M)));
	for (int n = 0; n < M; n++) t.push_back(cmk<float>((float)cosl(pi * (4 * n + 1) / (4.0L * H)), (float)-sinl(pi * (4 * n + 1) / (4.0L * H))));
	for (int k = 0; k < M; k++) t.push_back(cmk<float>((float)cosl(pi * k / (long double)H), (float)-sinl(pi * k / (long double)H)));
	return t;
}

template <int KIND, bool PAIR>
static void run(const char *what, PassArgs a, int lines, unsigned *flags, unsigned long long *d_stamps, bool stamps)
{
	a.kind = KIND;
	auto kern = row_fold_kernel<S, KIND, PAIR, false>;
	CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
	unsigned long long *null_stamps = nullptr;
	CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &null_stamps, sizeof null_stamps));
	const int nwork = PAIR ? lines / 2 : lines;
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int i = 0; i < 5; i++) hipLaunchKernelGGL(kern, dim3(lines), dim3(S::T), S::LDS, 0, a, nwork, flags);
	CK(hipDeviceSynchronize());
	const int R = 30;
	CK(hipEventRecord(e0, 0));
	for (int i = 0; i < R; i++) hipLaunchKernelGGL(kern, dim3(lines), dim3(S::T), S::LDS, 0, a, nwork, flags);
	CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	printf("%-34s %8.1f us per pass  (%.2f TB/s of read + write once)\n", what, ms * 1e3 / R, 2.0 * lines * S::N * S::C * 4 / (ms * 1e-3 / R) / 1e12);
	if (!stamps) return;
	CK(hipMemset(d_stamps, 0, (size_t)lines * 32 * 8));
	CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_stamps, sizeof d_stamps));
	hipLaunchKernelGGL(kern, dim3(lines), dim3(S::T), S::LDS, 0, a, nwork, flags);
	CK(hipDeviceSynchronize());
	std::vector<unsigned long long> h((size_t)lines * 32);
	CK(hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost));
	double sum[32] = {0}; int last = 0;
	for (int wg = 0; wg < lines; wg++) {
		unsigned long long prev = h[(size_t)wg * 32];
		for (int i = 1; i < 32; i++) { const unsigned long long v = h[(size_t)wg * 32 + i]; if (!v) continue; sum[i] += (double)(v - prev); prev = v; if (i > last) last = i; }
	}
	double tot = 0;
	printf("   clocks between barriers (mean over %d workgroups):", lines);
	for (int i = 1; i <= last; i++) if (sum[i] > 0) { printf(" [%d] %.0f", i, sum[i] / lines); tot += sum[i] / lines; }
	printf("   total %.0f\n", tot);
}

int main(int argc, char **argv)
{
	const int lines = argc > 1 ? atoi(argv[1]) : 4320;
	const size_t n = (size_t)lines * S::N * S::C;
	float *x, *y;
	CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4));
	std::vector<float> hx(n);
	unsigned s = 12345;
	for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; hx[i] = (s >> 8) * (1.0f / 16777216.0f); }
	CK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
	std::vector<cf> t = tables();
	cf *d_tab; CK(hipMalloc(&d_tab, t.size() * sizeof(cf))); CK(hipMemcpy(d_tab, t.data(), t.size() * sizeof(cf), hipMemcpyHostToDevice));
	unsigned *flags; CK(hipMalloc(&flags, lines * 4)); CK(hipMemset(flags, 0, lines * 4));
	unsigned long long *d_stamps; CK(hipMalloc(&d_stamps, (size_t)lines * 32 * 8));
	PassArgs a = {};
	a.N = S::N; a.C = S::C; a.nb0 = lines; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)S::N * S::C; a.nlines = lines;
	a.in = x; a.out = x; a.H = d_tab; a.scale = 1.f; a.in_scale0 = a.out_scale0 = 1.f;
	printf("# %d lines of %d x %d floats, %d threads, LDS %zu, in place\n", lines, S::N, S::C, S::T, S::LDS);
	run<0, false>("REDFT10 single", a, lines, nullptr, d_stamps, true);
	run<1, false>("REDFT01 single", a, lines, nullptr, d_stamps, true);
	run<0, true>("REDFT10 pairs (in place)", a, lines, flags, d_stamps, false);
	run<1, true>("REDFT01 pairs (in place)", a, lines, flags, d_stamps, false);
	a.out = y;
	run<0, false>("REDFT10 single, out of place", a, lines, nullptr, d_stamps, false);
	run<0, true>("REDFT10 pairs, out of place", a, lines, nullptr, d_stamps, false);
	run<1, true>("REDFT01 pairs, out of place", a, lines, nullptr, d_stamps, false);
	return 0;
}
