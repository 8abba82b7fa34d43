// zoomxbench.hip -- zoom's x stage (BASELINE config 3: 4320 lines of 1920 -> 7680 RGB pixels) on the duo row kernel (dct_duo.h,
// spec_kernels.h zoomx_rows_kernel): thread counts / radix sets / waves per SIMD side by side, each checked against the cosine series
// evaluated in long double on a few lines.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -ffp-contract=on -std=c++17 -Idspfun_amd/csrc -Iinclude tools/zoomxbench.hip -o tools/zoomxbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "spec_kernels.h"
using namespace dspfft;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void tab_kernel(float *tab, int M, int cw, int nsrc, double theta, double scale)
{
	const int KT = M / 4 + 1;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nsrc * KT) return;
	const int s = i / KT, k = i - s * KT;
	const ZoomXEntry e = zoomx_table_entry(M, cw, k, s, theta, scale);
	for (int j = 0; j < 4; j++) { tab[((size_t)(2 * s) * KT + k) * 4 + j] = (float)e.lo[j]; tab[((size_t)(2 * s + 1) * KT + k) * 4 + j] = (float)e.hi[j]; }
}

// experiment copy of zoomx_rows_kernel: MODE bit 0 = persistent (a workgroup walks lines b, b + grid, ...), bit 1 = no global stores (compute only),
// bit 2 = no FFT phases (phase 0 + closing phase + memory only)
template <class S, int C, int NSRC, int WPE, int MODE>
__global__ void __launch_bounds__(S::T, WPE) zoomx_exp_kernel(const ZoomXArgs a_)
{
	typedef ZoomXRowsT<S, C, NSRC> Z;
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	typename S::V *buf = reinterpret_cast<typename S::V *>(lds);
	const int tid = threadIdx.x;
	ZoomXArgs a = a_;
	if (MODE & 2) a.vw = a_.lines < 0 ? a_.vw : 0;           // never true at run time: the stores stay in the code, their condition fails
	PassArgs w;
	w.W = a.W;
	for (int line = blockIdx.x; line < a.lines; line += gridDim.x) {
		const long long bin = (long long)line * a.in_pitch, bout = (long long)line * a.out_pitch;
		typename Z::State st;
		Z::load(a, bin, 0, tid, st);
#pragma nounroll
		for (int c = 0; c < C; c++) {
			int t = tid; asm volatile("" : "+v"(t));
			typename S::Regs r;
			Z::phase0(a, buf, t, st);
			__syncthreads();
			if (c + 1 < C) Z::load(a, bin, c + 1, t, st);
			if (!(MODE & 4))
			static_for<1, S::NS + 2>([&](auto ph) {
				S::template fft_phase<ph>(w, buf, t, r);
				__syncthreads();
			});
			Z::finish(a, buf, bout, c, t, st);
			__syncthreads();
		}
		if (!(MODE & 1)) break;
	}
}

__global__ void tabs_kernel(float *tab, int M, int cw, int nsrc, double theta, double scale)
{
	const int L = M / 2, KT = M / 4 + 1;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nsrc * KT) return;
	const int q = i / KT, k = i - q * KT;
	const ZoomXEntry e = zoomx_table_entry(M, cw, k, q, theta, scale);
	for (int j = 0; j < 4; j++) { tab[((size_t)q * L + k) * 4 + j] = (float)e.lo[j]; if (k > 0 && 2 * k != L) tab[((size_t)q * L + (L - k)) * 4 + j] = (float)e.hi[j]; }
}

// experiment copy of zoomx_lean_kernel with per-phase time stamps: stamps[wg * 16 + i] = s_memtime at phase boundary i of the LAST channel
// (wave 0 of each workgroup), i = 0 start of the line, 1 after phase A + barrier, 2 after B, 3 after C's butterfly, 4 after exchange + stores issued
template <class S, int C, int NSRC, int WPE>
__global__ void __launch_bounds__(S::T, WPE) zoomx_lean_stamp_kernel(const ZoomXArgs a, unsigned long long *stamps)
{
	typedef ZoomXLeanT<S, C, NSRC, false> Z;
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	typename S::V *buf = reinterpret_cast<typename S::V *>(lds);
	const int tid = threadIdx.x;
	const long long bin = (long long)blockIdx.x * a.in_pitch, bout = (long long)blockIdx.x * a.out_pitch;
	PassArgs w;
	w.W = a.W;
	typename Z::State st;
	unsigned long long t0 = __builtin_amdgcn_s_memtime(), ta = 0, tb = 0, tc = 0, td = 0, tA = 0;
#pragma nounroll
	for (int c = 0; c < C; c++) {
		int t = tid; asm volatile("" : "+v"(t));
		if (c == C - 1) tA = __builtin_amdgcn_s_memtime();
		Z::phase_a(a, w, buf, bin, c, t);
		__syncthreads();
		if (c == C - 1) ta = __builtin_amdgcn_s_memtime();
		static_for<1, S::NS - 1>([&](auto I) {
			Z::template phase_b<I>(w, buf, t);
			__syncthreads();
		});
		if (c == C - 1) tb = __builtin_amdgcn_s_memtime();
		typename Z::Ex e;
		Z::phase_c(buf, t, e);
		if (c == C - 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tc = __builtin_amdgcn_s_memtime(); }
		float recv[Z::RL];
		static_for<0, Z::RL>([&](auto i) { recv[i] = __shfl_xor(e.s[i], 63); });
		Z::phase_c_emit(a, bout, c, t, e, recv, st);
		if (c == C - 1) td = __builtin_amdgcn_s_memtime();
		if (c + 1 < C) __syncthreads();
	}
	if (tid == 0) { unsigned long long *p = stamps + (size_t)blockIdx.x * 8; p[0] = t0; p[1] = tA; p[2] = ta; p[3] = tb; p[4] = tc; p[5] = td; }
}

struct Ctx { float *in, *out[2], *tab, *tabs; cf *W; int cw, vw, lines, M; double theta, scale; std::vector<float> hin; };

static size_t g_extra_lds = 0;      // experiment: more LDS per workgroup = fewer workgroups per CU
template <class S, int NSRC, int WPE, int MODE = -1>
static void bench(const char *name, Ctx &c)
{
	const size_t LDSB = S::LDS + g_extra_lds;
	auto k = [&]() { if constexpr (MODE == -2) return zoomx_lean_kernel<S, 3, NSRC, WPE, false>; else if constexpr (MODE < 0) return zoomx_rows_kernel<S, 3, NSRC, WPE>; else return zoomx_exp_kernel<S, 3, NSRC, WPE, MODE>; }();
	hipFuncAttributes fa; CHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k)));
	CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSB));
	int occ = 0; CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, S::T, LDSB));
	const int grid = (MODE > 0 && (MODE & 1)) ? occ * 256 : c.lines;
	ZoomXArgs a; a.in = c.in; a.tab = MODE == -2 ? c.tabs : c.tab; a.W = c.W; a.in_pitch = (long long)c.cw * 3; a.out_pitch = (long long)c.vw * 3; a.cw = c.cw; a.vw = c.vw; a.lines = c.lines;
	CHK(hipMemset(c.out[0], 0xff, (size_t)c.lines * c.vw * 3 * 4));
	a.out = c.out[0];
	hipLaunchKernelGGL(k, dim3(grid), dim3(S::T), LDSB, 0, a);
	CHK(hipDeviceSynchronize());
	// check lines 0, 1, last
	double md = 0, mx = 0;
	for (int line : {0, 1, c.lines - 1}) {
		std::vector<float> o((size_t)c.vw * 3);
		CHK(hipMemcpy(o.data(), c.out[0] + (size_t)line * c.vw * 3, o.size() * 4, hipMemcpyDeviceToHost));
		const float *in = c.hin.data() + (size_t)line * c.cw * 3;
		for (int b = 0; b < c.vw; b += 7) for (int ch = 0; ch < 3; ch++) {
			long double s = 0;
			for (int u = 0; u < c.cw; u++) s += (long double)(u ? 1.0 : 0.5) * c.scale * in[u * 3 + ch] * cosl((long double)u * (M_PIl * (b + 0.5L) / c.M + c.theta));
			md = fmax(md, fabs((double)s - o[b * 3 + ch])); mx = fmax(mx, fabs((double)s));
		}
	}
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const int reps = 20;
	for (int i = 0; i < 4; i++) { a.out = c.out[i & 1]; hipLaunchKernelGGL(k, dim3(grid), dim3(S::T), LDSB, 0, a); }
	CHK(hipEventRecord(e0, 0));
	for (int i = 0; i < reps; i++) { a.out = c.out[i & 1]; hipLaunchKernelGGL(k, dim3(grid), dim3(S::T), LDSB, 0, a); }
	CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
	float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
	CHK(hipGetLastError());
	const double us = ms * 1000 / reps, bytes = (double)c.lines * (c.vw + c.cw) * 3 * 4;
	printf("%-34s T=%4d lds=%6zu vgpr=%3d scratch=%4zu wg/CU=%d : %7.1f us = %.2f TB/s | rel err %.2e\n", name, S::T, (size_t)S::LDS, fa.numRegs, (size_t)fa.localSizeBytes, occ, us, bytes / us * 1e-6, md / mx);
	fflush(stdout);
}

template <class S, int NSRC, int WPE>
static void stamps(const char *name, Ctx &c)
{
	auto k = zoomx_lean_stamp_kernel<S, 3, NSRC, WPE>;
	CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
	ZoomXArgs a; a.in = c.in; a.tab = c.tabs; a.W = c.W; a.in_pitch = (long long)c.cw * 3; a.out_pitch = (long long)c.vw * 3; a.cw = c.cw; a.vw = c.vw; a.lines = c.lines; a.out = c.out[0];
	unsigned long long *d; CHK(hipMalloc(&d, (size_t)c.lines * 64));
	for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k, dim3(c.lines), dim3(S::T), S::LDS, 0, a, d);
	CHK(hipDeviceSynchronize());
	std::vector<unsigned long long> h((size_t)c.lines * 8);
	CHK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
	double seg[5] = {0, 0, 0, 0, 0}; int n = 0;
	for (int wgi = 600; wgi < c.lines - 600; wgi++) {           // steady state: skip the first and the last wave of workgroups
		const unsigned long long *p = &h[(size_t)wgi * 8];
		seg[0] += (double)(p[1] - p[0]); seg[1] += (double)(p[2] - p[1]); seg[2] += (double)(p[3] - p[2]); seg[3] += (double)(p[4] - p[3]); seg[4] += (double)(p[5] - p[4]); n++;
	}
	printf("%s: per workgroup, shader cycles / 100: channels 0-1 %.2f | last channel: phase A %.2f, B %.2f, C butterfly %.2f, exchange + stores issued %.2f\n",
	       name, seg[0] / n / 100, seg[1] / n / 100, seg[2] / n / 100, seg[3] / n / 100, seg[4] / n / 100);
	CHK(hipFree(d));
}

int main(int argc, char **argv)
{
	Ctx c; c.M = 7680; c.cw = 1920; c.vw = 7680; c.lines = argc > 1 ? atoi(argv[1]) : 4320; c.theta = M_PI * (100.25 + 1.5) / 7680; c.scale = 1.0 / (1920.0 * 1080.0);
	const size_t nin = (size_t)c.lines * c.cw * 3, nout = (size_t)c.lines * c.vw * 3;
	c.hin.resize(nin);
	unsigned long long s = 12345;
	for (size_t i = 0; i < nin; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; c.hin[i] = (float)((double)(s >> 40) / 16777216.0 - 0.5) * 1000.f; }
	CHK(hipMalloc(&c.in, nin * 4)); CHK(hipMalloc(&c.out[0], nout * 4)); CHK(hipMalloc(&c.out[1], nout * 4));
	CHK(hipMemcpy(c.in, c.hin.data(), nin * 4, hipMemcpyHostToDevice));
	const int L = c.M / 2, KT = L / 2 + 1;
	CHK(hipMalloc(&c.tab, (size_t)4 * 2 * KT * 16));
	std::vector<cf> W(L);
	for (int t = 0; t < L; t++) W[t] = cmk<float>((float)cos(2 * M_PI * t / L), (float)-sin(2 * M_PI * t / L));
	CHK(hipMalloc(&c.W, L * sizeof(cf))); CHK(hipMemcpy(c.W, W.data(), L * sizeof(cf), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(tab_kernel, dim3((4 * KT + 255) / 256), dim3(256), 0, 0, c.tab, c.M, c.cw, 4, c.theta, c.scale);
	CHK(hipMalloc(&c.tabs, (size_t)4 * L * 16)); CHK(hipMemset(c.tabs, 0, (size_t)4 * L * 16));
	hipLaunchKernelGGL(tabs_kernel, dim3((4 * KT + 255) / 256), dim3(256), 0, 0, c.tabs, c.M, c.cw, 4, c.theta, c.scale);
	CHK(hipDeviceSynchronize());
	if (argc > 2) {           // profiling runs: one variant only
		if (!strcmp(argv[2], "lean1")) { g_extra_lds = 30000; bench<RowDuoT<7680, 256, 16, 16, 15>, 1, 2, -2>("lean 256thr 16.16.15 wpe2, ONE workgroup per CU", c); g_extra_lds = 0; bench<RowDuoT<7680, 256, 16, 16, 15>, 1, 2, -2>("lean 256thr 16.16.15 wpe2, two per CU", c); }
		else if (!strcmp(argv[2], "lean")) bench<RowDuoT<7680, 256, 16, 16, 15>, 1, 2, -2>("lean 256thr 16.16.15 wpe2", c);
		else if (!strcmp(argv[2], "rows")) bench<RowDuoT<7680, 256, 8, 8, 4, 15>, 1, 2>("256thr 8.8.4.15 wpe2", c);
		return 0;
	}
	bench<RowDuoT<7680, 256, 16, 16, 15>, 1, 2, -2>("lean 256thr 16.16.15 wpe2", c);
	stamps<RowDuoT<7680, 256, 16, 16, 15>, 1, 2>("lean 256thr 16.16.15", c);
	stamps<RowDuoT<7680, 256, 8, 8, 4, 15>, 1, 2>("lean 256thr 8.8.4.15", c);
	return 0;
}
