// kstamp.hip -- one specialised kernel at a time on its BASELINE-sized workload, alone: time per launch and the clock behind every barrier
// (DSP_STAMP in spec_kernels.h / spec_fused.h: per-phase durations averaged over the workgroups).  For tuning without rebuilding the library.
// CAUTION: the stamps themselves cost time (a global pointer load and a store per phase: the stamped 8K row-pair kernel runs 246 us against 205 us
// unstamped) -- read the phase SHARES, never compare a stamped kernel's total with an unstamped one's.
//
//   hipcc -std=c++17 -O3 -fno-slp-vectorize -ffp-contract=on --offload-arch=gfx950 -Idspfun_amd/csrc -Iinclude tools/kstamp.hip -o tools/kstamp
//   tools/kstamp [pair|half|rt|zoomx|u8]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

__device__ unsigned long long *g_stamps;
#define DSP_STAMP(i) do { if (g_stamps && threadIdx.x == 0) g_stamps[(size_t)blockIdx.x * 32 + (i)] = clock64(); } while (0)
#include "spec_kernels.h"

using namespace dspfft;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static cf *upload(const std::vector<cf> &v) { cf *d; CK(hipMalloc(&d, v.size() * sizeof(cf))); CK(hipMemcpy(d, v.data(), v.size() * sizeof(cf), hipMemcpyHostToDevice)); return d; }
static cf *tab_T(int N) { std::vector<cf> t(N + 1); for (int j = 0; j <= N; j++) t[j] = cmk<float>((float)cos(M_PI * j / (2.0 * N)), (float)-sin(M_PI * j / (2.0 * N))); return upload(t); }
static cf *tab_W(int L) { std::vector<cf> t(L); for (int j = 0; j < L; j++) t[j] = cmk<float>((float)cos(2 * M_PI * j / L), (float)-sin(2 * M_PI * j / L)); return upload(t); }
static cf *tab_H(int N) { std::vector<cf> t(N / 2); for (int j = 0; j < N / 2; j++) t[j] = cmk<float>((float)cos(2 * M_PI * j / N), (float)-sin(2 * M_PI * j / N)); return upload(t); }

static unsigned long long *d_stamps;
static void stamps_on(bool on) { unsigned long long *p = on ? d_stamps : nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof p)); }
static void report(const char *what, int nwg, float us)
{
	std::vector<unsigned long long> h((size_t)nwg * 32);
	CK(hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost));
	double sum[32] = {0};
	for (int wg = 0; wg < nwg; wg++) {
		unsigned long long prev = h[(size_t)wg * 32];
		for (int i = 1; i < 32; i++) { const unsigned long long v = h[(size_t)wg * 32 + i]; if (!v) continue; sum[i] += (double)(v - prev); prev = v; }
	}
	double tot = 0;
	printf("%-44s %8.1f us   clocks per phase:", what, us);
	for (int i = 1; i < 32; i++) if (sum[i] > 0) { printf(" [%d] %.0f", i, sum[i] / nwg); tot += sum[i] / nwg; }
	printf("   total %.0f\n", tot);
}
template <class F> static float timed(F &&launch, int reps = 30)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int i = 0; i < 5; i++) launch();
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0, 0));
	for (int i = 0; i < reps; i++) launch();
	CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	return ms * 1e3f / reps;
}
template <class F> static void run(const char *what, int nwg, F &&launch)
{
	stamps_on(false);
	const float us = timed(launch);
	CK(hipMemset(d_stamps, 0, (size_t)nwg * 32 * 8));
	stamps_on(true);
	launch();
	CK(hipDeviceSynchronize());
	report(what, nwg, us);
}

template <class S> static void rt_case(float *x, const char *what)
{
	const int w = 1920, h = 1080, frames = 256;               // config 5's luma clip (2.1 GB: nothing of it stays in the caches)
	static float *clip = nullptr;
	if (!clip) { CK(hipMalloc(&clip, (size_t)w * h * frames * 4)); for (int i = 0; i < frames; i += 32) CK(hipMemcpy(clip + (size_t)i * w * h, x, (size_t)w * h * 32 * 4, hipMemcpyDeviceToDevice)); }
	x = clip;
	PassArgs af = {};
	af.N = S::N; af.K = S::K; af.B = S::K / 2; af.ninner = w; af.ntiles = w / S::K; af.es_in = af.es_out = w; af.nb0 = frames; af.nb1 = 1;
	af.sb0_in = af.sb0_out = (long long)w * h;
	af.in = x; af.out = x; af.T = tab_T(S::N); af.W = tab_W(S::N); af.scale = 1.f; af.in_scale0 = af.out_scale0 = 1.f; af.kind = 0;
	PassArgs ai = af; ai.kind = 1; ai.scale = 1.f / (2.f * h);
	MotionFilter mf = {};
	mf.ad = 1; mf.ah = h; mf.aw = w; mf.mh = h; mf.mw = w; mf.b1d = 1; mf.b1h = h; mf.b1w = w; mf.damp = mf.boost = 1.f; mf.quantizer = 40.f; mf.enabled = 1;
	motion_filter_set_divs(mf, 1);
	FilterOp f; f.p = mf;
	const int nwg = af.ntiles * frames;
	char name[128];
	CK(hipFuncSetAttribute(reinterpret_cast<const void *>(col_roundtrip_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
	snprintf(name, sizeof name, "%s, quantiser (256 frames)", what);
	run(name, nwg, [&]() { hipLaunchKernelGGL((col_roundtrip_kernel<S>), dim3(nwg), dim3(S::T), S::LDS, 0, af, ai, f, (unsigned long long *)nullptr); });
	f.p.enabled = 0;
	snprintf(name, sizeof name, "%s, no filter", what);
	run(name, nwg, [&]() { hipLaunchKernelGGL((col_roundtrip_kernel<S>), dim3(nwg), dim3(S::T), S::LDS, 0, af, ai, f, (unsigned long long *)nullptr); });
}

int main(int argc, char **argv)
{
	setvbuf(stdout, NULL, _IONBF, 0);
	const char *which = argc > 1 ? argv[1] : "pair half rt zoomx u8";
	const int W8 = 7680, H8 = 4320;
	const size_t n8 = (size_t)W8 * H8 * 3;
	float *x;
	CK(hipMalloc(&x, n8 * 4));
	{
		std::vector<float> hx(n8);
		unsigned s = 12345;
		for (size_t i = 0; i < n8; i++) { s = s * 1664525u + 1013904223u; hx[i] = (s >> 8) * (1.0f / 16777216.0f); }
		CK(hipMemcpy(x, hx.data(), n8 * 4, hipMemcpyHostToDevice));
	}
	CK(hipMalloc(&d_stamps, (size_t)300000 * 32 * 8));
	if (strstr(which, "pair")) {
		typedef RowSpec<7680, 3, 1024, 16, 15, 16> S;
		PassArgs a = {};
		a.N = S::N; a.C = 3; a.nb0 = H8; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)W8 * 3; a.nlines = H8;
		a.in = x; a.out = x; a.T = tab_T(S::N); a.W = tab_W(S::L); a.scale = 1.f; a.in_scale0 = a.out_scale0 = 1.f;
		CK(hipFuncSetAttribute(reinterpret_cast<const void *>(row_pair_kernel<S, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
		CK(hipFuncSetAttribute(reinterpret_cast<const void *>(row_pair_kernel<S, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
		a.kind = 0; run("row_pair 7680x3 REDFT10 (2160 pairs)", H8 / 2, [&]() { hipLaunchKernelGGL((row_pair_kernel<S, 0, true>), dim3(H8 / 2), dim3(S::T), S::LDS, 0, a); });
		a.kind = 1; run("row_pair 7680x3 REDFT01", H8 / 2, [&]() { hipLaunchKernelGGL((row_pair_kernel<S, 1, true>), dim3(H8 / 2), dim3(S::T), S::LDS, 0, a); });
	}
	if (strstr(which, "pipe")) {
		// round 6's pipelined pair kernel (one workgroup per CU; the stamps are the LAST pair's of each workgroup): [1] phase 0 of r1 (its wait for the line
		// included), [2] r2 requested + barrier (the slowest wave's wait for r1), [3..6] stages and last stage, [9] closing phase to registers; [10..16] the
		// same for r2, [19] closing phase + the pair's stores (+ the next pair's r1 requested)
		typedef RowSpec<7680, 3, 768, 16, 15, 16> S;
		PassArgs a = {};
		a.N = S::N; a.C = 3; a.nb0 = H8; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)W8 * 3; a.nlines = H8;
		a.in = x; a.out = x; a.T = tab_T(S::N); a.W = tab_W(S::L); a.scale = 1.f; a.in_scale0 = a.out_scale0 = 1.f;
		CK(hipFuncSetAttribute(reinterpret_cast<const void *>(row_pair_pipe_kernel<S, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_pipe_lds<S>()));
		CK(hipFuncSetAttribute(reinterpret_cast<const void *>(row_pair_pipe_kernel<S, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_pipe_lds<S>()));
		a.kind = 0; run("row_pair_pipe 7680x3 REDFT10 (768 threads, 256 workgroups)", 256, [&]() { hipLaunchKernelGGL((row_pair_pipe_kernel<S, 0, true>), dim3(256), dim3(S::T), pair_pipe_lds<S>(), 0, a, H8 / 2); });
		a.kind = 1; run("row_pair_pipe 7680x3 REDFT01 (next pair requested in front of the stores)", 256, [&]() { hipLaunchKernelGGL((row_pair_pipe_kernel<S, 1, true>), dim3(256), dim3(S::T), pair_pipe_lds<S>(), 0, a, H8 / 2); });
	}
	if (strstr(which, "half")) {
		typedef ColHalfSpec<4320, 16, 1024, 12, 12, 15> S;
		PassArgs a = {};
		a.N = S::N; a.K = 16; a.B = 8; a.ninner = W8 * 3; a.ntiles = W8 * 3 / 16; a.es_in = a.es_out = (long long)W8 * 3; a.nb0 = a.nb1 = 1;
		a.in = x; a.out = x; a.T = tab_T(S::N); a.W = tab_W(S::M); a.H = tab_H(S::N); a.scale = 1.f; a.in_scale0 = a.out_scale0 = 1.f;
		const int nwg = 2 * a.ntiles;
		CK(hipFuncSetAttribute(reinterpret_cast<const void *>(col_half_kernel<S, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
		CK(hipFuncSetAttribute(reinterpret_cast<const void *>(col_half_kernel<S, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
		a.kind = 0; run("col_half 4320 K=16 REDFT10 (2880 half tiles)", nwg, [&]() { hipLaunchKernelGGL((col_half_kernel<S, 0, true>), dim3(nwg), dim3(S::T), S::LDS, 0, a); });
		a.kind = 1; run("col_half 4320 K=16 REDFT01", nwg, [&]() { hipLaunchKernelGGL((col_half_kernel<S, 1, true>), dim3(nwg), dim3(S::T), S::LDS, 0, a); });
	}
	if (strstr(which, "rt")) {
		rt_case<ColSpec<1080, 16, 512, 12, 10, 9>>(x, "col_roundtrip 1080 K=16 T=512");
		rt_case<ColSpec<1080, 8, 256, 12, 10, 9>>(x, "col_roundtrip 1080 K=8 T=256");
		rt_case<ColSpec<1080, 8, 512, 12, 10, 9>>(x, "col_roundtrip 1080 K=8 T=512");
	}
	if (strstr(which, "u8")) {
		// motion config 5's 8-bit row ends on the luma clip: 256 x 1080 lines of 1920 samples
		typedef RowSpec<1920, 1, 128, 8, 8, 15> S;
		const int w = 1920, lines = 1080 * 256;
		uint8_t *p8; float *pf32;
		CK(hipMalloc(&p8, (size_t)w * lines)); CK(hipMalloc(&pf32, (size_t)w * lines * 4));
		CK(hipMemset(p8, 0x37, (size_t)w * lines));
		for (int i = 0; i < 256; i += 32) CK(hipMemcpy(pf32 + (size_t)i * w * 1080, x, (size_t)w * 1080 * 32 * 4, hipMemcpyDeviceToDevice));
		PassArgs a = {};
		a.N = S::N; a.C = 1; a.nb0 = lines; a.nb1 = 1; a.sb0_in = a.sb0_out = w; a.nlines = lines;
		a.in = pf32; a.out = pf32; a.T = tab_T(S::N); a.W = tab_W(S::L); a.scale = 1.f; a.in_scale0 = a.out_scale0 = 1.f;
		U8IO io = {};
		a.kind = 0; io.in = p8; io.out = nullptr; io.mul = 1.0;
		run("row_spec_u8 1920 REDFT10 (u8 -> f32, 276480 lines)", lines, [&]() { hipLaunchKernelGGL((row_spec_u8_kernel<S, 0>), dim3(lines), dim3(S::T), S::LDS, 0, a, io); });
		a.kind = 1; io.in = nullptr; io.out = p8; io.mul = 1.0 / 3840.0;
		run("row_spec_u8 1920 REDFT01 (f32 -> u8)", lines, [&]() { hipLaunchKernelGGL((row_spec_u8_kernel<S, 1>), dim3(lines), dim3(S::T), S::LDS, 0, a, io); });
	}
	if (strstr(which, "zoomx")) {
		// zoom's x stage of BASELINE config 3: 2160 lines of 1920 pixels (3 floats) -> 7680 pixels; the table's values do not matter for time
		typedef RowDuoT<7680, 256, 16, 16, 15> S;
		const int cw = 1920, vw = 7680, lines = 2160;
		std::vector<cf> tb((size_t)4 * S::L);
		for (size_t i = 0; i < tb.size(); i++) tb[i] = cmk<float>(0.5f + (float)(i % 7) * 0.01f, -0.25f);
		ZoomXArgs z = {};
		float *out; CK(hipMalloc(&out, (size_t)vw * lines * 3 * 4));
		z.in = x; z.out = out; z.tab = (const float *)upload(tb); z.W = tab_W(S::L); z.in_pitch = (long long)cw * 3; z.out_pitch = (long long)vw * 3; z.cw = cw; z.vw = vw; z.lines = lines;
		CK(hipFuncSetAttribute(reinterpret_cast<const void *>(zoomx_lean_kernel<S, 3, 1, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
		run("zoomx_lean 1920 -> 7680 x 3 (2160 lines)", lines, [&]() { hipLaunchKernelGGL((zoomx_lean_kernel<S, 3, 1, 2, false>), dim3(lines), dim3(S::T), S::LDS, 0, z); });
	}
	return 0;
}
