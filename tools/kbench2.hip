// kbench2.hip -- round-2 diagnosis of the specialised 4K passes (not part of the product):
//   (1) per-phase timeline of a workgroup from s_memtime stamps (diagnostic build: its own kernel, never the product's)
//   (2) "compute only" ablation: every workgroup reads and writes the SAME line/tile (L2-resident I/O), so what is left
//       is LDS + VALU + barriers at the real occupancy
//   (3) the real pass, cache-resident (one frame) and HBM-resident (frames rotated)
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -Idspfun_amd/csrc tools/kbench2.hip -o tools/kbench2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "dct_spec.h"
using namespace dspfft;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define RSTAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// -DNOSYNC drops the barriers between the LDS phases (wrong results): an upper bound on what decoupling the waves of a workgroup could buy
#ifdef NOSYNC
#define STAGE_SYNC() ((void)0)
#else
#define STAGE_SYNC() __syncthreads()
#endif
constexpr int MAXST = 12;
struct WgStamps { unsigned long long t[MAXST]; unsigned long long r0, r1; unsigned xcc, cu; };

template <class S, int KIND, bool ROWK, bool STAMPS>
__global__ void __launch_bounds__(S::T, S::WPE) pass_k(const PassArgs a, WgStamps *dbg)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	unsigned long long ts[MAXST] = {}, r0 = 0, r1 = 0;
	if constexpr (STAMPS) { RSTAMP(r0); STAMP(ts[0]); }
	if constexpr (ROWK) row_base(a, blockIdx.x, bin, bout); else S::base(a, blockIdx.x, bin, bout);
	auto *buf = reinterpret_cast<std::conditional_t<ROWK, cf, SigVec<float, 2>> *>(lds);
	S::template prefetch<KIND>(a, bin, tid, st);
	if constexpr (STAMPS) STAMP(ts[1]);
	S::template phase<KIND, 0>(a, buf, bout, tid, st);
	__syncthreads();
	if constexpr (STAMPS) STAMP(ts[2]);
	static_for<1, S::NPH>([&](auto ph) {
		S::template phase<KIND, ph>(a, buf, bout, tid, st);
		if constexpr (ph + 1 < S::NPH) STAGE_SYNC();
		if constexpr (STAMPS) STAMP(ts[2 + ph]);
	});
	if constexpr (STAMPS) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		STAMP(ts[2 + S::NPH]);
		RSTAMP(r1);
		if (tid == 0) {
			WgStamps w;
			for (int i = 0; i < MAXST; i++) w.t[i] = ts[i];
			w.r0 = r0; w.r1 = r1;
			unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); w.xcc = x & 0xf;
			unsigned h; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h)); w.cu = h;
			dbg[blockIdx.x] = w;
		}
	}
}

static const int H = 2160, W = 3840, C = 3;
static const size_t NF = (size_t)H * W * C;
static float *g_buf;
static const int NFR = 8;
static WgStamps *g_dbg;

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
static double time_us(F f, int iters = 40)
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

template <class S, int KIND, bool ROWK>
static void run(const char *name)
{
	static Tables tb = make_tables(S::N, ROWK ? S::N / 2 : S::N);
	PassArgs a; memset((void *)&a, 0, sizeof a);
	a.in = g_buf; a.out = g_buf; a.N = S::N; a.kind = KIND;
	a.T = tb.T; a.W = tb.Wt; a.scale = 1.f / 4000.f; a.in_scale0 = a.out_scale0 = 1.f;
	int nwork;
	if constexpr (ROWK) {
		a.C = C; a.nb0 = H; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)W * C; a.sb1_in = a.sb1_out = (long long)NF; nwork = H;
	} else {
		a.K = S::K; a.B = S::B; a.ninner = W * C; a.ntiles = W * C / S::K; a.es_in = a.es_out = (long long)W * C;
		a.nb0 = 1; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)NF; nwork = a.ntiles;
	}
	auto kern = pass_k<S, KIND, ROWK, false>;
	auto kern_s = pass_k<S, KIND, ROWK, true>;
	CHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
	CHK(hipFuncSetAttribute((const void *)kern_s, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
	int occ = 0; CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)kern, S::T, S::LDS));
	// (3) real pass
	double us1 = time_us([&](int) { hipLaunchKernelGGL(kern, dim3(nwork), dim3(S::T), S::LDS, 0, a, g_dbg); });
	double usN = time_us([&](int i) { PassArgs b = a; b.in = b.out = g_buf + (size_t)(i % NFR) * NF; hipLaunchKernelGGL(kern, dim3(nwork), dim3(S::T), S::LDS, 0, b, g_dbg); });
	// (2) compute only: every workgroup on the same line / tile
	PassArgs c = a;
	if constexpr (ROWK) { c.sb0_in = c.sb0_out = 0; } else { c.ntiles = 1; c.nb0 = nwork; c.sb0_in = c.sb0_out = 0; }
	double usC = time_us([&](int) { hipLaunchKernelGGL(kern, dim3(nwork), dim3(S::T), S::LDS, 0, c, g_dbg); });
	printf("%-34s kind=%d occ=%d | 1frame %6.1f us | rotating %6.1f us | same-tile (compute) %6.1f us\n", name, KIND, occ, us1, usN, usC);
	// (1) stamps
	for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern_s, dim3(nwork), dim3(S::T), S::LDS, 0, a, g_dbg);
	CHK(hipDeviceSynchronize());
	std::vector<WgStamps> h(nwork);
	CHK(hipMemcpy(h.data(), g_dbg, sizeof(WgStamps) * nwork, hipMemcpyDeviceToHost));
	const int NST = 3 + S::NPH;
	std::vector<double> seg(NST, 0.0);
	double life = 0; unsigned long long rmin = ~0ull, rmax = 0;
	for (auto &w : h) {
		for (int i = 1; i < NST; i++) seg[i] += (double)(w.t[i] - w.t[i - 1]);
		life += (double)(w.t[NST - 1] - w.t[0]);
		rmin = std::min(rmin, w.r0); rmax = std::max(rmax, w.r1);
	}
	const double span_us = (double)(rmax - rmin) / 100.0;   // s_memrealtime = 100 MHz
	// clock estimate: sum of lifetimes in cycles vs realtime
	double cyc = 0, rt = 0;
	for (auto &w : h) { cyc += (double)(w.t[NST - 1] - w.t[0]); rt += (double)(w.r1 - w.r0) / 100.0; }
	const double mhz = cyc / rt;
	printf("    stamped launch span %.1f us, clock %.0f MHz, mean workgroup life %.2f us (%.0f cyc); segments in us:", span_us, mhz, life / nwork / mhz, life / nwork);
	const char *lbl_pre = "issue-loads";
	for (int i = 1; i < NST; i++) {
		const char *l = i == 1 ? lbl_pre : i == 2 ? "wait+ph0" : i == NST - 1 ? "drain-stores" : "";
		printf(" [%d%s%s %.2f]", i, *l ? ":" : "", l, seg[i] / nwork / mhz);
	}
	printf("\n");
	// concurrency profile: how many workgroups alive per 2-us bin of the launch
	{
		const int nb = (int)(span_us / 2.0) + 1;
		std::vector<double> alive(nb, 0.0);
		for (auto &w : h) {
			double s = (double)(w.r0 - rmin) / 100.0, e = (double)(w.r1 - rmin) / 100.0;
			for (int b = 0; b < nb; b++) { double lo = b * 2.0, hi = lo + 2.0; double ov = std::min(e, hi) - std::max(s, lo); if (ov > 0) alive[b] += ov / 2.0; }
		}
		printf("    workgroups alive per 2 us:");
		for (int b = 0; b < nb; b++) printf(" %.0f", alive[b]);
		printf("\n");
	}
	// first-round vs later workgroups
	{
		std::vector<std::pair<unsigned long long, double>> v;
		for (auto &w : h) v.push_back({w.r0, (double)(w.t[NST - 1] - w.t[0]) / mhz});
		std::sort(v.begin(), v.end());
		const int q = nwork / 4;
		double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
		for (int i = 0; i < q; i++) { s0 += v[i].second; s1 += v[q + i].second; s2 += v[2 * q + i].second; s3 += v[3 * q + i].second; }
		printf("    mean life by start-time quartile: %.2f %.2f %.2f %.2f us\n", s0 / q, s1 / q, s2 / q, s3 / q);
	}
	// XCD placement (dct_spec.h xcd_remap assumes workgroups b and b + 8 share an XCD): alone, and with the same kernel running on a
	// second stream at the same time
	{
		auto consistent = [&](const std::vector<WgStamps> &v) { int ok = 0; for (int b = 0; b < nwork; b++) ok += v[b].xcc == v[b & 7].xcc; return 100.0 * ok / nwork; };
		printf("    XCC of workgroups 0..7: %u %u %u %u %u %u %u %u; workgroups on the XCC of (id mod 8): alone %.1f %%", h[0].xcc, h[1].xcc, h[2].xcc, h[3].xcc, h[4].xcc, h[5].xcc, h[6].xcc, h[7].xcc, consistent(h));
		static hipStream_t s1 = nullptr, s2 = nullptr;
		if (!s1) { CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); }
		PassArgs b2 = a; b2.in = b2.out = g_buf + NF;
		CHK(hipDeviceSynchronize());
		for (int i = 0; i < 2; i++) {
			hipLaunchKernelGGL(kern_s, dim3(nwork), dim3(S::T), S::LDS, s1, a, g_dbg);
			hipLaunchKernelGGL(kern_s, dim3(nwork), dim3(S::T), S::LDS, s2, b2, g_dbg + 4096);
		}
		CHK(hipDeviceSynchronize());
		std::vector<WgStamps> ha(nwork), hb(nwork);
		CHK(hipMemcpy(ha.data(), g_dbg, sizeof(WgStamps) * nwork, hipMemcpyDeviceToHost));
		CHK(hipMemcpy(hb.data(), g_dbg + 4096, sizeof(WgStamps) * nwork, hipMemcpyDeviceToHost));
		unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0;
		for (int i = 0; i < nwork; i++) { a0 = std::min(a0, ha[i].r0); a1 = std::max(a1, ha[i].r1); b0 = std::min(b0, hb[i].r0); b1 = std::max(b1, hb[i].r1); }
		const double ov = ((double)std::min(a1, b1) - (double)std::max(a0, b0)) / 100.0;
		printf(", two streams (overlapping %.0f of %.0f us) %.1f %% and %.1f %%\n", ov > 0 ? ov : 0.0, (double)(a1 - a0) / 100.0, consistent(ha), consistent(hb));
	}
	fflush(stdout);
}

int main()
{
	CHK(hipMalloc(&g_buf, NF * 4 * NFR));
	CHK(hipMalloc(&g_dbg, sizeof(WgStamps) * 8192));
	{
		std::vector<float> h(NF);
		for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f;
		for (int f = 0; f < NFR; f++) CHK(hipMemcpy(g_buf + (size_t)f * NF, h.data(), NF * 4, hipMemcpyHostToDevice));
	}
	if (getenv("KB_ROWT")) {
		run<RowSpec<3840, 3, 576, 12, 10, 16>, KIND_REDFT10, true>("ROW 3840x3 T=576 12,10,16");
		run<RowSpec<3840, 3, 576, 12, 10, 16>, KIND_REDFT01, true>("ROW 3840x3 T=576 12,10,16");
		run<RowSpec<3840, 3, 640, 12, 10, 16>, KIND_REDFT10, true>("ROW 3840x3 T=640 12,10,16");
		run<RowSpec<3840, 3, 640, 12, 10, 16>, KIND_REDFT01, true>("ROW 3840x3 T=640 12,10,16");
		run<RowSpec<3840, 3, 576, 10, 12, 16>, KIND_REDFT10, true>("ROW 3840x3 T=576 10,12,16");
		run<RowSpec<3840, 3, 576, 10, 12, 16>, KIND_REDFT01, true>("ROW 3840x3 T=576 10,12,16");
		run<RowSpec<3840, 3, 384, 12, 10, 16>, KIND_REDFT10, true>("ROW 3840x3 T=384 12,10,16");
		run<RowSpec<3840, 3, 384, 12, 10, 16>, KIND_REDFT01, true>("ROW 3840x3 T=384 12,10,16");
	}
	run<RowSpec<3840, 3, 512, 12, 10, 16>, KIND_REDFT10, true>("ROW 3840x3 T=512");
	run<RowSpec<3840, 3, 512, 12, 10, 16>, KIND_REDFT01, true>("ROW 3840x3 T=512");
	if (getenv("KB_COLT")) {
		run<ColSpec<2160, 8, 384, 12, 12, 15>, KIND_REDFT10, false>("COL 2160 K=8 T=384 12,12,15");
		run<ColSpec<2160, 8, 384, 12, 12, 15>, KIND_REDFT01, false>("COL 2160 K=8 T=384 12,12,15");
		run<ColSpec<2160, 8, 320, 12, 12, 15>, KIND_REDFT10, false>("COL 2160 K=8 T=320 12,12,15");
		run<ColSpec<2160, 8, 320, 12, 12, 15>, KIND_REDFT01, false>("COL 2160 K=8 T=320 12,12,15");
		run<ColSpec<2160, 8, 512, 9, 16, 15>, KIND_REDFT10, false>("COL 2160 K=8 T=512 9,16,15");
		run<ColSpec<2160, 8, 512, 9, 16, 15>, KIND_REDFT01, false>("COL 2160 K=8 T=512 9,16,15");
		run<ColSpec<2160, 8, 512, 16, 15, 9>, KIND_REDFT10, false>("COL 2160 K=8 T=512 16,15,9");
		run<ColSpec<2160, 8, 512, 16, 15, 9>, KIND_REDFT01, false>("COL 2160 K=8 T=512 16,15,9");
	}
	run<ColSpec<2160, 8, 512, 12, 12, 15>, KIND_REDFT10, false>("COL 2160 K=8 T=512");
	run<ColSpec<2160, 8, 512, 12, 12, 15>, KIND_REDFT01, false>("COL 2160 K=8 T=512");
	return 0;
}
