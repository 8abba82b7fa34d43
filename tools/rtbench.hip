// rtbench.hip -- experiment harness (not part of the product): where the fused column roundtrip (spec_kernels.h col_roundtrip_kernel,
// motion's default per-frame mode) spends a workgroup's life.  Same phases, s_memtime stamps between them; 64 luma frames of 1920x1080.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -ffp-contract=on -std=c++17 -Idspfun_amd/csrc tools/rtbench.hip -o tools/rtbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "spec_kernels.h"
using namespace dspfft;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define STAMP(v) do { asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); } while (0)
constexpr int MAXST = 20;
struct Stamps { unsigned long long t[MAXST]; };

template <class S, bool STAMPS, bool FILTER>
__global__ void __launch_bounds__(S::T, rt_waves_per_simd<S>()) rt_k(const typename S::PA af, const typename S::PA ai, const FilterOp filt, unsigned long long *coded, Stamps *dbg)
{
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	typename S::V *buf = reinterpret_cast<typename S::V *>(lds);
	const int tid = threadIdx.x;
	typename S::StateRT st;
	unsigned long long ts[MAXST] = {};
	int ns = 0;
	long long bin, bout;
	S::base(af, blockIdx.x, bin, bout);
	if constexpr (STAMPS) STAMP(ts[ns++]);
	bool hit = false;
	S::template prefetch<KIND_REDFT10>(af, bin, tid, st, hit);
	if constexpr (STAMPS) STAMP(ts[ns++]);
	S::template phase<KIND_REDFT10, 0>(af, buf, bout, tid, st);
	__syncthreads();
	if constexpr (STAMPS) STAMP(ts[ns++]);
	static_for<1, S::NS + 2>([&](auto ph) {
		int t = tid; asm volatile("" : "+v"(t));
		S::template phase<KIND_REDFT10, ph>(af, buf, bout, t, st);
		__syncthreads();
		if constexpr (STAMPS) STAMP(ts[ns++]);
	});
	unsigned long long mine = 0;
	int t = tid; asm volatile("" : "+v"(t));
	S::mid_read(af, ai, buf, bout, t, st, filt, mine);
	__syncthreads();
	if constexpr (STAMPS) STAMP(ts[ns++]);
	asm volatile("" : "+v"(t));
	S::mid_write(buf, t, st);
	__syncthreads();
	if constexpr (STAMPS) STAMP(ts[ns++]);
	typename S::PA a2 = ai;
	asm volatile("" : "+s"(a2.W), "+s"(a2.T), "+s"(a2.out));
	static_for<1, S::NPH>([&](auto ph) {
		asm volatile("" : "+v"(t));
		S::template phase<KIND_REDFT01, ph>(a2, buf, bout, t, st);
		if constexpr (ph + 1 < S::NPH) __syncthreads();
		if constexpr (STAMPS) { if constexpr (ph + 1 == S::NPH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(ts[ns++]); }
	});
	if (coded && mine == 0xffffffffffffull) *coded = mine;
	if constexpr (STAMPS) { if (tid == 0) { Stamps w; for (int i = 0; i < MAXST; i++) w.t[i] = ts[i]; dbg[blockIdx.x] = w; } }
}

static const int H = 1080, W = 1920, FR = 64;
template <class S>
static int bench(const char *label)
{
	printf("== %s\n", label);
	float *buf; CHK(hipMalloc(&buf, (size_t)H * W * FR * 4));
	{ std::vector<float> h((size_t)H * W * FR); for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) % 255); CHK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
	std::vector<cf> T(H + 1), Wv(H);
	for (int j = 0; j <= H; j++) T[j] = cmk((float)cos(M_PI * j / (2.0 * H)), (float)-sin(M_PI * j / (2.0 * H)));
	for (int t = 0; t < H; t++) Wv[t] = cmk((float)cos(2 * M_PI * t / H), (float)-sin(2 * M_PI * t / H));
	cf *dT, *dW; CHK(hipMalloc(&dT, T.size() * 8)); CHK(hipMalloc(&dW, Wv.size() * 8));
	CHK(hipMemcpy(dT, T.data(), T.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dW, Wv.data(), Wv.size() * 8, hipMemcpyHostToDevice));
	PassArgs a; memset((void *)&a, 0, sizeof a);
	a.in = buf; a.out = buf; a.N = H; a.T = dT; a.W = dW; a.in_scale0 = a.out_scale0 = 1.f;
	a.K = S::K; a.B = S::B; a.ninner = W; a.ntiles = W / S::K; a.es_in = a.es_out = W; a.nb0 = FR; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)H * W;
	PassArgs af = a, ai = a; af.kind = KIND_REDFT10; af.scale = 1.f; ai.kind = KIND_REDFT01; ai.scale = 1.f / (2.f * H);
	const int nwork = a.ntiles * FR;
	FilterOp f; memset(&f, 0, sizeof f);
	f.p.ad = 1; f.p.ah = H; f.p.aw = W; f.p.mh = H; f.p.mw = W; f.p.b1d = 1; f.p.b1h = H; f.p.b1w = W; f.p.damp = f.p.boost = 1.f; f.p.quantizer = 3.f; f.p.enabled = 1;
	motion_filter_set_divs(f.p, 1);
	Stamps *dbg; CHK(hipMalloc(&dbg, sizeof(Stamps) * nwork));
	auto run = [&](auto kern, const char *name, bool stamps) {
		CHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS));
		hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
		float best = 1e9;
		for (int rep = 0; rep < 4; rep++) {
			CHK(hipEventRecord(e0));
			hipLaunchKernelGGL(kern, dim3(nwork), dim3(S::T), S::LDS, 0, af, ai, f, nullptr, dbg);
			CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
			float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
		}
		CHK(hipGetLastError());
		printf("%-28s %d tiles of 1080 x 16: %.1f us = %.2f TB/s (read + write)\n", name, nwork, best * 1e3, 2.0 * H * W * FR * 4 / best / 1e9);
		if (stamps) {
			std::vector<Stamps> h(nwork);
			CHK(hipMemcpy(h.data(), dbg, sizeof(Stamps) * nwork, hipMemcpyDeviceToHost));
			const int NST = 3 + (S::NS + 1) + 2 + (S::NPH - 1);
			std::vector<double> seg(NST, 0.0);
			for (auto &w : h) for (int i = 1; i < NST; i++) seg[i] += (double)(w.t[i] - w.t[i - 1]);
			double tot = 0; for (int i = 1; i < NST; i++) tot += seg[i];
			printf("    mean workgroup life %.0f clocks of s_memtime (100 MHz: %.1f us); share per segment:", tot / nwork, tot / nwork / 100.0);
			const char *names[] = {"", "issue-loads", "wait+ph0", "f-stage1", "f-stage2", "f-last-read", "f-last-write", "mid-read(filter)", "mid-write", "i-stage1", "i-stage2", "i-last-read", "i-last-write", "i-unpack+store"};
			for (int i = 1; i < NST; i++) printf(" [%s %.0f%%]", i < 14 ? names[i] : "?", 100.0 * seg[i] / tot);
			printf("\n");
		}
	};
	run(rt_k<S, false, true>, "fused roundtrip", false);
	run(rt_k<S, true, true>, "fused roundtrip (stamped)", true);
	CHK(hipFree(buf)); CHK(hipFree(dT)); CHK(hipFree(dW)); CHK(hipFree(dbg));
	return 0;
}

int main()
{
	bench<ColSpec<1080, 16, 512, 8, 9, 15>>("radices 8, 9, 15 (product)");
	bench<ColSpec<1080, 16, 512, 10, 12, 9>>("radices 10, 12, 9");
	bench<ColSpec<1080, 16, 512, 12, 10, 9>>("radices 12, 10, 9");
	bench<ColSpec<1080, 16, 512, 9, 10, 12>>("radices 9, 10, 12");
	bench<ColSpec<1080, 16, 512, 8, 15, 9>>("radices 8, 15, 9");
	return 0;
}
