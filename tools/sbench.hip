// sbench.hip -- schedule experiments for the 4K roundtrip (not part of the product): the four specialised passes
// (ROW10, COL10, COL01, ROW01) of 4 frames, launched under different multi-stream schedules, all in ONE process with
// interleaved rounds (cdna_hip_programming.md rule 24).
//   single      one stream, frame after frame
//   aligned     two streams, one frame each, streams re-synchronised every step (bench.py's default)
//   free        two streams, free running
//   lagK        two streams, stream B's frame starts when stream A has finished K passes of its frame (K = 1, 2)
//   lockstep    two streams, re-synchronised before every pass (same pass shape always co-runs)
//   batch2      one stream, two frames per launch
//   batch4      one stream, four frames per launch
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -Idspfun_amd/csrc tools/sbench.hip -o tools/sbench
//        (add -DDSP_COL_PADC=0 / -DDSP_ROW_PADC=0 for the no-pad LDS layouts: 54 / 36 allocation granules of 1280 B)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>
#include <algorithm>
#include "dct_spec.h"
using namespace dspfft;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <class S, int KIND, bool ROWK>
__global__ void __launch_bounds__(S::T, S::WPE) pass_k(const PassArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	if constexpr (ROWK) row_base(a, blockIdx.x, bin, bout); else S::base(a, blockIdx.x, bin, bout);
	auto *buf = reinterpret_cast<std::conditional_t<ROWK, cf, SigVec<float, 2>> *>(lds);
	S::template prefetch<KIND>(a, bin, tid, st);
	S::template phase<KIND, 0>(a, buf, bout, tid, st);
	__syncthreads();
	static_for<1, S::NPH>([&](auto ph) {
		S::template phase<KIND, ph>(a, buf, bout, tid, st);
		if constexpr (ph + 1 < S::NPH) __syncthreads();
	});
}

#ifdef CS_DMA
// -DCS_DMA: the REAL column REDFT10 kernel with its tile loaded by LDS-DMA (global_load_lds_dwordx4) instead of
// through registers.  Since round 3 a lane's 16 loaded bytes are already the LDS slot format (columns (c0, c2) / (c1, c3) paired), so the
// DMA needs no conversion: lane q of a wave instruction fetches padded slot q's source row -- makhoul_src of the slot's row, the even/odd
// reorder applied on the SOURCE address -- and the 64 lanes land in 64 consecutive slots (1 KiB).  Pad slots fetch row 0 (junk nobody
// reads).  State::pre and phase 0's ds_write_b128 pass disappear.  (in_scale0 is 1 in this bench.)
template <class S, int KIND>
__global__ void __launch_bounds__(S::T, S::WPE) pass_k_dma(const PassArgs a)
{
	static_assert(KIND == KIND_REDFT10, "DMA variant: forward column pass");
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	typedef SigVec<float, 2> V;
	V *buf = reinterpret_cast<V *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	S::base(a, blockIdx.x, bin, bout);
	constexpr int SLOTS = S::ROWS * S::NP, ROUNDS = (SLOTS + S::T - 1) / S::T, BLK = S::SB + S::PADC;
#pragma unroll
	for (int r = 0; r < ROUNDS; r++) {
		const int q = tid + r * S::T;
		if (q - (tid & 63) < SLOTS) {                        // wave-uniform: the wave's first slot exists
			const int qq = q < SLOTS ? q : SLOTS - 1;
			const int rowp = qq / S::NP, jp = qq - rowp * S::NP;
			const int blk = rowp / BLK, rr = rowp - blk * BLK;
			const int n = rr < S::SB ? blk * S::SB + rr : 0;     // pad row: any valid source
			const int y = makhoul_src(n, S::N);
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.in + bin + (long long)y * a.es_in + S::VW * jp),
			                                 (__attribute__((address_space(3))) void *)(buf + (q & ~63)), 16, 0, 0);
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	static_for<1, S::NPH>([&](auto ph) {
		S::template phase<KIND, ph>(a, buf, bout, tid, st);
		if constexpr (ph + 1 < S::NPH) __syncthreads();
	});
}
#endif

#ifdef SPLIT
#include <hip/hip_runtime.h>
#include "backend.h"
#include "spec_list.h"
#include "spec_kernels.h"
typedef ColHalfSpec<2160, 16, 512, 12, 10, 9> HS;
static cf *g_WM, *g_Hh;
#endif
static const int H = 2160, W = 3840, C = 3;
static const size_t NF = (size_t)H * W * C;
static int NFR = 4;            // frames per step (6 for the three-stream experiment)
static const int NFR_MAX = 6;
static float *g_buf;

#ifdef PAIR768
typedef RowSpec<3840, 3, 768, 12, 10, 16> RS;
#else
typedef RowSpec<3840, 3, 512, 12, 10, 16> RS;
#endif
// column-tile variants measured in round 3: -DCS_K4: 2160 x 4-float tiles, 34.7 KB, four 256-thread workgroups per CU;
// -DCS_2STAGE: two stages 48 x 45 (one LDS round trip fewer; 96 butterflies per stage -> 128 threads); -DCS_4STAGE: 6 x 6 x 6 x 10
// (more, shorter butterflies: 720 / 432 work items per stage for 512 threads)
#if defined(CS_K4)
typedef ColSpec<2160, 4, 256, 12, 12, 15> CS;
#elif defined(CS_2STAGE)
typedef ColSpec<2160, 8, 128, 48, 45> CS;
#elif defined(CS_2STAGE_K16)
typedef ColSpec<2160, 16, 256, 48, 45> CS;
#elif defined(CS_4STAGE)
typedef ColSpec<2160, 8, 512, 6, 6, 6, 10> CS;
#elif defined(CS_4STAGE_768)
typedef ColSpec<2160, 8, 768, 6, 6, 6, 10> CS;
#elif defined(CS_4STAGE_B)
typedef ColSpec<2160, 8, 512, 8, 6, 5, 9> CS;
#else
typedef ColSpec<2160, 8, 512, 12, 12, 15> CS;
#endif

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
static Tables g_trow, g_tcol;

#ifdef USELIB
// -DUSELIB (link with -ldspfft_hip): the same schedules with the LIBRARY's launches (dspfft_execute_pass) instead of this file's kernels
#include "../include/dspfft.h"
static dspfft_plan g_fwd, g_inv;
#endif
// pass p (0 ROW10, 1 COL10, 2 COL01, 3 ROW01) over `nfr` consecutive frames starting at frame f0
static void launch_pass(int p, int f0, int nfr, hipStream_t s)
{
#ifdef USELIB
	if (nfr == 1) {
		if (!g_fwd) {
			const int dims[2] = {H, W}, k10[2] = {DSPFFT_REDFT10, DSPFFT_REDFT10}, k01[2] = {DSPFFT_REDFT01, DSPFFT_REDFT01};
			if (dspfft_plan_many_r2r(&g_fwd, 2, dims, C, NULL, C, 1, NULL, C, 1, k10) || dspfft_plan_many_r2r_ordered(&g_inv, 2, dims, C, NULL, C, 1, NULL, C, 1, k01, 1)) { printf("plan failed\n"); exit(1); }
			dspfft_plan_set_scale(g_inv, 1.0f / (4.0f * W * H));
		}
		float *b = g_buf + (size_t)f0 * NF;
		if (dspfft_execute_pass(p <= 1 ? g_fwd : g_inv, p & 1, b, b, s)) { printf("execute_pass failed: %s\n", dspfft_last_error()); exit(1); }
		return;
	}
#endif
	PassArgs a; memset((void *)&a, 0, sizeof a);
	a.in = a.out = g_buf + (size_t)f0 * NF;
	a.in_scale0 = a.out_scale0 = 1.f;
	const bool row = (p == 0 || p == 3);
	a.kind = (p <= 1) ? KIND_REDFT10 : KIND_REDFT01;
	a.scale = (p <= 1) ? 1.f : (row ? 1.f / (2.f * W) : 1.f / (2.f * H));
#ifdef SPLIT
	if (row) {
		a.N = W; a.C = C; a.nb0 = H; a.nb1 = nfr; a.sb0_in = a.sb0_out = (long long)W * C; a.sb1_in = a.sb1_out = (long long)NF;
		a.T = g_trow.T; a.W = g_trow.Wt;
		if (p == 0) hipLaunchKernelGGL((row_pair_kernel<RS, KIND_REDFT10>), dim3(H / 2 * nfr), dim3(RS::T), RS::LDS, s, a);
		else hipLaunchKernelGGL((row_pair_kernel<RS, KIND_REDFT01>), dim3(H / 2 * nfr), dim3(RS::T), RS::LDS, s, a);
	} else {
		a.N = H; a.K = HS::K; a.B = HS::K / 2; a.ninner = W * C; a.ntiles = W * C / HS::K; a.es_in = a.es_out = (long long)W * C;
		a.nb0 = nfr; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)NF;
		a.T = g_tcol.T; a.W = g_WM; a.H = g_Hh;
		if (p == 1) hipLaunchKernelGGL((col_half_kernel<HS, KIND_REDFT10>), dim3(2 * a.ntiles * nfr), dim3(HS::T), HS::LDS, s, a);
		else hipLaunchKernelGGL((col_half_kernel<HS, KIND_REDFT01>), dim3(2 * a.ntiles * nfr), dim3(HS::T), HS::LDS, s, a);
	}
	return;
#endif
	if (row) {
		a.N = W; a.C = C; a.nb0 = H; a.nb1 = nfr; a.sb0_in = a.sb0_out = (long long)W * C; a.sb1_in = a.sb1_out = (long long)NF;
		a.T = g_trow.T; a.W = g_trow.Wt;
		if (p == 0) hipLaunchKernelGGL((pass_k<RS, KIND_REDFT10, true>), dim3(H * nfr), dim3(RS::T), RS::LDS, s, a);
		else hipLaunchKernelGGL((pass_k<RS, KIND_REDFT01, true>), dim3(H * nfr), dim3(RS::T), RS::LDS, s, a);
	} else {
		a.N = H; a.K = CS::K; a.B = CS::B; a.ninner = W * C; a.ntiles = W * C / CS::K; a.es_in = a.es_out = (long long)W * C;
		a.nb0 = nfr; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)NF;
		a.T = g_tcol.T; a.W = g_tcol.Wt;
#ifdef CS_DMA
		if (p == 1) hipLaunchKernelGGL((pass_k_dma<CS, KIND_REDFT10>), dim3(a.ntiles * nfr), dim3(CS::T), CS::LDS + 1024, s, a);
#else
		if (p == 1) hipLaunchKernelGGL((pass_k<CS, KIND_REDFT10, false>), dim3(a.ntiles * nfr), dim3(CS::T), CS::LDS, s, a);
#endif
		else hipLaunchKernelGGL((pass_k<CS, KIND_REDFT01, false>), dim3(a.ntiles * nfr), dim3(CS::T), CS::LDS, s, a);
	}
}

// spins for about `us` microseconds (s_memrealtime counts at 100 MHz): one wave, used to stagger stream B behind stream A
__global__ void delay_k(int us)
{
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(32);
}
static int g_lag_us = 20, g_join_every = 8, g_stepno = 0;
static hipStream_t sA, sB;
static hipEvent_t evA[8], evB[8], evStepA, evStepB;

static void step(const std::string &mode)
{
	if (mode == "single") {
		for (int f = 0; f < NFR; f++) for (int p = 0; p < 4; p++) launch_pass(p, f, 1, sA);
	} else if (mode == "batch2") {
		for (int f = 0; f < NFR; f += 2) for (int p = 0; p < 4; p++) launch_pass(p, f, 2, sA);
	} else if (mode == "batch4") {
		for (int p = 0; p < 4; p++) launch_pass(p, 0, 4, sA);
	} else if (mode == "free") {
		for (int f = 0; f < NFR; f++) for (int p = 0; p < 4; p++) launch_pass(p, f, 1, (f & 1) ? sB : sA);
	} else if (mode == "aligned") {
		CHK(hipStreamWaitEvent(sA, evStepB, 0)); CHK(hipStreamWaitEvent(sB, evStepA, 0));
		for (int f = 0; f < NFR; f++) for (int p = 0; p < 4; p++) launch_pass(p, f, 1, (f & 1) ? sB : sA);
		CHK(hipEventRecord(evStepA, sA)); CHK(hipEventRecord(evStepB, sB));
	} else if (mode == "free3") {
		// three streams, one frame each in flight: 298 MB hot, more than the 256 MB Infinity Cache holds
		static hipStream_t sC = nullptr;
		if (!sC) CHK(hipStreamCreateWithFlags(&sC, hipStreamNonBlocking));
		for (int f = 0; f < NFR; f++) for (int p = 0; p < 4; p++) launch_pass(p, f, 1, f % 3 == 0 ? sA : f % 3 == 1 ? sB : sC);
		if (g_stepno++ % 8 == 7) { CHK(hipEventRecord(evA[7], sC)); CHK(hipStreamWaitEvent(sA, evA[7], 0)); }
	} else if (mode == "stagger") {
		// streams re-joined every g_join_every steps; after a join stream B starts g_lag_us behind stream A
		if (g_stepno % g_join_every == 0) {
			CHK(hipStreamWaitEvent(sA, evStepB, 0)); CHK(hipStreamWaitEvent(sB, evStepA, 0));
			if (g_lag_us > 0) hipLaunchKernelGGL(delay_k, dim3(1), dim3(64), 0, sB, g_lag_us);
		}
		for (int f = 0; f < NFR; f++) for (int p = 0; p < 4; p++) launch_pass(p, f, 1, (f & 1) ? sB : sA);
		g_stepno++;
		if (g_stepno % g_join_every == 0) { CHK(hipEventRecord(evStepA, sA)); CHK(hipEventRecord(evStepB, sB)); }
	} else if (mode == "lockstep") {
		for (int f = 0; f < NFR; f += 2)
			for (int p = 0; p < 4; p++) {
				CHK(hipStreamWaitEvent(sA, evStepB, 0)); CHK(hipStreamWaitEvent(sB, evStepA, 0));
				launch_pass(p, f, 1, sA); launch_pass(p, f + 1, 1, sB);
				CHK(hipEventRecord(evStepA, sA)); CHK(hipEventRecord(evStepB, sB));
			}
	} else if (mode == "lag1" || mode == "lag2") {
		const int lag = mode == "lag1" ? 1 : 2;
		// stream A: frames 0, 2; stream B: frames 1, 3; B's first frame of the step starts after A has done `lag` passes
		for (int f = 0; f < NFR; f += 2) {
			for (int p = 0; p < 4; p++) {
				launch_pass(p, f, 1, sA);
				if (f == 0 && p == lag - 1) CHK(hipEventRecord(evA[0], sA));
			}
		}
		CHK(hipStreamWaitEvent(sB, evA[0], 0));
		for (int f = 1; f < NFR; f += 2) for (int p = 0; p < 4; p++) launch_pass(p, f, 1, sB);
		// the next step's stream A waits for B to be `lag` passes from finishing its last frame: keep it simple -- A runs free
	} else if (mode == "pipe") {
		// strict complementary pipeline: B's pass p of frame f+1 starts when A's pass p of frame f is done, and vice versa
		for (int f = 0; f < NFR; f += 2)
			for (int p = 0; p < 4; p++) {
				CHK(hipStreamWaitEvent(sA, evB[p], 0));
				launch_pass(p, f, 1, sA);
				CHK(hipEventRecord(evA[p], sA));
				CHK(hipStreamWaitEvent(sB, evA[p], 0));
				launch_pass(p, f + 1, 1, sB);
				CHK(hipEventRecord(evB[(p + 1) & 3], sB));
			}
	}
}

static double run_mode(const std::string &mode, int steps)
{
	for (int i = 0; i < 3; i++) step(mode);
	CHK(hipDeviceSynchronize());
	hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
	CHK(hipEventRecord(a, sA));
	CHK(hipStreamWaitEvent(sB, a, 0));
	for (int i = 0; i < steps; i++) step(mode);
	CHK(hipEventRecord(evStepB, sB)); CHK(hipStreamWaitEvent(sA, evStepB, 0));
	CHK(hipEventRecord(b, sA)); CHK(hipEventSynchronize(b));
	float ms; CHK(hipEventElapsedTime(&ms, a, b));
	CHK(hipGetLastError());
	CHK(hipDeviceSynchronize());
	CHK(hipEventDestroy(a)); CHK(hipEventDestroy(b));
	return (double)steps * NFR * H * W / 1e6 / (ms * 1e-3);
}

int main(int argc, char **argv)
{
	const int steps = argc > 1 ? atoi(argv[1]) : 100;
	CHK(hipMalloc(&g_buf, NF * 4 * NFR_MAX));
	{
		std::vector<float> h(NF);
		for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f;
		for (int f = 0; f < NFR_MAX; f++) CHK(hipMemcpy(g_buf + (size_t)f * NF, h.data(), NF * 4, hipMemcpyHostToDevice));
	}
	g_trow = make_tables(W, W / 2); g_tcol = make_tables(H, H);
#ifdef SPLIT
	{
		const int M = H / 2;
		std::vector<cf> Wv(M), Hv(M);
		for (int t = 0; t < M; t++) { Wv[t] = cmk((float)cos(2 * M_PI * t / M), (float)-sin(2 * M_PI * t / M)); Hv[t] = cmk((float)cos(2 * M_PI * t / H), (float)-sin(2 * M_PI * t / H)); }
		CHK(hipMalloc(&g_WM, M * 8)); CHK(hipMalloc(&g_Hh, M * 8));
		CHK(hipMemcpy(g_WM, Wv.data(), M * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(g_Hh, Hv.data(), M * 8, hipMemcpyHostToDevice));
		CHK(hipFuncSetAttribute((const void *)row_pair_kernel<RS, KIND_REDFT10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RS::LDS));
		CHK(hipFuncSetAttribute((const void *)row_pair_kernel<RS, KIND_REDFT01>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RS::LDS));
		CHK(hipFuncSetAttribute((const void *)col_half_kernel<HS, KIND_REDFT10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HS::LDS));
		CHK(hipFuncSetAttribute((const void *)col_half_kernel<HS, KIND_REDFT01>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HS::LDS));
		printf("SPLIT build: row pairs + half tiles %dx%d (LDS %zu B)\n", HS::M, HS::K, (size_t)HS::LDS);
	}
#endif
	CHK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
	for (int i = 0; i < 8; i++) { CHK(hipEventCreateWithFlags(&evA[i], hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&evB[i], hipEventDisableTiming)); CHK(hipEventRecord(evA[i], sA)); CHK(hipEventRecord(evB[i], sB)); }
	CHK(hipEventCreateWithFlags(&evStepA, hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&evStepB, hipEventDisableTiming));
	CHK(hipEventRecord(evStepA, sA)); CHK(hipEventRecord(evStepB, sB));
	CHK(hipFuncSetAttribute((const void *)pass_k<RS, KIND_REDFT10, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RS::LDS));
	CHK(hipFuncSetAttribute((const void *)pass_k<RS, KIND_REDFT01, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RS::LDS));
	CHK(hipFuncSetAttribute((const void *)pass_k<CS, KIND_REDFT10, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CS::LDS));
	CHK(hipFuncSetAttribute((const void *)pass_k<CS, KIND_REDFT01, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CS::LDS));
#ifdef CS_DMA
	CHK(hipFuncSetAttribute((const void *)pass_k_dma<CS, KIND_REDFT10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CS::LDS + 1024));
	{	// correctness of the DMA-loaded pass against the register-staged one on frame 0 (then frame 0 is restored)
		std::vector<float> ref(NF), got(NF), keep(NF);
		CHK(hipMemcpy(keep.data(), g_buf, NF * 4, hipMemcpyDeviceToHost));
		PassArgs a; memset((void *)&a, 0, sizeof a);
		a.in = a.out = g_buf; a.in_scale0 = a.out_scale0 = 1.f; a.kind = KIND_REDFT10; a.scale = 1.f;
		a.N = H; a.K = CS::K; a.B = CS::B; a.ninner = W * C; a.ntiles = W * C / CS::K; a.es_in = a.es_out = (long long)W * C; a.nb0 = 1; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)NF;
		a.T = g_tcol.T; a.W = g_tcol.Wt;
		hipLaunchKernelGGL((pass_k<CS, KIND_REDFT10, false>), dim3(a.ntiles), dim3(CS::T), CS::LDS, 0, a);
		CHK(hipMemcpy(ref.data(), g_buf, NF * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(g_buf, keep.data(), NF * 4, hipMemcpyHostToDevice));
		hipLaunchKernelGGL((pass_k_dma<CS, KIND_REDFT10>), dim3(a.ntiles), dim3(CS::T), CS::LDS + 1024, 0, a);
		CHK(hipMemcpy(got.data(), g_buf, NF * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(g_buf, keep.data(), NF * 4, hipMemcpyHostToDevice));
		double e = 0, m = 0;
		for (size_t i = 0; i < NF; i++) { e = std::max(e, (double)fabsf(ref[i] - got[i])); m = std::max(m, (double)fabsf(ref[i])); }
		printf("CS_DMA: LDS-DMA column REDFT10 vs register-staged: max abs diff %.3g (max |value| %.3g)\n", e, m);
	}
#endif
	printf("ROW LDS %zu B (%.2f granules of 1280), COL LDS %zu B (%.2f granules); %d steps of %d frames\n", (size_t)RS::LDS, RS::LDS / 1280.0, (size_t)CS::LDS, CS::LDS / 1280.0, steps, NFR);
	{	// each pass alone, one frame (cache-resident)
		for (int p = 0; p < 4; p++) {
			for (int i = 0; i < 3; i++) launch_pass(p, 0, 1, sA);
			hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
			CHK(hipEventRecord(a, sA));
			for (int i = 0; i < 30; i++) launch_pass(p, 0, 1, sA);
			CHK(hipEventRecord(b, sA)); CHK(hipEventSynchronize(b));
			float ms; CHK(hipEventElapsedTime(&ms, a, b));
			printf("pass %d alone: %.1f us\n", p, ms * 1000 / 30);
		}
	}
	if (argc > 2 && !strcmp(argv[2], "three")) {
		for (int round = 0; round < 3; round++) {
			NFR = 4; const double f2 = run_mode("free", steps);
			NFR = 6; const double f2b = run_mode("free", steps); g_stepno = 0; const double f3 = run_mode("free3", steps);
			NFR = 2; const double f2c = run_mode("free", 2 * steps);
			NFR = 4;
			printf("round %d: two streams 4 frames/step %.0f, 6 frames/step %.0f, 2 frames/step (all cache-resident) %.0f; three streams 6 frames/step %.0f\n", round, f2, f2b, f2c, f3);
		}
		return 0;
	}
	if (argc > 2 && !strcmp(argv[2], "stagger")) {
		// does a free-running pair of streams keep its phase?  and which lag behind stream A is best for stream B?
		for (int round = 0; round < 2; round++) {
			printf("round %d: aligned=%.0f free(%d steps)=%.0f free(%d steps)=%.0f", round, run_mode("aligned", steps), steps, run_mode("free", steps), 10 * steps, run_mode("free", 10 * steps));
			for (int je : {1, 8, 64})
				for (int lag : {0, 5, 10, 15, 20, 25, 30, 40, 60}) {
					g_join_every = je; g_lag_us = lag; g_stepno = 0;
					printf(" j%d/lag%d=%.0f", je, lag, run_mode("stagger", 4 * steps)); fflush(stdout);
				}
			printf("\n");
		}
		return 0;
	}
	const char *modes[] = {"single", "aligned", "free", "lag1", "lag2", "lockstep", "pipe", "batch2", "batch4"};
	for (int round = 0; round < 3; round++) {
		printf("round %d:", round);
		for (const char *m : modes) { double v = run_mode(m, steps); printf(" %s=%.0f", m, v); fflush(stdout); }
		printf("  (Mpix/s)\n");
	}
	return 0;
}
