// rowchan.hip -- VERDICT r2 item 5 as a kernel: the row pass of frames whose interleaved line fills a CU's LDS on its own (7680 x 3
// floats, 3840 x 3 doubles: 92 KB, ONE workgroup per CU) run as CHANNEL LINES -- one workgroup per (line, channel), 31 KB, reading and
// writing its channel at a stride of three samples (RowChanSpecT in dct_spec.h).  Timed beside the shipped interleaved kernel and the
// planar kernel of the same length (the bound: same butterflies, contiguous lines), results compared with the interleaved kernel's.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -Idspfun_amd/csrc tools/rowchan.hip -o tools/rowchan
//   tools/rowchan            (8K float, then 4K double)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "dct_spec.h"
using namespace dspfft;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MAP 0: block b -> line b / G, channel b % G (the channels of a line land on G different XCDs)
// MAP 1: the G channel lines of a line go to the SAME XCD (blocks b, b + 8, ... share one), back to back, so that their stores meet in one L2
template <int G, int MAP> __device__ inline void chan_work(int b, int lines, int &line, int &ch)
{
	if (MAP == 0) { line = b / G; ch = b - line * G; return; }
	const int full = (lines >> 3) << 3;
	if (b < full * G) { const int x = b & 7, j = b >> 3, q = j / G; line = q * 8 + x; ch = j - q * G; }
	else { const int r = b - full * G; line = full + r / G; ch = r - (r / G) * G; }
}

template <class S, int KIND, int MAP>
__global__ void __launch_bounds__(S::T, S::WPE) row_k(const typename S::PA a)
{
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	typename S::CX *planes = reinterpret_cast<typename S::CX *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	int line = blockIdx.x, ch = 0;
	if constexpr (S::GS != S::C) chan_work<S::GS, MAP>(blockIdx.x, a.nb0 * a.nb1, line, ch);
	row_base(a, line, bin, bout);
	bin += ch; bout += ch;
	S::template prefetch<KIND>(a, bin, tid, st, nullptr, nullptr, ch);
	S::template phase<KIND, 0>(a, planes, bout, tid, st);
	__syncthreads();
	static_for<1, S::NPH>([&](auto ph) {
		S::template phase<KIND, ph>(a, planes, bout, tid, st);
		if constexpr (ph + 1 < S::NPH) __syncthreads();
	});
}

template <class Re> struct Tab { cx<Re> *T, *W; };
template <class Re> static Tab<Re> make_tables(int N, int L)
{
	std::vector<cx<Re>> T(N + 1), Wv(L);
	for (int j = 0; j <= N; j++) T[j] = cmk((Re)cos(M_PI * j / (2.0 * N)), (Re)-sin(M_PI * j / (2.0 * N)));
	for (int t = 0; t < L; t++) Wv[t] = cmk((Re)cos(2 * M_PI * t / L), (Re)-sin(2 * M_PI * t / L));
	Tab<Re> r;
	CHK(hipMalloc(&r.T, T.size() * sizeof(cx<Re>))); CHK(hipMalloc(&r.W, Wv.size() * sizeof(cx<Re>)));
	CHK(hipMemcpy(r.T, T.data(), T.size() * sizeof(cx<Re>), hipMemcpyHostToDevice)); CHK(hipMemcpy(r.W, Wv.data(), Wv.size() * sizeof(cx<Re>), hipMemcpyHostToDevice));
	return r;
}

// nbuf > 1: the launches rotate over nbuf frames `fstride` samples apart (in place: what a clip does to the caches); extra_lds: dynamic
// LDS asked for beyond the spec's own, to hold the number of resident workgroups per CU down
template <class S, int KIND, int MAP>
static float run(const typename S::PA &a0, int wgs, int reps, int nbuf = 1, size_t fstride = 0, size_t extra_lds = 0)
{
	auto k = row_k<S, KIND, MAP>;
	const size_t lds = S::LDS + extra_lds;
	CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	auto go = [&](int i) { typename S::PA a = a0; a.in += (size_t)(i % nbuf) * fstride; a.out += (size_t)(i % nbuf) * fstride; hipLaunchKernelGGL(k, dim3(wgs), dim3(S::T), lds, 0, a); };
	for (int i = 0; i < 2 * nbuf; i++) go(i);
	CHK(hipEventRecord(e0, 0));
	for (int i = 0; i < reps; i++) go(i);
	CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
	float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
	CHK(hipGetLastError());
	return ms * 1000 / reps;
}

// IL: shipped interleaved spec; PL: planar spec of the same length; CH: channel lines
template <class Re, class IL, class PL, class CH>
static void bench(const char *name, int H)
{
	constexpr int W = IL::N, C = 3;
	const size_t NF = (size_t)H * W * C;
	Tab<Re> tb = make_tables<Re>(W, W / 2);
	std::vector<Re> h(NF);
	unsigned long long s = 12345;
	for (size_t i = 0; i < NF; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; h[i] = (Re)((double)(s >> 40) / 16777216.0 - 0.5); }
	Re *src, *o1, *o2;
	CHK(hipMalloc(&src, NF * sizeof(Re))); CHK(hipMalloc(&o1, NF * sizeof(Re))); CHK(hipMalloc(&o2, NF * sizeof(Re)));
	CHK(hipMemcpy(src, h.data(), NF * sizeof(Re), hipMemcpyHostToDevice));
	for (int kind = 0; kind < 2; kind++) {
		typename IL::PA a; memset((void *)&a, 0, sizeof a);
		a.in = src; a.out = o1; a.in_scale0 = (Re)0.5; a.out_scale0 = (Re)0.7; a.scale = (Re)(1.0 / W);
		a.kind = kind ? KIND_REDFT01 : KIND_REDFT10;
		a.N = W; a.C = C; a.nb0 = H; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)W * C; a.sb1_in = a.sb1_out = (long long)NF;
		a.T = tb.T; a.W = tb.W;
		typename IL::PA p = a;          // planar: the same bytes as 3 H contiguous lines (timing only)
		p.C = 1; p.nb0 = 3 * H; p.sb0_in = p.sb0_out = W; p.out = o2;
		typename IL::PA c = a; c.out = o2;
		const float t_il = kind ? run<IL, 1, 0>(a, H, 20) : run<IL, 0, 0>(a, H, 20);
		const float t_pl = kind ? run<PL, 1, 0>(p, 3 * H, 20) : run<PL, 0, 0>(p, 3 * H, 20);
		const float t_c0 = kind ? run<CH, 1, 0>(c, 3 * H, 20) : run<CH, 0, 0>(c, 3 * H, 20);
		const float t_c1 = kind ? run<CH, 1, 1>(c, 3 * H, 20) : run<CH, 0, 1>(c, 3 * H, 20);
		// in place as the plans run it (the channel lines of a line read and write the same cache lines at different times)
		typename IL::PA ci = c; ci.in = o2; ci.out = o2;
		const float t_ci = kind ? run<CH, 1, 1>(ci, 3 * H, 20) : run<CH, 0, 1>(ci, 3 * H, 20);
		if (kind) run<CH, 1, 1>(c, 3 * H, 1); else run<CH, 0, 1>(c, 3 * H, 1);      // out of place again for the comparison
		std::vector<Re> r1(NF), r2(NF);
		CHK(hipMemcpy(r1.data(), o1, NF * sizeof(Re), hipMemcpyDeviceToHost)); CHK(hipMemcpy(r2.data(), o2, NF * sizeof(Re), hipMemcpyDeviceToHost));
		double md = 0, mx = 0;
		for (size_t i = 0; i < NF; i++) { md = fmax(md, fabs((double)r1[i] - (double)r2[i])); mx = fmax(mx, fabs((double)r1[i])); }
		const double gb = 2.0 * NF * sizeof(Re) / 1e9;
		printf("%s %s: interleaved %.1f us (%.2f TB/s) | planar x3 %.1f us | channel lines: plain map %.1f us, same-XCD map %.1f us (%.2f TB/s), in place %.1f us | max |diff| %.3g of %.3g\n",
		       name, kind ? "REDFT01" : "REDFT10", t_il, gb / t_il * 1e3, t_pl, t_c0, t_c1, gb / t_c1 * 1e3, t_ci, md, mx);
	}
	CHK(hipFree(src)); CHK(hipFree(o1)); CHK(hipFree(o2));
	// in place over frames that do not fit the Infinity Cache together (>= 398 MB), resident workgroups held down by extra LDS
	const int nbuf = (int)((420e6 + NF * sizeof(Re) - 1) / (NF * sizeof(Re)));
	Re *clip; CHK(hipMalloc(&clip, NF * sizeof(Re) * nbuf)); CHK(hipMemset(clip, 0, NF * sizeof(Re) * nbuf));
	for (int kind = 0; kind < 2; kind++) {
		typename IL::PA a; memset((void *)&a, 0, sizeof a);
		a.in = clip; a.out = clip; a.in_scale0 = (Re)0.5; a.out_scale0 = (Re)0.7; a.scale = (Re)(1.0 / W);
		a.kind = kind ? KIND_REDFT01 : KIND_REDFT10;
		a.N = W; a.C = C; a.nb0 = H; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)W * C; a.sb1_in = a.sb1_out = (long long)NF;
		a.T = tb.T; a.W = tb.W;
		const float t_il = kind ? run<IL, 1, 0>(a, H, 24, nbuf, NF) : run<IL, 0, 0>(a, H, 24, nbuf, NF);
		printf("%s %s in place over %d frames: interleaved %.1f us | channel lines", name, kind ? "REDFT01" : "REDFT10", nbuf, t_il);
		for (size_t extra : {(size_t)0, (size_t)2048, (size_t)10240, (size_t)22528, (size_t)50000}) {
			const float t = kind ? run<CH, 1, 1>(a, 3 * H, 24, nbuf, NF, extra) : run<CH, 0, 1>(a, 3 * H, 24, nbuf, NF, extra);
			printf(" %d/CU: %.1f us", (int)(160 * 1024 / (CH::LDS + extra)), t);
		}
		printf("\n");
	}
	CHK(hipFree(clip));
}

int main()
{
	bench<float, RowSpecT<float, 7680, 3, 1024, 16, 15, 16>, RowSpecT<float, 7680, 1, 256, 16, 15, 16>, RowChanSpecT<float, 7680, 3, 256, 16, 15, 16>>("7680x4320x3 f32", 4320);
	bench<float, RowSpecT<float, 3840, 3, 512, 12, 10, 16>, RowSpecT<float, 3840, 1, 256, 12, 10, 16>, RowChanSpecT<float, 3840, 3, 256, 12, 10, 16>>("3840x2160x3 f32", 2160);
	bench<double, RowSpecT<double, 3840, 3, 512, 12, 10, 16>, RowSpecT<double, 3840, 1, 256, 12, 10, 16>, RowChanSpecT<double, 3840, 3, 256, 12, 10, 16>>("3840x2160x3 f64", 2160);
	return 0;
}
