// rowseq.hip -- VERDICT r2 item 5 to the letter, as a kernel: "load the interleaved line once (all channels in registers), then run the
// channels ONE AFTER ANOTHER through a single N/2-complex LDS plane".  A 7680 x 3 float line keeps its 92 KB out of LDS: a 512-thread
// workgroup holds the line in registers (45 per thread), scatters one channel at a time into a 31 KB plane, runs that channel's stages,
// keeps the channel's outputs in registers (48 per thread) and stores whole pixels at the end -- two such workgroups share a CU (the
// shipped kernel: one 1024-thread workgroup, three planes).  REDFT10 row pass, in place over two 8K frames (HBM-resident), timed beside
// the shipped interleaved kernel; results compared.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -Idspfun_amd/csrc tools/rowseq.hip -o tools/rowseq
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
__global__ void __launch_bounds__(S::T, S::WPE) row_k(const typename S::PA a)
{
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	typename S::CX *planes = reinterpret_cast<typename S::CX *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	row_base(a, blockIdx.x, bin, bout);
	S::template prefetch<KIND>(a, bin, tid, st);
	S::template phase<KIND, 0>(a, planes, bout, tid, st);
	__syncthreads();
	static_for<1, S::NPH>([&](auto ph) {
		S::template phase<KIND, ph>(a, planes, bout, tid, st);
		if constexpr (ph + 1 < S::NPH) __syncthreads();
	});
}

// S1: the PLANAR spec of the line length (one channel plane in LDS); G channels in the interleaved line
template <class S1, int G, int WAVES>
__global__ void __launch_bounds__(S1::T, WAVES) row_seq10_kernel(const typename S1::PA a)
{
	typedef typename S1::Re Re;
	typedef typename S1::CX CX;
	constexpr int N = S1::N, L = S1::L, T = S1::T, PL = S1::PL;
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	CX *plane = reinterpret_cast<CX *>(lds);
	const int tid = threadIdx.x;
	long long bin, bout;
	row_base(a, blockIdx.x, bin, bout);
	Pix<G, Re> in[S1::PIX_ROUNDS];
	static_for<0, S1::PIX_ROUNDS>([&](auto i) {
		const int x = tid + i * T;
		if ((i + 1) * T <= N || x < N) in[i] = load_pix<G, Re>(a.in + bin + (long long)x * G);
	});
	Pix<G, Re> o[4 * S1::K_ROUNDS];
	static_for<0, G>([&](auto c) {
		int t = tid; asm volatile("" : "+v"(t));       // per channel: keeps index arithmetic from being hoisted and carried across the channels
		typename S1::template State<KIND_REDFT10> st;
		static_for<0, S1::PIX_ROUNDS>([&](auto i) { st.pre[i] = in[i].v[c]; });
		static_for<0, S1::NPH - 1>([&](auto ph) {
			S1::template phase<KIND_REDFT10, ph>(a, plane, bout, t, st);
			__syncthreads();
		});
		// the closing phase of REDFT10 (RowSpecG::phase<KIND_REDFT10, NS + 2>) with its four outputs per k kept in registers
		static_for<0, S1::K_ROUNDS>([&](auto ri) {
			const int k = t + ri * T;
			if (!((ri + 1) * T <= L / 2 + 1 || k <= L / 2)) return;
			const int km = k ? L - k : 0;
			const CX tk = a.T[k];
			const CX tlk = cmul(cconj(tk), cmk<Re>((Re)0.70710678118654752440, (Re)-0.70710678118654752440));
			const CX t1 = csqr(csqr(tk));
			const CX zk = plane[k];
			const CX zm = cconj(plane[km]);
			const CX E = cadd(zk, zm);
			const CX D = cmul_mi(csub(zk, zm));
			const CX P = cmul(t1, D);
			const CX wk = cmul(tk, cadd(E, P));
			const CX wm = cmul(tlk, cconj(csub(E, P)));
			const Re sc = a.scale;
			o[ri * 4 + 0].v[c] = wk.x * (k == 0 ? sc * a.out_scale0 : sc);
			o[ri * 4 + 1].v[c] = -wk.y * sc;
			o[ri * 4 + 2].v[c] = wm.x * sc;
			o[ri * 4 + 3].v[c] = -wm.y * sc;
		});
		__syncthreads();                                // the plane is the next channel's
	});
	static_for<0, S1::K_ROUNDS>([&](auto ri) {
		const int k = tid + ri * T;
		if (!((ri + 1) * T <= L / 2 + 1 || k <= L / 2)) return;
		store_pix<G, Re>(a.out + bout + (long long)k * G, o[ri * 4 + 0]);
		if (k > 0) store_pix<G, Re>(a.out + bout + (long long)(N - k) * G, o[ri * 4 + 1]);
		if (L - k != k) store_pix<G, Re>(a.out + bout + (long long)(L - k) * G, o[ri * 4 + 2]);
		if (k > 0 && L + k != N - k) store_pix<G, Re>(a.out + bout + (long long)(L + k) * G, o[ri * 4 + 3]);
	});
}

// the same with the channels in GROUPS: channels 0 and 1 together through two planes (SA: the C = 2 spec: 480-512 butterflies per stage for
// 512 threads), then channel 2 alone (SB: C = 1) -- 12 barrier phases per line instead of 18, 62 KB of LDS, still two workgroups per CU
template <class S, int C0, int G, class IN, class OUT>
__device__ inline void seq_group(const typename S::PA &a, typename S::CX *planes, long long bout, int tid, const IN (&in)[S::PIX_ROUNDS], OUT (&o)[4 * S::K_ROUNDS])
{
	typedef typename S::Re Re;
	typedef typename S::CX CX;
	constexpr int L = S::L, T = S::T, PL = S::PL, CG = S::C;
	int t = tid; asm volatile("" : "+v"(t));
	typename S::template State<KIND_REDFT10> st;
	static_for<0, S::PIX_ROUNDS>([&](auto i) { static_for<0, CG>([&](auto c) { st.pre[i * CG + c] = in[i].v[C0 + c]; }); });
	static_for<0, S::NPH - 1>([&](auto ph) {
		S::template phase<KIND_REDFT10, ph>(a, planes, bout, t, st);
		__syncthreads();
	});
	static_for<0, S::K_ROUNDS>([&](auto ri) {
		const int k = t + ri * T;
		if (!((ri + 1) * T <= L / 2 + 1 || k <= L / 2)) return;
		const int km = k ? L - k : 0;
		const CX tk = a.T[k];
		const CX tlk = cmul(cconj(tk), cmk<Re>((Re)0.70710678118654752440, (Re)-0.70710678118654752440));
		const CX t1 = csqr(csqr(tk));
		static_for<0, CG>([&](auto c) {
			const CX zk = planes[c * PL + k];
			const CX zm = cconj(planes[c * PL + km]);
			const CX E = cadd(zk, zm);
			const CX D = cmul_mi(csub(zk, zm));
			const CX P = cmul(t1, D);
			const CX wk = cmul(tk, cadd(E, P));
			const CX wm = cmul(tlk, cconj(csub(E, P)));
			const Re sc = a.scale;
			o[ri * 4 + 0].v[C0 + c] = wk.x * (k == 0 ? sc * a.out_scale0 : sc);
			o[ri * 4 + 1].v[C0 + c] = -wk.y * sc;
			o[ri * 4 + 2].v[C0 + c] = wm.x * sc;
			o[ri * 4 + 3].v[C0 + c] = -wm.y * sc;
		});
	});
	__syncthreads();
}
template <class SA, class SB, int WAVES>
__global__ void __launch_bounds__(SA::T, WAVES) row_seq21_kernel(const typename SA::PA a)
{
	typedef typename SA::Re Re;
	constexpr int N = SA::N, L = SA::L, T = SA::T, G = 3;
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	typename SA::CX *planes = reinterpret_cast<typename SA::CX *>(lds);
	const int tid = threadIdx.x;
	long long bin, bout;
	row_base(a, blockIdx.x, bin, bout);
	Pix<G, Re> in[SA::PIX_ROUNDS];
	static_for<0, SA::PIX_ROUNDS>([&](auto i) {
		const int x = tid + i * T;
		if ((i + 1) * T <= N || x < N) in[i] = load_pix<G, Re>(a.in + bin + (long long)x * G);
	});
	Pix<G, Re> o[4 * SA::K_ROUNDS];
	seq_group<SA, 0, G>(a, planes, bout, tid, in, o);
	seq_group<SB, 2, G>(a, planes, bout, tid, in, o);
	static_for<0, SA::K_ROUNDS>([&](auto ri) {
		const int k = tid + ri * T;
		if (!((ri + 1) * T <= L / 2 + 1 || k <= L / 2)) return;
		store_pix<G, Re>(a.out + bout + (long long)k * G, o[ri * 4 + 0]);
		if (k > 0) store_pix<G, Re>(a.out + bout + (long long)(N - k) * G, o[ri * 4 + 1]);
		if (L - k != k) store_pix<G, Re>(a.out + bout + (long long)(L - k) * G, o[ri * 4 + 2]);
		if (k > 0 && L + k != N - k) store_pix<G, Re>(a.out + bout + (long long)(L + k) * G, o[ri * 4 + 3]);
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

template <class K, class PA>
static float run(K k, int threads, size_t lds, const PA &a0, int wgs, int reps, int nbuf, size_t fstride)
{
	CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	auto go = [&](int i) { PA a = a0; a.in += (size_t)(i % nbuf) * fstride; a.out += (size_t)(i % nbuf) * fstride; hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), lds, 0, a); };
	for (int i = 0; i < 2 * nbuf; i++) go(i);
	CHK(hipEventRecord(e0, 0));
	for (int i = 0; i < reps; i++) go(i);
	CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
	float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
	CHK(hipGetLastError());
	return ms * 1000 / reps;
}

template <class Re, class IL, class S1>
static void bench(const char *name, int H, int nbuf)
{
	constexpr int W = IL::N, C = 3;
	const size_t NF = (size_t)H * W * C;
	Tab<Re> tb = make_tables<Re>(W, W / 2);
	std::vector<Re> h(NF);
	unsigned long long s = 12345;
	for (size_t i = 0; i < NF; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; h[i] = (Re)((double)(s >> 40) / 16777216.0 - 0.5); }
	Re *src, *o1, *o2, *clip;
	CHK(hipMalloc(&src, NF * sizeof(Re))); CHK(hipMalloc(&o1, NF * sizeof(Re))); CHK(hipMalloc(&o2, NF * sizeof(Re))); CHK(hipMalloc(&clip, NF * sizeof(Re) * nbuf));
	CHK(hipMemcpy(src, h.data(), NF * sizeof(Re), hipMemcpyHostToDevice)); CHK(hipMemset(clip, 0, NF * sizeof(Re) * nbuf));
	typename IL::PA a; memset((void *)&a, 0, sizeof a);
	a.in = src; a.out = o1; a.in_scale0 = (Re)0.5; a.out_scale0 = (Re)0.7; a.scale = (Re)(1.0 / W);
	a.kind = KIND_REDFT10;
	a.N = W; a.C = C; a.nb0 = H; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)W * C; a.sb1_in = a.sb1_out = (long long)NF;
	a.T = tb.T; a.W = tb.W;
	typename IL::PA b = a; b.out = o2;
	run(row_k<IL, 0>, IL::T, IL::LDS, a, H, 1, 1, 0);
	run(row_seq10_kernel<S1, 3, 4>, S1::T, S1::LDS, b, H, 1, 1, 0);
	std::vector<Re> r1(NF), r2(NF);
	CHK(hipMemcpy(r1.data(), o1, NF * sizeof(Re), hipMemcpyDeviceToHost)); CHK(hipMemcpy(r2.data(), o2, NF * sizeof(Re), hipMemcpyDeviceToHost));
	double md = 0, mx = 0;
	for (size_t i = 0; i < NF; i++) { md = fmax(md, fabs((double)r1[i] - (double)r2[i])); mx = fmax(mx, fabs((double)r1[i])); }
	typename IL::PA c = a; c.in = clip; c.out = clip;
	const float t_il = run(row_k<IL, 0>, IL::T, IL::LDS, c, H, 24, nbuf, NF);
	const float t_s4 = run(row_seq10_kernel<S1, 3, 4>, S1::T, S1::LDS, c, H, 24, nbuf, NF);
	const float t_s4p = run(row_seq10_kernel<S1, 3, 4>, S1::T, S1::LDS + 70000, c, H, 24, nbuf, NF);     // one workgroup per CU (extra LDS: two no longer fit 160 KB): what the second one is worth
	if constexpr (std::is_same<Re, float>::value) {
		typedef RowSpecT<Re, IL::N, 2, S1::T, 16, 15, 16> S2;
		typename IL::PA b2 = a; b2.out = o2;
		CHK(hipMemset(o2, 0, NF * sizeof(Re)));
		run(row_seq21_kernel<S2, S1, 4>, S2::T, S2::LDS, b2, H, 1, 1, 0);
		CHK(hipMemcpy(r2.data(), o2, NF * sizeof(Re), hipMemcpyDeviceToHost));
		double md2 = 0;
		for (size_t i = 0; i < NF; i++) md2 = fmax(md2, fabs((double)r1[i] - (double)r2[i]));
		const float t21 = run(row_seq21_kernel<S2, S1, 4>, S2::T, S2::LDS, c, H, 24, nbuf, NF);
		printf("%s: channels (0, 1) together then 2 (12 phases, %zu B of LDS): %.1f us, max |diff| %.3g\n", name, (size_t)S2::LDS, t21, md2);
	}
	const double gb = 2.0 * NF * sizeof(Re) / 1e9;
	printf("%s REDFT10 in place over %d frames: interleaved (three planes, %d threads) %.1f us = %.2f TB/s | channels in sequence (%d threads, one plane): two per CU %.1f us = %.2f TB/s, one per CU %.1f us | max |diff| %.3g of %.3g\n",
	       name, nbuf, IL::T, t_il, gb / t_il * 1e3, S1::T, t_s4, gb / t_s4 * 1e3, t_s4p, md, mx);
}

int main()
{
	bench<float, RowSpecT<float, 7680, 3, 1024, 16, 15, 16>, RowSpecT<float, 7680, 1, 512, 16, 15, 16>>("7680x4320x3 f32", 4320, 2);
	bench<float, RowSpecT<float, 7680, 3, 1024, 16, 15, 16>, RowSpecT<float, 7680, 1, 512, 8, 8, 4, 15>>("7680x4320x3 f32 (8.8.4.15)", 4320, 2);
	bench<double, RowSpecT<double, 3840, 3, 512, 12, 10, 16>, RowSpecT<double, 3840, 1, 512, 12, 10, 16>>("3840x2160x3 f64", 2160, 3);
	return 0;
}
