// r05_probe.hip -- two questions behind round 5's designs, answered with no transform arithmetic in the way.
//
//   hipcc -O3 --offload-arch=gfx950 tools/r05_probe.hip -o tools/r05_probe && tools/r05_probe
//
// (1) ROW FOLD (a 7680 x 3 float line as two 3840-sample transforms through one 46 KB plane): may the first transform store its outputs
//     as every other pixel (12 of every 24 bytes) and the second fill the gaps a few microseconds later, or must whole pixel pairs be
//     stored at once?  And may the inverse direction load every other pixel twice instead of whole lines once?
//       mode 0  whole-line loads (pixel n and N-1-n), work, work, pixel PAIRS stored once              [the ideal]
//       mode 1  whole-line loads, work, even pixels stored, work, odd pixels stored                     [partial cache lines meet in L2?]
//       mode 2  every-other-pixel loads, work, every-other-pixel loads, work, pixel pairs stored        [partial loads]
//       mode 3  mode 1's stores with mode 2's loads
//     512 threads, 46 KB of LDS claimed (three workgroups per CU), `work` = a chain of dependent FMAs per thread standing in for a transform.
// (2) HOST LINK: what rate does a pinned host buffer move at -- hipMemcpyAsync on one stream, split over 2 / 4 streams, and a KERNEL that
//     reads (writes) the pinned buffer directly (what a first (last) pass fused with the upload (download) would do); both directions at once.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <functional>
#include <time.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f3 __attribute__((ext_vector_type(3)));
static __device__ inline f3 ld3(const float *p) { f3 t; __builtin_memcpy(&t, p, 12); return t; }
static __device__ inline void st3(float *p, f3 t) { __builtin_memcpy(p, &t, 12); }

template <int WORK> static __device__ inline float burn(float v, float a)
{
#pragma unroll 16
	for (int i = 0; i < WORK; i++) v = __builtin_fmaf(v, a, 1e-7f);
	return v;
}

constexpr int N = 7680, H = N / 2;

// T threads, WPE waves per SIMD asked of the register allocator; RND = rounds of pixel n (and its mirror) per thread (7.5 at 512 threads)
template <int MODE, int WORK, int T = 512, int WPE = 6>
__global__ void __launch_bounds__(T, WPE) fold_probe(const float *in, float *out, float a)
{
	constexpr int RND = (H + T - 1) / T;
	extern __shared__ float lds[];
	const int tid = threadIdx.x;
	const long long base = (long long)blockIdx.x * N * 3;
	f3 p[RND], q[RND];
	if (MODE == 0 || MODE == 1) {
		// pixel n ascending, pixel N-1-n descending: every wave instruction 768 contiguous bytes
		for (int i = 0; i < RND; i++) { const int n = tid + i * T; if (n < H) { p[i] = ld3(in + base + 3ll * n); q[i] = ld3(in + base + 3ll * (N - 1 - n)); } }
	} else {
		// even pixels only (the even coefficients of a REDFT01 line): 12 of every 24 bytes
		for (int i = 0; i < RND; i++) { const int n = tid + i * T; if (n < H) p[i] = ld3(in + base + 6ll * n); }
	}
	float acc = 0;
	for (int i = 0; i < RND; i++) { const int n = tid + i * T; if (n < H) acc += p[i].x + p[i].y + p[i].z; }
	acc = burn<WORK>(acc, a);
	lds[tid] = acc; __syncthreads(); acc += lds[(tid + 64) % T];
	if (MODE == 1 || MODE == 3) {
		for (int i = 0; i < RND; i++) { const int n = tid + i * T; if (n < H) { f3 v = p[i]; v.x += acc; st3(out + base + 6ll * n, v); } }
	}
	if (MODE == 2 || MODE == 3) {
		for (int i = 0; i < RND; i++) { const int n = tid + i * T; if (n < H) q[i] = ld3(in + base + 6ll * n + 3); }
	}
	for (int i = 0; i < RND; i++) { const int n = tid + i * T; if (n < H) acc += q[i].x + q[i].y + q[i].z; }
	acc = burn<WORK>(acc, a);
	__syncthreads(); lds[tid] = acc; __syncthreads(); acc += lds[(tid + 64) % T];
	if (MODE == 1 || MODE == 3) {
		for (int i = 0; i < RND; i++) { const int n = tid + i * T; if (n < H) { f3 v = q[i]; v.x += acc; st3(out + base + 6ll * n + 3, v); } }
	} else {
		for (int i = 0; i < RND; i++) {
			const int n = tid + i * T;
			if (n < H) { f3 v = p[i], w = q[i]; v.x += acc; w.y += acc; st3(out + base + 6ll * n, v); st3(out + base + 6ll * n + 3, w); }
		}
	}
}

// The row PAIR butterfly of the split column pass (rows y1 = 2p, y2 = N-1-2p -> r1 + r2, r1 - r2) with ONE output line per workgroup: partners
// b and b + 8 (same XCD, dispatched back to back) both read the two lines and keep the sum / the difference.  SHARE = 0: each reads only its own
// line (the floor); 1: both read both.
template <int WORK, int SHARE, int T = 512, int WPE = 4>
__global__ void __launch_bounds__(T, WPE) pair_probe(const float *in, float *out, float a)
{
	constexpr int RND = (H + T - 1) / T;
	extern __shared__ float lds[];
	const int tid = threadIdx.x, b = blockIdx.x;
	const int p = (b >> 4) * 8 + (b & 7), h = (b >> 3) & 1;
	const long long l1 = (long long)(2 * p) * N * 3, l2 = (long long)(2 * p + 1) * N * 3, mine = h ? l2 : l1;
	const float sg = h ? -1.f : 1.f;
	f3 u[RND], v[RND];
	for (int i = 0; i < RND; i++) {
		const int n = tid + i * T;
		if (n < H) {
			if (SHARE) {
				const f3 a1 = ld3(in + l1 + 3ll * n), a2 = ld3(in + l2 + 3ll * n), b1 = ld3(in + l1 + 3ll * (N - 1 - n)), b2 = ld3(in + l2 + 3ll * (N - 1 - n));
				u[i] = a1 + sg * a2; v[i] = b1 + sg * b2;
			} else { u[i] = ld3(in + mine + 3ll * n); v[i] = ld3(in + mine + 3ll * (N - 1 - n)); }
		}
	}
	float acc = 0;
	for (int i = 0; i < RND; i++) { const int n = tid + i * T; if (n < H) acc += u[i].x + u[i].y + u[i].z; }
	acc = burn<WORK>(acc, a);
	lds[tid] = acc; __syncthreads(); acc += lds[(tid + 64) % T];
	for (int i = 0; i < RND; i++) { const int n = tid + i * T; if (n < H) acc += v[i].x + v[i].y + v[i].z; }
	acc = burn<WORK>(acc, a);
	__syncthreads(); lds[tid] = acc; __syncthreads(); acc += lds[(tid + 64) % T];
	for (int i = 0; i < RND; i++) {
		const int n = tid + i * T;
		if (n < H) { f3 x = u[i], w = v[i]; x.x += acc; w.y += acc; st3(out + mine + 6ll * n, x); st3(out + mine + 6ll * n + 3, w); }
	}
}
template <int WORK, int SHARE, int T = 512, int WPE = 4>
static void run_pair(const char *what, float *in, float *out, int lines, hipStream_t st, size_t lds)
{
	CK(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_probe<WORK, SHARE, T, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int i = 0; i < 3; i++) hipLaunchKernelGGL((pair_probe<WORK, SHARE, T, WPE>), dim3(lines), dim3(T), lds, st, in, out, 0.999f);
	CK(hipStreamSynchronize(st));
	const int R = 20;
	CK(hipEventRecord(e0, st));
	for (int i = 0; i < R; i++) hipLaunchKernelGGL((pair_probe<WORK, SHARE, T, WPE>), dim3(lines), dim3(T), lds, st, in, out, 0.999f);
	CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	const double us = ms * 1e3 / R, bytes = 2.0 * lines * N * 3 * 4;
	printf("pair share %d work %4d T %4d lds %3zu KB %-58s %8.1f us  %6.2f TB/s (of read + write once)\n", SHARE, WORK, T, lds >> 10, what, us, bytes / us / 1e6);
}

template <int MODE, int WORK, int T = 512, int WPE = 6>
static void run_fold(const char *what, float *in, float *out, int lines, hipStream_t st, size_t lds = 46 * 1024)
{
	CK(hipFuncSetAttribute(reinterpret_cast<const void *>(fold_probe<MODE, WORK, T, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int i = 0; i < 3; i++) hipLaunchKernelGGL((fold_probe<MODE, WORK, T, WPE>), dim3(lines), dim3(T), lds, st, in, out, 0.999f);
	CK(hipStreamSynchronize(st));
	const int R = 20;
	CK(hipEventRecord(e0, st));
	for (int i = 0; i < R; i++) hipLaunchKernelGGL((fold_probe<MODE, WORK, T, WPE>), dim3(lines), dim3(T), lds, st, in, out, 0.999f);
	CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	const double us = ms * 1e3 / R, bytes = 2.0 * lines * N * 3 * 4;
	printf("fold mode %d work %4d T %4d lds %3zu KB %-58s %8.1f us  %6.2f TB/s (read + write once)\n", MODE, WORK, T, lds >> 10, what, us, bytes / us / 1e6);
}

// ---- host link ----
__global__ void __launch_bounds__(256) copy_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static double time_it(int reps, hipStream_t *sts, int ns, const std::function<void()> &f)
{
	f();
	for (int s = 0; s < ns; s++) CK(hipStreamSynchronize(sts[s]));
	timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
	for (int r = 0; r < reps; r++) { f(); for (int s = 0; s < ns; s++) CK(hipStreamSynchronize(sts[s])); }
	clock_gettime(CLOCK_MONOTONIC, &t1);
	return ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / reps;
}

int main(int argc, char **argv)
{
	const bool do_fold = argc < 2 || strstr(argv[1], "fold"), do_link = argc < 2 || strstr(argv[1], "link");
	hipStream_t st[4];
	for (auto &s : st) CK(hipStreamCreate(&s));
	if (do_fold) {
		const int lines = 4320;
		const size_t n = (size_t)lines * N * 3;
		float *in, *out;
		CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
		CK(hipMemset(in, 0, n * 4)); CK(hipMemset(out, 0, n * 4));
		printf("# 7680 x 4320 x 3 floats (398 MB), out of place; then in place\n");
		run_fold<0, 0>("whole loads, pairs stored once, no work", in, out, lines, st[0]);
		run_fold<0, 650>("whole loads, pairs stored once", in, out, lines, st[0]);
		run_fold<1, 650>("whole loads, even pixels stored, work, odd pixels stored", in, out, lines, st[0]);
		run_fold<2, 650>("every-other-pixel loads twice, pairs stored once", in, out, lines, st[0]);
		run_fold<3, 650>("every-other-pixel loads and stores", in, out, lines, st[0]);
		run_fold<1, 1300>("as mode 1, twice the work between the partial stores", in, out, lines, st[0]);
		run_fold<0, 1300>("as mode 0, twice the work", in, out, lines, st[0]);
		printf("# workgroups per CU (mode 0, in place): three of 512 threads (46 KB), two of 512 (70 KB claimed), two of 768, one of 1024 (92 KB: today's 8K row kernel)\n");
		run_fold<0, 650, 512, 6>("3 x 512", in, in, lines, st[0]);
		run_fold<0, 1000, 512, 6>("3 x 512", in, in, lines, st[0]);
		run_fold<0, 1300, 512, 6>("3 x 512", in, in, lines, st[0]);
		run_fold<0, 650, 512, 4>("2 x 512", in, in, lines, st[0], 70 * 1024);
		run_fold<0, 1000, 512, 4>("2 x 512", in, in, lines, st[0], 70 * 1024);
		run_fold<0, 1300, 512, 4>("2 x 512", in, in, lines, st[0], 70 * 1024);
		run_fold<0, 433, 768, 6>("2 x 768 (the same work per line: 650 x 512 / 768)", in, in, lines, st[0], 70 * 1024);
		run_fold<0, 667, 768, 6>("2 x 768 (1000 x 512 / 768)", in, in, lines, st[0], 70 * 1024);
		run_fold<0, 867, 768, 6>("2 x 768 (1300 x 512 / 768)", in, in, lines, st[0], 70 * 1024);
		run_fold<0, 325, 1024, 4>("1 x 1024 (650 x 512 / 1024)", in, in, lines, st[0], 92 * 1024);
		run_fold<0, 500, 1024, 4>("1 x 1024 (1000 x 512 / 1024)", in, in, lines, st[0], 92 * 1024);
		run_fold<0, 650, 1024, 4>("1 x 1024 (1300 x 512 / 1024)", in, in, lines, st[0], 92 * 1024);
		printf("# row pairs, one output line per workgroup, partners on one XCD (out of place: the partner still reads the line)\n");
		run_pair<650, 0>("own line only, 2 x 512", in, out, lines, st[0], 70 * 1024);
		run_pair<650, 1>("both lines read by both partners, 2 x 512", in, out, lines, st[0], 70 * 1024);
		run_pair<1000, 0>("own line only, 2 x 512", in, out, lines, st[0], 70 * 1024);
		run_pair<1000, 1>("both lines read by both partners, 2 x 512", in, out, lines, st[0], 70 * 1024);
		run_pair<1000, 1, 512, 6>("both lines read by both partners, 3 x 512", in, out, lines, st[0], 46 * 1024);
		printf("# in place\n");
		run_fold<0, 650>("whole loads, pairs stored once", in, in, lines, st[0]);
		run_fold<1, 650>("whole loads, even pixels stored, work, odd pixels stored", in, in, lines, st[0]);
		run_fold<2, 650>("every-other-pixel loads twice, pairs stored once", in, in, lines, st[0]);
		CK(hipFree(in)); CK(hipFree(out));
	}
	if (do_link) {
		const size_t bytes = (size_t)3840 * 2160 * 3 * 4;     // one 4K RGB float frame, 99.5 MB
		void *h0, *h1, *d0, *d1;
		CK(hipHostMalloc(&h0, bytes, hipHostMallocDefault)); CK(hipHostMalloc(&h1, bytes, hipHostMallocDefault));
		CK(hipMalloc(&d0, bytes)); CK(hipMalloc(&d1, bytes));
		memset(h0, 1, bytes); memset(h1, 2, bytes);
		const int R = 10;
		auto report = [&](const char *what, double s, double moved) { printf("link %-70s %7.3f ms  %6.1f GB/s\n", what, s * 1e3, moved / s / 1e9); };
		for (int ns : {1, 2, 4}) {
			char buf[128];
			const size_t part = bytes / ns;
			double s = time_it(R, st, ns, [&]() { for (int i = 0; i < ns; i++) CK(hipMemcpyAsync((char *)d0 + i * part, (char *)h0 + i * part, part, hipMemcpyHostToDevice, st[i])); });
			snprintf(buf, sizeof buf, "hipMemcpyAsync H2D, %d stream(s)", ns); report(buf, s, bytes);
			s = time_it(R, st, ns, [&]() { for (int i = 0; i < ns; i++) CK(hipMemcpyAsync((char *)h1 + i * part, (char *)d0 + i * part, part, hipMemcpyDeviceToHost, st[i])); });
			snprintf(buf, sizeof buf, "hipMemcpyAsync D2H, %d stream(s)", ns); report(buf, s, bytes);
		}
		double s = time_it(R, st, 2, [&]() {
			CK(hipMemcpyAsync(d0, h0, bytes, hipMemcpyHostToDevice, st[0]));
			CK(hipMemcpyAsync(h1, d1, bytes, hipMemcpyDeviceToHost, st[1]));
		});
		report("hipMemcpyAsync H2D and D2H at once (two streams), bytes each way", s, bytes);
		void *dh0, *dh1;
		CK(hipHostGetDevicePointer(&dh0, h0, 0)); CK(hipHostGetDevicePointer(&dh1, h1, 0));
		for (int blocks : {64, 256, 1024, 4096}) {
			char buf[128];
			s = time_it(R, st, 1, [&]() { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, st[0], (const float4 *)dh0, (float4 *)d0, bytes / 16); });
			snprintf(buf, sizeof buf, "kernel reads pinned host -> device, %d x 256 threads", blocks); report(buf, s, bytes);
			s = time_it(R, st, 1, [&]() { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, st[0], (const float4 *)d0, (float4 *)dh1, bytes / 16); });
			snprintf(buf, sizeof buf, "kernel writes device -> pinned host, %d x 256 threads", blocks); report(buf, s, bytes);
		}
		s = time_it(R, st, 2, [&]() {
			hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, st[0], (const float4 *)dh0, (float4 *)d0, bytes / 16);
			hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, st[1], (const float4 *)d1, (float4 *)dh1, bytes / 16);
		});
		report("kernel read + kernel write at once (two streams), bytes each way", s, bytes);
		// a pipeline in slabs: upload slab i+1 while slab i is "processed" (device copy) and slab i-1 goes down
		for (int slabs : {2, 4, 8}) {
			char buf[128];
			const size_t part = bytes / slabs;
			hipEvent_t up[8], done[8];
			for (int i = 0; i < slabs; i++) { CK(hipEventCreateWithFlags(&up[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming)); }
			s = time_it(R, st, 3, [&]() {
				for (int i = 0; i < slabs; i++) {
					CK(hipMemcpyAsync((char *)d0 + i * part, (char *)h0 + i * part, part, hipMemcpyHostToDevice, st[0])); CK(hipEventRecord(up[i], st[0]));
					CK(hipStreamWaitEvent(st[1], up[i], 0));
					hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, st[1], (const float4 *)((char *)d0 + i * part), (float4 *)((char *)d1 + i * part), part / 16);
					CK(hipEventRecord(done[i], st[1]));
					CK(hipStreamWaitEvent(st[2], done[i], 0));
					CK(hipMemcpyAsync((char *)h1 + i * part, (char *)d1 + i * part, part, hipMemcpyDeviceToHost, st[2]));
				}
			});
			snprintf(buf, sizeof buf, "up -> device copy -> down pipelined in %d slabs on three streams (whole roundtrip)", slabs); report(buf, s, bytes);
		}
		CK(hipHostFree(h0)); CK(hipHostFree(h1)); CK(hipFree(d0)); CK(hipFree(d1));
	}
	return 0;
}
