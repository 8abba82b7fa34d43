// colbench.hip -- experiment harness (not part of the product): plain column passes (spec_kernels.h col_spec_kernel) for candidate
// radix orders / thread counts of one column length, on a stack of planar frames (HBM-resident, ~530 MB).
// A stage of radix R has (N / R) * (K / 4) butterflies; more of them than threads means a second, mostly idle round.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -ffp-contract=on -std=c++17 -Idspfun_amd/csrc tools/colbench.hip -o tools/colbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "spec_kernels.h"
using namespace dspfft;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <class S>
static void bench(const char *label, int W)
{
	const int H = S::N;
	const int FR = (int)(530e6 / 4 / H / W);
	float *buf; CHK(hipMalloc(&buf, (size_t)H * W * FR * 4));
	CHK(hipMemset(buf, 0, (size_t)H * W * FR * 4));
	std::vector<cf> T(H + 1), Wv(H);
	for (int j = 0; j <= H; j++) T[j] = cmk((float)cos(M_PI * j / (2.0 * H)), (float)-sin(M_PI * j / (2.0 * H)));
	for (int t = 0; t < H; t++) Wv[t] = cmk((float)cos(2 * M_PI * t / H), (float)-sin(2 * M_PI * t / H));
	cf *dT, *dW; CHK(hipMalloc(&dT, T.size() * 8)); CHK(hipMalloc(&dW, Wv.size() * 8));
	CHK(hipMemcpy(dT, T.data(), T.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dW, Wv.data(), Wv.size() * 8, hipMemcpyHostToDevice));
	PassArgs a; memset((void *)&a, 0, sizeof a);
	a.in = buf; a.out = buf; a.N = H; a.T = dT; a.W = dW; a.in_scale0 = a.out_scale0 = 1.f; a.scale = 1.f / (2.f * H);
	a.K = S::K; a.B = S::B; a.ninner = W; a.ntiles = W / S::K; a.es_in = a.es_out = W; a.nb0 = FR; a.nb1 = 1; a.sb0_in = a.sb0_out = (long long)H * W;
	const int nwork = a.ntiles * FR;
	float t[2];
	for (int kind = 0; kind < 2; kind++) {
		a.kind = kind;
		hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
		float best = 1e9;
		for (int rep = 0; rep < 4; rep++) {
			CHK(hipEventRecord(e0));
			if (kind == 0) { if (launch_col_spec<S, 0>(a, nwork, nullptr)) exit(1); } else { if (launch_col_spec<S, 1>(a, nwork, nullptr)) exit(1); }
			CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
			float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
		}
		t[kind] = best;
	}
	const double bytes = 2.0 * H * W * FR * 4;
	printf("N=%d K=%d T=%d %-18s lds=%6zu | REDFT10 %7.1f us %5.2f TB/s | REDFT01 %7.1f us %5.2f TB/s\n", S::N, S::K, S::T, label, S::LDS, t[0] * 1e3, bytes / t[0] / 1e9, t[1] * 1e3, bytes / t[1] / 1e9);
	CHK(hipFree(buf)); CHK(hipFree(dT)); CHK(hipFree(dW));
}
#define B(W, N, K, T, ...) bench<ColSpec<N, K, T, __VA_ARGS__>>(#__VA_ARGS__, W)
int main(int argc, char **argv)
{
	const int which = argc > 1 ? atoi(argv[1]) : 0;
	if (which == 0 || which == 1080) {
		B(1920, 1080, 16, 512, 8, 9, 15); B(1920, 1080, 16, 512, 10, 12, 9); B(1920, 1080, 16, 512, 12, 10, 9); B(1920, 1080, 16, 512, 9, 8, 15); B(1920, 1080, 16, 512, 12, 6, 15);
	}
	if (which == 0 || which == 540) {
		B(960, 540, 16, 256, 4, 9, 15); B(960, 540, 16, 256, 6, 10, 9); B(960, 540, 16, 256, 10, 6, 9); B(960, 540, 16, 512, 6, 10, 9); B(960, 540, 16, 512, 4, 9, 15); B(960, 540, 16, 256, 12, 5, 9);
	}
	if (which == 0 || which == 720) {
		B(1280, 720, 16, 256, 6, 8, 15); B(1280, 720, 16, 512, 6, 8, 15); B(1280, 720, 16, 512, 8, 10, 9); B(1280, 720, 16, 512, 10, 8, 9); B(1280, 720, 16, 256, 8, 10, 9);
	}
	if (which == 0 || which == 1440) {
		B(2560, 1440, 8, 512, 8, 12, 15); B(2560, 1440, 8, 512, 12, 8, 15); B(2560, 1440, 8, 512, 10, 16, 9); B(2560, 1440, 8, 512, 16, 10, 9); B(2560, 1440, 8, 512, 12, 10, 12);
	}
	if (which == 0 || which == 2160) {
		B(3840, 2160, 8, 512, 12, 12, 15); B(3840, 2160, 8, 512, 9, 16, 15); B(3840, 2160, 8, 512, 16, 9, 15); B(3840, 2160, 8, 512, 16, 15, 9); B(3840, 2160, 8, 512, 15, 16, 9);
	}
	return 0;
}
