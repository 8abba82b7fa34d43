/* cbench.c -- the headline loop of bench.py from plain C through the C ABI (include/dspfft.h): 4 frames of 3840x2160x3 on two
 * streams, dspfft_execute_many_repeat.  Separates what the Python / torch environment costs from what the library costs.
 *   gcc -O2 -std=c11 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tools/cbench.c -o tools/cbench \
 *       -Ldspfun_amd/csrc -Wl,-rpath,$PWD/dspfun_amd/csrc -ldspfft_hip -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64
 *   tools/cbench [steps] [frames] [rejoin_every] [data: 0 splitmix64 uniform (bench.py's), 1 the 1000-level ramp tools/sbench.hip uses, 2 zeros]
 *                [inverse plan order: 0 last axis first (ROW, COL), 1 first axis first (COL, ROW)] [width height: default 3840 2160]
 *                [precision: 0 float, 1 double] */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <dspfft.h>
#include <hip/hip_runtime_api.h>
#define HIP(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at line %d\n", __LINE__); return 1; } } while (0)
#define DSP(x) do { if (x) { fprintf(stderr, "dspfft: %s (line %d)\n", dspfft_last_error(), __LINE__); return 1; } } while (0)
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
	const int steps = argc > 1 ? atoi(argv[1]) : 300, nfr = argc > 2 ? atoi(argv[2]) : 4, rejoin = argc > 3 ? atoi(argv[3]) : 8, data = argc > 4 ? atoi(argv[4]) : 0, order = argc > 5 ? atoi(argv[5]) : 0;
	const int Wc = argc > 6 ? atoi(argv[6]) : 3840, Hc = argc > 7 ? atoi(argv[7]) : 2160;
	const int f64 = argc > 8 ? atoi(argv[8]) : 0;
	const int H = Hc, W = Wc, C = 3;
	const size_t NF = (size_t)H * W * C * (f64 ? 2 : 1);            /* in floats: a double frame is two float frames long */
	float *buf, *h = malloc(NF * 4);
	unsigned long long s = 0xD5F0002ull;
	for (size_t i = 0; i < NF; i++) { s += 0x9E3779B97F4A7C15ull; unsigned long long z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; if (f64) { if (i < NF / 2) ((double *)h)[i] = data == 0 ? (double)(z >> 11) * (1.0 / 9007199254740992.0) : data == 1 ? (double)((i * 2654435761u) % 1000) / 1000. : 0.; continue; } h[i] = data == 0 ? (float)(z >> 40) * (1.0f / 16777216.0f) : data == 1 ? (float)((i * 2654435761u) % 1000) / 1000.f : 0.f; }
	HIP(hipMalloc((void **)&buf, NF * 4 * nfr));
	for (int f = 0; f < nfr; f++) HIP(hipMemcpy(buf + f * NF, h, NF * 4, hipMemcpyHostToDevice));
	dspfft_plan fwd, inv;
	const int dims[2] = {H, W}, k10[2] = {DSPFFT_REDFT10, DSPFFT_REDFT10}, k01[2] = {DSPFFT_REDFT01, DSPFFT_REDFT01};
	if (f64) {
		DSP(dspfft_plan_many_r2r_f64(&fwd, 2, dims, C, NULL, C, 1, NULL, C, 1, k10));
		DSP(dspfft_plan_many_r2r_f64(&inv, 2, dims, C, NULL, C, 1, NULL, C, 1, k01));
		DSP(dspfft_plan_set_scale_f64(inv, 1.0 / (4.0 * W * H)));
	} else {
		DSP(dspfft_plan_many_r2r(&fwd, 2, dims, C, NULL, C, 1, NULL, C, 1, k10));
		DSP(dspfft_plan_many_r2r_ordered(&inv, 2, dims, C, NULL, C, 1, NULL, C, 1, k01, order));
		DSP(dspfft_plan_set_scale(inv, 1.0f / (4.0f * W * H)));
	}
	hipStream_t st[2];
	HIP(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking)); HIP(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
	dspfft_plan plans[64]; const void *in[64]; void *out[64]; void *streams[64];
	for (int f = 0; f < nfr; f++) for (int k = 0; k < 2; k++) { const int i = 2 * f + k; plans[i] = k ? inv : fwd; in[i] = out[i] = buf + f * NF; streams[i] = st[f & 1]; }
	{	/* each pass alone on frame 0 (cache-resident), as tools/sbench.hip prints them */
		hipEvent_t a, b; HIP(hipEventCreate(&a)); HIP(hipEventCreate(&b));
		for (int k = 0; k < 2 && !f64; k++) for (int p = 0; p < dspfft_plan_num_passes(k ? inv : fwd); p++) {
			for (int i = 0; i < 3; i++) DSP(dspfft_execute_pass(k ? inv : fwd, p, buf, buf, st[0]));
			HIP(hipEventRecord(a, st[0]));
			for (int i = 0; i < 30; i++) DSP(dspfft_execute_pass(k ? inv : fwd, p, buf, buf, st[0]));
			HIP(hipEventRecord(b, st[0])); HIP(hipEventSynchronize(b));
			float ms; HIP(hipEventElapsedTime(&ms, a, b));
			printf("%s pass %d alone: %.1f us\n", k ? "inverse" : "forward", p, ms * 1000 / 30);
		}
		HIP(hipMemcpy(buf, h, NF * 4, hipMemcpyHostToDevice));
	}
	for (int round = 0; round < 3; round++) {
		DSP(dspfft_execute_many_repeat(2 * nfr, plans, in, out, streams, 60, rejoin, 0, 0, NULL));
		HIP(hipDeviceSynchronize());
		const double t0 = now();
		DSP(dspfft_execute_many_repeat(2 * nfr, plans, in, out, streams, steps, rejoin, 0, 0, NULL));
		const double t1 = now();
		HIP(hipDeviceSynchronize());
		const double t2 = now();
		printf("data %d order %d round %d: %d steps of %d frames, rejoin %d: %.0f Mpix/s (%.4f ms/step, host enqueue %.4f ms/step)\n", data, order, round, steps, nfr, rejoin,
		       (double)steps * nfr * H * W / 1e6 / (t2 - t0), (t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3);
	}
	HIP(hipMemcpy(h, buf, NF * 4, hipMemcpyDeviceToHost));
	return 0;
}
