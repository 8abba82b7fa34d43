/*
 * scan_gpu -- plain-C harness reproducing the arithmetic of the reference's `scan` main loop
 * (scan/scan.c:289-298 forward transform and normalisation, :352-359 buffers and the out-of-place
 * FFTW_MEASURE inverse plan, :377-383 DC broadcast, :421-459 per-frame scatter / inverse / accumulate)
 * over include/fftw3.h, with the zigzag order (scan/scan_methods.c:77-115) from the engine's
 * dspfft_scan_zigzag.  Output: the final `sum` image (== input when every index was scanned) and,
 * per frame, max|sum - input| on stderr (the quantity --measure-parity thresholds, scan.c:508-526).
 *
 *   scan_gpu in.{ppm,pf} out.pf [step] [method-prefix]      (default method: zigzag, from the device;
 *                                                            other methods from host/scan_orders.c)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include <fftw3.h>
#include <dspfft.h>
#include <hip/hip_runtime_api.h>
#include "precision.h"
#include "rawio.h"
#include "scan_orders.h"

int main(int argc, char *argv[])
{
	if (argc < 3) { fprintf(stderr, "usage: %s <in> <out.pf> [step]\n", argv[0]); return 2; }
	size_t width, height, channels = 3;
	float *pix;
	if (read_image(argv[1], &width, &height, &pix)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
	const int method = argc > 4 ? scan_order_find_prefix(argv[4]) : SCAN_ZIGZAG;    /* scan.c:176 scan_method_find_prefix */
	if (method < 0) { fprintf(stderr, "unknown scan method %s\n", argv[4]); return 2; }
	size_t limit = scan_order_limit(method, width, height);                             /* scan_context.c:30 */
	size_t max_interval = scan_order_max_interval(method, width, height);
	size_t step = argc > 3 ? strtoul(argv[3], NULL, 10) : (limit + 31) / 32;
	if (!step) step = 1;

	coeff *coeffs = fftw(alloc_real)(width * height * channels);    /* scan.c:275 */
	memcpy(coeffs, pix, sizeof(coeff) * width * height * channels);

	fftw(init_threads)();                                           /* scan.c:289-290 */
	fftw(plan_with_nthreads)(1);
	fftw(plan) forward = fftw(plan_many_r2r)(2, (int[2]){height, width}, channels, coeffs, NULL, channels, 1, coeffs, NULL, channels, 1,
	                                         (fftw(r2r_kind)[2]){FFTW_REDFT10, FFTW_REDFT10}, FFTW_ESTIMATE);
	fftw(execute)(forward);
	fftw(destroy_plan)(forward);
	for (size_t i = 0; i < width * height * channels; i++) coeffs[i] /= width * height * 4;   /* scan.c:296-298 */

	/* zigzag: the engine's device generator, copied back once; other methods: host/scan_orders.c per index */
	uint32_t *order = NULL, *d_order;
	if (method == SCAN_ZIGZAG) {
		order = malloc(sizeof(uint32_t) * limit);
		if (hipMalloc((void **)&d_order, sizeof(uint32_t) * limit) != hipSuccess ||
		    dspfft_scan_zigzag(d_order, width, height, 0, limit, NULL) ||
		    hipMemcpy(order, d_order, sizeof(uint32_t) * limit, hipMemcpyDeviceToHost) != hipSuccess) {
			fprintf(stderr, "zigzag failed: %s\n", dspfft_last_error());
			return 1;
		}
		hipFree(d_order);
	}
	size_t (*coords)[2] = malloc(sizeof(*coords) * (max_interval + 1) * step);         /* scan.c:346 */

	size_t nframes = (limit + step - 1) / step;                      /* scan.c:347-348 */
	coeff *reconstruction = fftw(alloc_real)(width * height * channels);   /* scan.c:352-354 */
	coeff *image = fftw(alloc_real)(width * height * channels);
	memset(reconstruction, 0, sizeof(*reconstruction) * width * height * channels);
	fftw(plan) inverse = fftw(plan_many_r2r)(2, (int[2]){height, width}, channels, reconstruction, NULL, channels, 1, image, NULL, channels, 1,
	                                         (fftw(r2r_kind)[2]){FFTW_REDFT01, FFTW_REDFT01}, FFTW_MEASURE);   /* scan.c:359 */
	coeff *sum = calloc(width * height * channels, sizeof(*sum));
	for (size_t i = 0; i < width * height; i++) memcpy(sum + i * channels, coeffs, sizeof(*sum) * channels);   /* scan.c:382-383 */

	double exec_s = 0;
	for (size_t i = 0; i < nframes; i++) {                           /* scan.c:421-459 */
		memset(reconstruction, 0, sizeof(*reconstruction) * width * height * channels);
		size_t ncoords = 0;
		for (size_t s = i * step; s < i * step + step && s < limit; s++) {               /* scan.c:423-427 */
			if (order) { coords[ncoords][0] = order[s] / width; coords[ncoords][1] = order[s] % width; ncoords++; }
			else ncoords += scan_order_coords(method, width, height, s, coords + ncoords);
		}
		for (size_t ci = 0; ci < ncoords; ci++) {                                        /* scan.c:430-432 */
			size_t y = coords[ci][0], x = coords[ci][1];
			size_t lin = y * width + x;                   /* `box` emits x = i >= width on tall images: the reference's pointer arithmetic
			                                                 lands in a later row; past the end of the buffer it is undefined, skipped here */
			if (lin >= width * height) continue;
			memcpy(reconstruction + lin * channels, coeffs + lin * channels, sizeof(*reconstruction) * channels);
		}
		memset(reconstruction, 0, sizeof(*coeffs) * channels);       /* clear DC, scan.c:445 */
		struct timespec t0, t1;
		clock_gettime(CLOCK_MONOTONIC, &t0);
		fftw(execute)(inverse);
		clock_gettime(CLOCK_MONOTONIC, &t1);
		exec_s += (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
		double err = 0;
		for (size_t j = 0; j < width * height * channels; j++) {
			sum[j] += image[j];
			double e = fabs((double)sum[j] - pix[j]);
			if (e > err) err = e;
		}
		fprintf(stderr, "frame %zu/%zu max|sum-input| = %.3e\n", i + 1, nframes, err);
	}
	/* what the host-pointer boundary costs per output frame (scan.c:447): upload of `reconstruction`, two axis passes, download of `image` */
	fprintf(stderr, "fftw(execute) of the %zux%zu inverse plan: %.3f ms per frame over %zu frames (%.1f GB/s over both transfers)\n", width, height,
	        exec_s / nframes * 1e3, nframes, 2.0 * sizeof(coeff) * width * height * channels / (exec_s / nframes) / 1e9);
	int rc = write_pf(argv[2], width, height, sum);
	fftw(destroy_plan)(inverse);
	fftw(free)(reconstruction); fftw(free)(image); fftw(free)(coeffs);
	free(sum); free(order); free(coords); free(pix);
	fftw(cleanup)(); fftw(cleanup_threads)();                        /* scan.c:563-564 */
	return rc;
}
