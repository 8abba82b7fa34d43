/*
 * zoom_gpu -- plain-C harness around the transform core of the reference's `zoom`: the image's REDFT10 x REDFT10 through
 * fftw(plan_many_r2r) exactly as zoom/zoom.c:263-265 calls it (include/fftw3.h: host buffer in, host buffer out), then ONE output frame of
 * the scaled-basis product of zoom.c:347-375 on the device -- by fast transforms on the DCT-III grid where the scaled lengths are integers
 * (dspfft_zoomfft_*), by chirp-z transforms for every other scale, offset and the centered basis (dspfft_zoomczt_*), by the dense product on the
 * matrix cores for axes beyond the listed convolution lengths (dspfft_zoom_basis + dspfft_zoom_product).
 * Everything the tool does around that -- option parsing, the viewport / position rules of zoom.c:268-298, the animation expressions of
 * :320-345, MagickWand / libav I/O -- is outside the hot path and not mirrored: the frame's geometry comes in as numbers.
 *
 *   zoom_gpu <input.ppm|.pf> <output.raw> <basis 0 interpolated|1 centered|2 native> <xnum> <xden> <ynum> <yden> <vx> <vy> <vw> <vh> [auto|fft|czt|gemm]
 * output.raw: "PFS\nVW VH 1\n-1.0\n" + VH x VW x 3 little-endian f32, top row first; stderr names the path taken.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>
#include <fftw3.h>
#include <dspfft.h>
#include "precision.h"
#include "rawio.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char *argv[])
{
	if (argc < 12) { fprintf(stderr, "usage: %s <input> <output.raw> <basis> <xnum> <xden> <ynum> <yden> <vx> <vy> <vw> <vh> [auto|fft|czt|gemm]\n", argv[0]); return 2; }
	const int basis = atoi(argv[3]);
	const double xnum = atof(argv[4]), xden = atof(argv[5]), ynum = atof(argv[6]), yden = atof(argv[7]), vx = atof(argv[8]), vy = atof(argv[9]);
	const size_t vw = strtoull(argv[10], NULL, 10), vh = strtoull(argv[11], NULL, 10);
	const char *method = argc > 12 ? argv[12] : "auto";
	const int any = !strcmp(method, "auto");
	if (basis < 0 || basis > 2 || !(xnum > 0 && xden > 0 && ynum > 0 && yden > 0) || !vw || !vh) { fprintf(stderr, "bad geometry\n"); return 2; }

	size_t width, height;
	coeff *coeffs;
	if (read_image_coeff(argv[1], &width, &height, &coeffs)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
	/* zoom.c:263-265 */
	fftw(plan) p = fftw(plan_many_r2r)(2, (int[]){(int)height, (int)width}, 3, coeffs, NULL, 3, 1, coeffs, NULL, 3, 1, (fftw_r2r_kind[]){FFTW_REDFT10, FFTW_REDFT10}, FFTW_ESTIMATE);
	if (!p) { fprintf(stderr, "no plan\n"); return 1; }
	fftw(execute)(p);
	fftw(destroy_plan)(p);

	const size_t npix = width * height, nout = vw * vh * 3;
	float *h32 = malloc(sizeof(float) * npix * 3), *d_coeffs = NULL, *d_out = NULL, *d_work = NULL, *frame = malloc(sizeof(float) * nout);
	for (size_t i = 0; i < npix * 3; i++) h32[i] = (float)coeffs[i];
	HIP(hipMalloc((void **)&d_coeffs, sizeof(float) * npix * 3));
	HIP(hipMalloc((void **)&d_out, sizeof(float) * nout));
	HIP(hipMemcpy(d_coeffs, h32, sizeof(float) * npix * 3, hipMemcpyHostToDevice));
	free(h32);

	const char *how = NULL;
	if (any || !strcmp(method, "fft")) {
		dspfft_zoomfft z = NULL;
		const int rc = dspfft_zoomfft_create(&z, (int)width, (int)height, basis, xnum, xden, ynum, yden, (int)vw, (int)vh);
		if (rc == 0) {
			HIP(hipMalloc((void **)&d_work, sizeof(float) * dspfft_zoomfft_work_floats(z)));
			if (dspfft_zoomfft_execute(z, d_coeffs, vx, vy, d_out, d_work, NULL)) { fprintf(stderr, "zoomfft: %s\n", dspfft_zoomfft_last_error()); return 1; }
			dspfft_zoomfft_destroy(z);
			how = "fft";
		} else if (rc != -2 || !any) { fprintf(stderr, "zoomfft: %s\n", dspfft_zoomfft_last_error()); return 1; }       /* -2: not on the DCT-III grid */
	}
	if (!how && (any || !strcmp(method, "czt"))) {
		dspfft_zoomczt z = NULL;
		const int rc = dspfft_zoomczt_create(&z, (int)width, (int)height, basis, xnum, xden, ynum, yden, (int)vw, (int)vh);
		if (rc == 0) {
			HIP(hipMalloc((void **)&d_work, sizeof(float) * dspfft_zoomczt_work_floats(z)));
			if (dspfft_zoomczt_execute(z, d_coeffs, vx, vy, d_out, d_work, NULL)) { fprintf(stderr, "zoomczt: %s\n", dspfft_zoomfft_last_error()); return 1; }
			dspfft_zoomczt_destroy(z);
			how = "czt";
		} else if (rc != -2 || !any) { fprintf(stderr, "zoomczt: %s\n", dspfft_zoomfft_last_error()); return 1; }       /* -2: beyond the listed lengths */
	}
	if (!how) {                                                        /* zoom.c:347-375 as two dense products */
		const size_t cw = dspfft_zoom_ncomponents(xnum, xden, width), ch = dspfft_zoom_ncomponents(ynum, yden, height);
		float *xb = NULL, *yb = NULL;
		HIP(hipMalloc((void **)&xb, sizeof(float) * vw * cw));
		HIP(hipMalloc((void **)&yb, sizeof(float) * vh * ch));
		HIP(hipMalloc((void **)&d_work, sizeof(float) * dspfft_zoom_work_floats((int)width, (int)height, ch, (int)vw)));
		if (dspfft_zoom_basis(xb, basis, xnum, xden, vx, vw, width, NULL) || dspfft_zoom_basis(yb, basis, ynum, yden, vy, vh, height, NULL) ||
		    dspfft_zoom_product(d_coeffs, (int)width, (int)height, xb, cw, yb, ch, d_out, (int)vw, (int)vh, d_work, NULL)) {
			fprintf(stderr, "zoom product: %s\n", dspfft_zoom_last_error()); return 1;
		}
		HIP(hipDeviceSynchronize());
		HIP(hipFree(xb)); HIP(hipFree(yb));
		how = "gemm";
	}
	HIP(hipMemcpy(frame, d_out, sizeof(float) * nout, hipMemcpyDeviceToHost));
	FILE *f = fopen(argv[2], "wb");
	if (!f) { perror(argv[2]); return 1; }
	fprintf(f, "PFS\n%zu %zu 1\n-1.0\n", vw, vh);
	if (fwrite(frame, sizeof(float), nout, f) != nout || fclose(f)) { perror("write"); return 1; }
	fprintf(stderr, "scale %g x %g at (%g, %g) by %s\n", xnum / xden, ynum / yden, vx, vy, how);
	HIP(hipFree(d_work)); HIP(hipFree(d_coeffs)); HIP(hipFree(d_out));
	free(frame); fftw(free)(coeffs); fftw(cleanup)();
	return 0;
}
