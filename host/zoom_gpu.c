/*
 * zoom_gpu -- plain-C harness reproducing the core of the reference's `zoom` (zoom/zoom.c:170-375): the image's REDFT10 x REDFT10
 * through fftw(plan_many_r2r) exactly as zoom.c:263-265 calls it (include/fftw3.h: host buffer in, host buffer out), then per output
 * frame the scaled-basis product of zoom.c:347-375 on the device -- by fast transforms on the DCT-III grid where the scaled lengths are
 * integers (dspfft_zoomfft_*), by chirp-z transforms for every other scale, offset and the centered basis (dspfft_zoomczt_*), by the
 * dense product on the matrix cores for axes beyond the listed convolution lengths (dspfft_zoom_basis + dspfft_zoom_product).
 * MagickWand / libav I/O (zoom.c:230-262,300-312,395-410) is replaced by host/rawio.h: P6 / PF in, raw float frames out; -g (linear
 * RGB through libavutil's transfer functions) and --showsamples (a rendering aid drawn over the frame, zoom.c:377-393) are not mirrored.
 *
 *   zoom_gpu [-s <scale>|-r <WxH>] [-p <XxY>] [-v <WxH>] [-c] [-P] [-%] [--basis interpolated|native|centered] [-n frames]
 *            [-x expr] [-y expr] [-S expr] [-X expr] [-Y expr] [--method auto|fft|czt|gemm] <input.ppm|.pf> <output.raw>
 * output.raw: "PFS\nVW VH FRAMES\n-1.0\n" + FRAMES x VH x VW x 3 little-endian f32, top row first.
 * The animation expressions are libavutil's language as host/expr_eval.c restates it (variables i n x y xs ys w h vw vh, zoom.c:219).
 */
#include <getopt.h>
#include <math.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>
#include <fftw3.h>
#include <dspfft.h>
#include "precision.h"
#include "rawio.h"
#include "expr_eval.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
enum { INTERPOLATED, CENTERED, NATIVE };                               /* zoom.c:20-26 */
enum { M_AUTO, M_FFT, M_CZT, M_GEMM };

int main(int argc, char *argv[])
{
	double vx = 0, vy = 0;                                             /* zoom.c:118-128 */
	size_t vw = 0, vh = 0;
	bool centered = false, input_coords = false, pct_coords = false;
	double xscale_num = 1, yscale_num = 1, logical_width = 0, logical_height = 0;
	unsigned long long xscale_den = 1, yscale_den = 1;
	int scaling_type = INTERPOLATED, method = M_AUTO;
	size_t nframes = 1;
	const char *exprstrs[5] = {0};
	const struct option opts[] = {{"basis", required_argument, NULL, 2}, {"method", required_argument, NULL, 8}, {0}};
	int c;
	while ((c = getopt_long(argc, argv, "s:v:p:cPr:%n:x:y:S:X:Y:", opts, NULL)) != -1) {
		switch (c) {
		case 's': {                                                    /* zoom.c:153-166 */
			int n = 0;
			if (sscanf(optarg, "%lf%n/%llu%n", &xscale_num, &n, &xscale_den, &n) <= 0) return 2;
			optarg += n;
			if (!*optarg) { yscale_num = xscale_num; yscale_den = xscale_den; break; }
			if (sscanf(optarg, "x%lf/%llu", &yscale_num, &yscale_den) <= 0) return 2;
			break;
		}
		case 'r': sscanf(optarg, "%lfx%lf", &logical_width, &logical_height); break;
		case 'v': sscanf(optarg, "%zux%zu", &vw, &vh); break;
		case 'p': sscanf(optarg, "%lfx%lf", &vx, &vy); break;
		case 'c': centered = true; break;
		case 'P': input_coords = true; break;
		case '%': pct_coords = true; break;
		case 'n': nframes = strtoull(optarg, NULL, 10); break;
		case 'x': exprstrs[0] = optarg; break;
		case 'y': exprstrs[1] = optarg; break;
		case 'S': exprstrs[2] = optarg; break;
		case 'X': exprstrs[3] = optarg; break;
		case 'Y': exprstrs[4] = optarg; break;
		case 2:
			if (!strcmp(optarg, "centered")) scaling_type = CENTERED;
			else if (!strcmp(optarg, "native")) scaling_type = NATIVE;
			else if (strcmp(optarg, "interpolated")) return 2;
			break;
		case 8:
			if (!strcmp(optarg, "fft")) method = M_FFT; else if (!strcmp(optarg, "czt")) method = M_CZT;
			else if (!strcmp(optarg, "gemm")) method = M_GEMM; else if (strcmp(optarg, "auto")) return 2;
			break;
		default: return 2;
		}
	}
	if (argc - optind < 2) { fprintf(stderr, "usage: %s [options] <input> <output.raw>\n", argv[0]); return 2; }
	const char *names[] = {"i", "n", "x", "y", "xs", "ys", "w", "h", "vw", "vh", NULL};      /* zoom.c:219 */
	struct expr *ex[5] = {0};
	for (int i = 0; i < 5; i++)
		if (exprstrs[i] && !(ex[i] = expr_parse(exprstrs[i], names))) { fprintf(stderr, "cannot parse \"%s\"\n", exprstrs[i]); return 1; }

	size_t width, height;
	coeff *coeffs;
	if (read_image_coeff(argv[optind], &width, &height, &coeffs)) { fprintf(stderr, "cannot read %s\n", argv[optind]); return 1; }
	/* zoom.c:263-265 */
	fftw(plan) p = fftw(plan_many_r2r)(2, (int[]){(int)height, (int)width}, 3, coeffs, NULL, 3, 1, coeffs, NULL, 3, 1, (fftw_r2r_kind[]){FFTW_REDFT10, FFTW_REDFT10}, FFTW_ESTIMATE);
	if (!p) { fprintf(stderr, "no plan\n"); return 1; }
	fftw(execute)(p);
	fftw(destroy_plan)(p);

	if (logical_width) { xscale_num = logical_width; xscale_den = width; }      /* zoom.c:268-275 */
	if (logical_height) { yscale_num = logical_height; yscale_den = height; }
	if (width * xscale_num / xscale_den < 1) { xscale_num = 1; xscale_den = width; }      /* :277-284 */
	if (height * yscale_num / yscale_den < 1) { yscale_num = 1; yscale_den = height; }
	if (!vw) vw = width * xscale_num / xscale_den;                     /* :286-289 */
	if (!vh) vh = height * yscale_num / yscale_den;
	if (pct_coords) { vx *= vw / 100; vy *= vy / 100; }                /* :292-295, as written there */
	else if (input_coords) { vx *= xscale_num / xscale_den; vy *= yscale_num / yscale_den; }
	else if (centered) { vx = (width * xscale_num / xscale_den - vw) / 2; vy = (height * yscale_num / yscale_den - vh) / 2; }

	const size_t npix = width * height, nout = vw * vh * 3;
	float *h32 = malloc(sizeof(float) * npix * 3), *d_coeffs = NULL, *d_out = NULL, *frame = malloc(sizeof(float) * nout);
	for (size_t i = 0; i < npix * 3; i++) h32[i] = (float)coeffs[i];
	HIP(hipMalloc((void **)&d_coeffs, sizeof(float) * npix * 3));
	HIP(hipMalloc((void **)&d_out, sizeof(float) * nout));
	HIP(hipMemcpy(d_coeffs, h32, sizeof(float) * npix * 3, hipMemcpyHostToDevice));
	free(h32);
	FILE *f = fopen(argv[optind + 1], "wb");
	if (!f) { perror(argv[optind + 1]); return 1; }
	fprintf(f, "PFS\n%zu %zu %zu\n-1.0\n", vw, vh, nframes);

	size_t written = 0;
	for (size_t d = 0; d < nframes; d++) {                             /* zoom.c:320-345 */
		double vars[] = {d, nframes, vx, vy, xscale_num / xscale_den, yscale_num / yscale_den, width, height, vw, vh, 0};
		if (ex[2]) { xscale_num = yscale_num = expr_eval(ex[2], vars); xscale_den = yscale_den = 1; }
		if (ex[3]) { xscale_num = expr_eval(ex[3], vars); xscale_den = 1; }
		if (ex[4]) { yscale_num = expr_eval(ex[4], vars); yscale_den = 1; }
		vars[4] = xscale_num / xscale_den; vars[5] = yscale_num / yscale_den;
		if (ex[0]) vx = expr_eval(ex[0], vars);
		if (ex[1]) vy = expr_eval(ex[1], vars);
		if (!(isfinite(vx) && isfinite(vy) && isfinite(xscale_num / xscale_den) && isfinite(yscale_num / yscale_den))) {
			fprintf(stderr, "Skipping non-finite expression result at frame %zu\n", d);
			continue;
		}
		const char *how = NULL;
		float *d_work = NULL;
		if (method == M_AUTO || method == M_FFT) {
			dspfft_zoomfft z = NULL;
			const int rc = dspfft_zoomfft_create(&z, (int)width, (int)height, scaling_type, xscale_num, (double)xscale_den, yscale_num, (double)yscale_den, (int)vw, (int)vh);
			if (rc == 0) {
				HIP(hipMalloc((void **)&d_work, sizeof(float) * dspfft_zoomfft_work_floats(z)));
				if (dspfft_zoomfft_execute(z, d_coeffs, vx, vy, d_out, d_work, NULL)) { fprintf(stderr, "zoomfft: %s\n", dspfft_zoomfft_last_error()); return 1; }
				dspfft_zoomfft_destroy(z);
				how = "fft";
			} else if (rc != -2 || method == M_FFT) { fprintf(stderr, "zoomfft: %s\n", dspfft_zoomfft_last_error()); return 1; }
		}
		if (!how && (method == M_AUTO || method == M_CZT)) {
			dspfft_zoomczt z = NULL;
			const int rc = dspfft_zoomczt_create(&z, (int)width, (int)height, scaling_type, xscale_num, (double)xscale_den, yscale_num, (double)yscale_den, (int)vw, (int)vh);
			if (rc == 0) {
				HIP(hipMalloc((void **)&d_work, sizeof(float) * dspfft_zoomczt_work_floats(z)));
				if (dspfft_zoomczt_execute(z, d_coeffs, vx, vy, d_out, d_work, NULL)) { fprintf(stderr, "zoomczt: %s\n", dspfft_zoomfft_last_error()); return 1; }
				dspfft_zoomczt_destroy(z);
				how = "czt";
			} else if (rc != -2 || method == M_CZT) { fprintf(stderr, "zoomczt: %s\n", dspfft_zoomfft_last_error()); return 1; }
		}
		if (!how) {                                                    /* zoom.c:347-375 as two dense products */
			const size_t cw = dspfft_zoom_ncomponents(xscale_num, (double)xscale_den, width), ch = dspfft_zoom_ncomponents(yscale_num, (double)yscale_den, height);
			float *xb = NULL, *yb = NULL;
			HIP(hipMalloc((void **)&xb, sizeof(float) * vw * cw));
			HIP(hipMalloc((void **)&yb, sizeof(float) * vh * ch));
			HIP(hipMalloc((void **)&d_work, sizeof(float) * dspfft_zoom_work_floats((int)width, (int)height, ch, (int)vw)));
			if (dspfft_zoom_basis(xb, scaling_type, xscale_num, (double)xscale_den, vx, vw, width, NULL) ||
			    dspfft_zoom_basis(yb, scaling_type, yscale_num, (double)yscale_den, vy, vh, height, NULL) ||
			    dspfft_zoom_product(d_coeffs, (int)width, (int)height, xb, cw, yb, ch, d_out, (int)vw, (int)vh, d_work, NULL)) {
				fprintf(stderr, "zoom product: %s\n", dspfft_zoom_last_error()); return 1;
			}
			HIP(hipDeviceSynchronize());
			HIP(hipFree(xb)); HIP(hipFree(yb));
			how = "gemm";
		}
		HIP(hipMemcpy(frame, d_out, sizeof(float) * nout, hipMemcpyDeviceToHost));
		HIP(hipFree(d_work));
		if (fwrite(frame, sizeof(float), nout, f) != nout) { perror("write"); return 1; }
		written++;
		fprintf(stderr, "frame %zu: scale %g x %g at (%g, %g) by %s\n", d, xscale_num / xscale_den, yscale_num / yscale_den, vx, vy, how);
	}
	fclose(f);
	for (int i = 0; i < 5; i++) if (ex[i]) expr_free(ex[i]);
	HIP(hipFree(d_coeffs)); HIP(hipFree(d_out));
	free(frame); fftw(free)(coeffs); fftw(cleanup)();
	return written ? 0 : 1;
}
