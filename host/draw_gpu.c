/*
 * draw_gpu -- plain-C harness reproducing the core of the reference's `draw` (applybasis/draw.c:66-76): sparse DCT
 * coefficients on a zeroed canvas -> fftw(plan_r2r_2d)(h, w, coefs, coefs, REDFT01, REDFT01, ESTIMATE) -> fftw(execute) -> image,
 * over include/fftw3.h, i.e. the call the tool makes, served by the MI355X engine.  The tool writes the canvas through MagickWand
 * (draw.c:78-88: "I" = one intensity channel); here it leaves as raw floats.
 *
 *   draw_gpu <W>x<H> <out.raw> [<X>x<Y>:<strength> | <X>x<Y>] ...     (draw.c:43-52: a component without a strength shares
 *                                                                      what the given strengths leave of 1)
 * out.raw: "P1F\nW H\n" + W*H samples of the build's coeff type (float; double with -DCOEFF_PRECISION_D), top row first.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fftw3.h>
#include "precision.h"

struct coef { int x, y; coeff w; };

int main(int argc, char *argv[])
{
	int w = 512, h = 512;                                             /* draw.c:36 */
	if (argc < 3 || sscanf(argv[1], "%dx%d", &w, &h) != 2 || w < 1 || h < 1) {
		fprintf(stderr, "usage: %s <W>x<H> <out.raw> [<X>x<Y>[:<strength>]] ...\n", argv[0]);
		return 2;
	}
	const int fns = argc - 3;
	struct coef *ba = calloc(fns ? fns : 1, sizeof *ba);
	int nc = 0;
	coeff energy = 0;
	for (int i = 0; i < fns; i++) {                                   /* draw.c:45-52 */
		double s = -1;
		ba[i].w = -1;
		if (sscanf(argv[3 + i], "%dx%d:%lf", &ba[i].x, &ba[i].y, &s) == 2) nc++;
		else { ba[i].w = (coeff)s; energy += ba[i].w; }
		if (ba[i].x < 0 || ba[i].x >= w || ba[i].y < 0 || ba[i].y >= h) { fprintf(stderr, "component %s lies outside the canvas\n", argv[3 + i]); return 2; }
	}
	for (int i = 0; i < fns; i++)
		if (ba[i].w == -1) ba[i].w = (1 - energy) / nc;               /* draw.c:64-65 */
	coeff *coefs = fftw(alloc_real)((size_t)w * h);                   /* draw.c:66-67 */
	if (!coefs) return 1;
	memset(coefs, 0, sizeof(*coefs) * (size_t)w * h);
	for (int i = 0; i < fns; i++)
		coefs[(size_t)ba[i].y * w + ba[i].x] = ba[i].w / 4;           /* draw.c:69-70 */
	coefs[0] += (coeff)0.5;                                           /* draw.c:71 */
	free(ba);

	fftw(plan) p = fftw(plan_r2r_2d)(h, w, coefs, coefs, FFTW_REDFT01, FFTW_REDFT01, FFTW_ESTIMATE);   /* draw.c:74 */
	fftw(execute)(p);
	fftw(destroy_plan)(p);

	FILE *f = fopen(argv[2], "wb");
	if (!f) { perror(argv[2]); return 1; }
	fprintf(f, "P1F\n%d %d\n", w, h);
	const int ok = fwrite(coefs, sizeof *coefs, (size_t)w * h, f) == (size_t)w * h;
	fclose(f);
	fftw(free)(coefs);                                                /* draw.c:90-92 */
	fftw(cleanup)();
	return ok ? 0 : 1;
}
