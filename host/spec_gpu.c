/*
 * spec_gpu / ispec_gpu -- plain-C harness reproducing the transform core of the reference's
 * `spec` (spec/spec.c:59-78) and `ispec` (spec/ispec.c:153-167) over the FFTW-named API of
 * include/fftw3.h, i.e. exactly the calls the tools make, served by the MI355X engine.
 * Only the "-t copy" (retain sign, linear, gain 1... i.e. no display encoding) path is reproduced:
 * the display encodings of spec.c:81-139 are elementwise host code outside the hot path.
 *
 *   spec_gpu  spec  in.{ppm,pf} out.pf     uniform-range coefficients in [-1,1]
 *   spec_gpu  ispec in.pf       out.pf     image back
 * spec_gpu_d is the same source built with -DCOEFF_PRECISION_D: coeff = double and fftw(call) = fftw_call, the
 * reference's default build of spec (spec/Makefile:1); its files are "PD" (f64 samples).
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <fftw3.h>
#include "precision.h"
#include "rawio.h"

int main(int argc, char *argv[])
{
	if (argc != 4 || (strcmp(argv[1], "spec") && strcmp(argv[1], "ispec"))) {
		fprintf(stderr, "usage: %s spec|ispec <in> <out.pf>\n", argv[0]);
		return 2;
	}
	const int inverse = !strcmp(argv[1], "ispec");
	size_t w, h, d = 3, l;
	coeff *pix;
	if (read_image_coeff(argv[2], &w, &h, &pix)) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
	l = w * h * d;

	coeff *f = fftw(alloc_real)(l);                                  /* spec.c:59 / ispec.c:80 */
	if (!f) return 1;
	memcpy(f, pix, sizeof(coeff) * l);
	free(pix);
	const intermediate sqrt2 = sqrt(2.0);

	if (!inverse) {
		/* spec/spec.c:63-65 */
		fftw(plan) p = fftw(plan_many_r2r)(2, (int[]){h, w}, d, f, NULL, d, 1, f, NULL, d, 1,
		                                   (fftw_r2r_kind[]){FFTW_REDFT10, FFTW_REDFT10}, FFTW_ESTIMATE);
		fftw(execute)(p);
		fftw(destroy_plan)(p);
		/* spec/spec.c:70-78 */
		for (size_t xz = 0; xz < w * d; xz++) f[xz] /= sqrt2;
		for (size_t y = 0; y < h; y++)
			for (size_t z = 0; z < d; z++) f[y * w * d + z] /= sqrt2;
		intermediate norm = w * h * 2;
		for (size_t i = 0; i < l; i++) f[i] /= norm;
	} else {
		/* spec/ispec.c:153-159 */
		for (size_t xz = 0; xz < w * d; xz++) f[xz] *= sqrt2;
		for (size_t y = 0; y < h; y++)
			for (size_t z = 0; z < d; z++) f[y * w * d + z] *= sqrt2;
		for (size_t i = 0; i < l; i++) f[i] /= 2;
		/* spec/ispec.c:165-167 */
		fftw(plan) p = fftw(plan_many_r2r)(2, (int[]){h, w}, d, f, NULL, d, 1, f, NULL, d, 1,
		                                   (fftw_r2r_kind[]){FFTW_REDFT01, FFTW_REDFT01}, FFTW_ESTIMATE);
		fftw(execute)(p);
		fftw(destroy_plan)(p);
	}
	int rc = write_coeff(argv[3], w, h, f);
	fftw(free)(f);
	fftw(cleanup)();
	return rc;
}
