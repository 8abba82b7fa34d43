/* host/rawio.h -- tiny image I/O for the harnesses (the reference uses MagickWand / libav, which
 * are outside the hot path: SURVEY.md section 2 rows 9, 11).  Formats:
 *   P6 PPM (8-bit RGB)  -> coeff in [0,1] (what MagickExportImagePixels(FloatPixel) yields, spec/spec.c:60)
 *   PF  PFM-like raw    -> "PF\nW H\n-1.0\n" + W*H*3 little-endian f32, top row first (no flip)
 *   PD  same header with magic "PD" and f64 samples (written by a COEFF_PRECISION_D build, read by either) */
#ifndef HOST_RAWIO_H
#define HOST_RAWIO_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__attribute__((unused)) static int read_image(const char *path, size_t *w, size_t *h, float **pix /* malloc'd h*w*3 */)
{
	FILE *f = strcmp(path, "-") ? fopen(path, "rb") : stdin;
	if (!f) { perror(path); return 1; }
	char magic[3] = {0};
	int maxv = 0; double scale = 0; unsigned long ww, hh;
	if (fscanf(f, "%2s", magic) != 1) return 1;
	if (!strcmp(magic, "P6")) {
		if (fscanf(f, "%lu %lu %d", &ww, &hh, &maxv) != 3 || maxv != 255) return 1;
		fgetc(f);
		unsigned char *b = malloc(ww * hh * 3);
		if (fread(b, 1, ww * hh * 3, f) != ww * hh * 3) return 1;
		*pix = malloc(sizeof(float) * ww * hh * 3);
		for (size_t i = 0; i < ww * hh * 3; i++) (*pix)[i] = b[i] / 255.0f;
		free(b);
	} else if (!strcmp(magic, "PF")) {
		if (fscanf(f, "%lu %lu %lf", &ww, &hh, &scale) != 3) return 1;
		fgetc(f);
		*pix = malloc(sizeof(float) * ww * hh * 3);
		if (fread(*pix, sizeof(float), ww * hh * 3, f) != ww * hh * 3) return 1;
	} else return 1;
	*w = ww; *h = hh;
	if (f != stdin) fclose(f);
	return 0;
}

/* the same with samples in the build's coeff type; PD files keep their double samples in a D build */
__attribute__((unused)) static int read_image_coeff(const char *path, size_t *w, size_t *h, coeff **pix /* malloc'd h*w*3 */)
{
	FILE *f = fopen(path, "rb");
	if (!f) { perror(path); return 1; }
	char magic[3] = {0};
	unsigned long ww = 0, hh = 0; double scale; int maxv = 0;
	if (fscanf(f, "%2s", magic) != 1) { fclose(f); return 1; }
	if (!strcmp(magic, "PD")) {
		if (fscanf(f, "%lu %lu %lf", &ww, &hh, &scale) != 3) return 1;
		fgetc(f);
		double *b = malloc(sizeof(double) * ww * hh * 3);
		if (fread(b, sizeof(double), ww * hh * 3, f) != ww * hh * 3) return 1;
		*pix = malloc(sizeof(coeff) * ww * hh * 3);
		for (size_t i = 0; i < ww * hh * 3; i++) (*pix)[i] = (coeff)b[i];
		free(b);
	} else if (!strcmp(magic, "P6")) {
		if (fscanf(f, "%lu %lu %d", &ww, &hh, &maxv) != 3 || maxv != 255) return 1;
		fgetc(f);
		unsigned char *b = malloc(ww * hh * 3);
		if (fread(b, 1, ww * hh * 3, f) != ww * hh * 3) return 1;
		*pix = malloc(sizeof(coeff) * ww * hh * 3);
		for (size_t i = 0; i < ww * hh * 3; i++) (*pix)[i] = b[i] / (coeff)255;   /* QuantumScale in the build's precision */
		free(b);
	} else if (!strcmp(magic, "PF")) {
		if (fscanf(f, "%lu %lu %lf", &ww, &hh, &scale) != 3) return 1;
		fgetc(f);
		float *b = malloc(sizeof(float) * ww * hh * 3);
		if (fread(b, sizeof(float), ww * hh * 3, f) != ww * hh * 3) return 1;
		*pix = malloc(sizeof(coeff) * ww * hh * 3);
		for (size_t i = 0; i < ww * hh * 3; i++) (*pix)[i] = (coeff)b[i];
		free(b);
	} else { fclose(f); return 1; }
	fclose(f);
	*w = ww; *h = hh;
	return 0;
}

__attribute__((unused)) static int write_pf(const char *path, size_t w, size_t h, const float *pix)
{
	FILE *f = strcmp(path, "-") ? fopen(path, "wb") : stdout;
	if (!f) { perror(path); return 1; }
	fprintf(f, "PF\n%zu %zu\n-1.0\n", w, h);
	fwrite(pix, sizeof(float), w * h * 3, f);
	if (f != stdout) fclose(f);
	return 0;
}

/* PF for a float build, PD for a double build */
__attribute__((unused)) static int write_coeff(const char *path, size_t w, size_t h, const coeff *pix)
{
	FILE *f = strcmp(path, "-") ? fopen(path, "wb") : stdout;
	if (!f) { perror(path); return 1; }
	fprintf(f, "%s\n%zu %zu\n-1.0\n", sizeof(coeff) == 8 ? "PD" : "PF", w, h);
	fwrite(pix, sizeof(coeff), w * h * 3, f);
	if (f != stdout) fclose(f);
	return 0;
}
#endif
