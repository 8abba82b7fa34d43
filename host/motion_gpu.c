/*
 * motion_gpu -- plain-C harness reproducing the transform pipeline of the reference's `motion`
 * (motion/motion.c:485-573 plans and constants, :591-811 the block loop) for 8-bit yuv420p/yuv444p/mono Y4M
 * streams over include/fftw3.h: every component plane is one block {w x h x D} per depth-block
 * (the reference's `-b 0x0xD`; D = 0 takes the whole clip as one 3-D block, D = 1 is its default
 * per-frame 2-D transform, motion.c:174), with the uniform-range normalisation (:644-647), an optional
 * quantiser (-q, :570,740-744), the inverse (:748-753) and the 8-bit store (:756-776).  libav I/O
 * (ffapi) is replaced by a minimal Y4M reader/writer; chroma geometry follows motion.c:61-67.
 *
 *   motion_gpu in.y4m out.y4m [D] [quant]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fftw3.h>
#include "precision.h"

struct plane { int w, h; };

int main(int argc, char *argv[])
{
	if (argc < 3) { fprintf(stderr, "usage: %s <in.y4m> <out.y4m> [block depth D=1, 0 = whole clip] [quant]\n", argv[0]); return 2; }
	int depth = argc > 3 ? atoi(argv[3]) : 1;
	const double quant = argc > 4 ? atof(argv[4]) : 0;
	FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[2], "wb");
	if (!in || !out) { perror("open"); return 1; }
	char hdr[256];
	if (!fgets(hdr, sizeof hdr, in) || strncmp(hdr, "YUV4MPEG2", 9)) { fprintf(stderr, "not a Y4M stream\n"); return 1; }
	int W = 0, H = 0, cshift_w = 1, cshift_h = 1, components = 3;
	for (char *t = strtok(hdr, " \n"); t; t = strtok(NULL, " \n")) {
		if (t[0] == 'W') W = atoi(t + 1);
		else if (t[0] == 'H') H = atoi(t + 1);
		else if (t[0] == 'C') {
			if (!strncmp(t + 1, "444", 3)) cshift_w = cshift_h = 0;
			else if (!strncmp(t + 1, "mono", 4)) components = 1;
			else if (strncmp(t + 1, "420", 3)) { fprintf(stderr, "unsupported chroma %s\n", t); return 1; }
		}
	}
	if (W < 1 || H < 1) { fprintf(stderr, "bad Y4M header\n"); return 1; }
	struct plane pl[3];
	size_t frame_bytes = 0;
	for (int i = 0; i < components; i++) {
		/* motion.c:61-67: chroma planes are ceil(dim / 2^shift) */
		pl[i].w = i ? -((-W) >> cshift_w) : W;
		pl[i].h = i ? -((-H) >> cshift_h) : H;
		frame_bytes += (size_t)pl[i].w * pl[i].h;
	}
	/* read the whole clip (the reference stages whole depth-blocks in host RAM too, motion.c:502-511) */
	size_t nframes = 0, cap = 16;
	uint8_t *clip = malloc(cap * frame_bytes);
	char fh[64];
	while (fgets(fh, sizeof fh, in)) {
		if (strncmp(fh, "FRAME", 5)) { fprintf(stderr, "bad frame header\n"); return 1; }
		if (nframes == cap) clip = realloc(clip, (cap *= 2) * frame_bytes);
		if (fread(clip + nframes * frame_bytes, 1, frame_bytes, in) != frame_bytes) break;
		nframes++;
	}
	fclose(in);
	if (!nframes) { fprintf(stderr, "no frames\n"); return 1; }
	if (depth <= 0 || (size_t)depth > nframes) depth = (int)nframes;
	const size_t nblocks_d = nframes / depth;                        /* trailing partial block dropped (motion.c:384-386) */

	fftw(init_threads)();                                            /* motion.c:485-486 */
	fftw(plan_with_nthreads)(1);
	size_t mincomponent = 0;
	for (int i = 0; i < components; i++) if ((size_t)pl[i].w * pl[i].h * depth > mincomponent) mincomponent = (size_t)pl[i].w * pl[i].h * depth;
	coeff *coeffs = fftw(alloc_real)(mincomponent);                  /* motion.c:500: ONE scratch buffer for every block */

	fftw(plan) planforward[3], planinverse[3];
	for (int i = 0; i < components; i++) {                          /* motion.c:522-552, plans shared between equal planes */
		planforward[i] = planinverse[i] = NULL;
		for (int j = 0; j < i; j++) if (pl[j].w == pl[i].w && pl[j].h == pl[i].h) { planforward[i] = planforward[j]; planinverse[i] = planinverse[j]; }
		if (planforward[i]) continue;
		const int n[3] = {depth, pl[i].h, pl[i].w};
		planforward[i] = fftw(plan_many_r2r)(3, n, 1, coeffs, n, 1, 0, coeffs, n, 1, 0,
		                                     (const fftw_r2r_kind[3]){FFTW_REDFT10, FFTW_REDFT10, FFTW_REDFT10}, FFTW_ESTIMATE);
		planinverse[i] = fftw(plan_many_r2r)(3, n, 1, coeffs, n, 1, 0, coeffs, n, 1, 0,
		                                     (const fftw_r2r_kind[3]){FFTW_REDFT01, FFTW_REDFT01, FFTW_REDFT01}, FFTW_ESTIMATE);
	}
	fprintf(out, "YUV4MPEG2 W%d H%d F25:1 Ip A1:1 C%s\n", W, H, components == 1 ? "mono" : cshift_w ? "420jpeg" : "444");
	const intermediate sqrt2 = sqrt(2.0);
	unsigned long long coeffs_coded = 0;
	for (size_t bz = 0; bz < nblocks_d; bz++) {
		size_t plane_off = 0;
		for (int i = 0; i < components; i++) {
			const int w = pl[i].w, h = pl[i].h;
			const size_t psz = (size_t)w * h;
			const intermediate normalization = 1 / sqrt((double)w * h * depth * 8);     /* motion.c:567 */
			const coeff quantizer = quant * 8 * sqrt((double)w * h * depth);             /* motion.c:570 */
			for (int z = 0; z < depth; z++) {                                         /* motion.c:617-638 */
				const uint8_t *src = clip + (bz * depth + z) * frame_bytes + plane_off;
				for (size_t p = 0; p < psz; p++) coeffs[(size_t)z * psz + p] = src[p];
			}
			fftw(execute)(planforward[i]);                                             /* motion.c:641 */
			for (int z = 0; z < depth; z++)                                            /* motion.c:644-647 */
				for (int y = 0; y < h; y++)
					for (int x = 0; x < w; x++)
						coeffs[((size_t)z * h + y) * w + x] *= 2 * sqrt2 / ((x ? 1 : sqrt2) * (y ? 1 : sqrt2) * (z ? 1 : sqrt2));
			if (quant)                                                                 /* motion.c:740-744 */
				for (size_t p = 0; p < psz * depth; p++) coeffs_coded += !!(coeffs[p] = round(coeffs[p] / quantizer) * quantizer);
			for (int z = 0; z < depth; z++)                                            /* motion.c:748-751 */
				for (int y = 0; y < h; y++)
					for (int x = 0; x < w; x++)
						coeffs[((size_t)z * h + y) * w + x] *= ((x ? 1 : sqrt2) * (y ? 1 : sqrt2) * (z ? 1 : sqrt2)) / (2 * sqrt2);
			fftw(execute)(planinverse[i]);                                             /* motion.c:753 */
			for (int z = 0; z < depth; z++) {                                         /* motion.c:756-776, scalefactor = 1 */
				uint8_t *dst = clip + (bz * depth + z) * frame_bytes + plane_off;
				for (size_t p = 0; p < psz; p++) {
					intermediate pel = coeffs[(size_t)z * psz + p] * normalization;
					pel *= normalization;
					dst[p] = pel > 255 ? 255 : pel < 0 ? 0 : lround(pel);
				}
			}
			plane_off += psz;
		}
		for (int z = 0; z < depth; z++) {
			fputs("FRAME\n", out);
			fwrite(clip + (bz * depth + z) * frame_bytes, 1, frame_bytes, out);
		}
	}
	fclose(out);
	if (quant) fprintf(stderr, "coefficients coded: %llu\n", coeffs_coded);
	for (int i = 0; i < components; i++) {
		int shared = 0;
		for (int j = 0; j < i; j++) if (planforward[j] == planforward[i]) shared = 1;
		if (!shared) { fftw(destroy_plan)(planforward[i]); fftw(destroy_plan)(planinverse[i]); }
	}
	fftw(free)(coeffs);
	free(clip);
	fftw(cleanup)(); fftw(cleanup_threads)();                       /* motion.c:832,837 */
	return 0;
}
