/*
 * callsite_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * f64 restatements of the O(N) arithmetic the reference tools wrap around their FFTW calls
 * (SURVEY.md section 8 rows a4, a6, a7 and the motion pixel path of a3).  They exist so the
 * fused GPU epilogues/prologues can be compared with "transform, then the reference's loop".
 * Each function cites the reference lines it follows.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

/* spec/spec.c:70-78 -- first row and first column / sqrt2, everything / (2wh).
 * f is the interleaved h*w*d output of REDFT10^2 (spec/spec.c:63-64). */
void oracle_spec_normalise_f64(double *f, int w, int h, int d)
{
	const double r2 = sqrt(2.0);
	for (size_t xz = 0; xz < (size_t)w * d; xz++) f[xz] /= r2;
	for (int y = 0; y < h; y++)
		for (int z = 0; z < d; z++) f[(size_t)y * w * d + z] /= r2;
	const double norm = (double)w * h * 2;
	for (size_t i = 0; i < (size_t)w * h * d; i++) f[i] /= norm;
}

/* spec/ispec.c:153-159 -- the exact inverse scaling applied before REDFT01^2. */
void oracle_ispec_denormalise_f64(double *f, int w, int h, int d)
{
	const double r2 = sqrt(2.0);
	for (size_t xz = 0; xz < (size_t)w * d; xz++) f[xz] *= r2;
	for (int y = 0; y < h; y++)
		for (int z = 0; z < d; z++) f[(size_t)y * w * d + z] *= r2;
	for (size_t i = 0; i < (size_t)w * h * d; i++) f[i] /= 2;
}

/* scan/scan.c:296-298 -- coeffs /= 4wh */
void oracle_scan_normalise_f64(double *f, int w, int h, int c)
{
	const double s = (double)w * h * 4;
	for (size_t i = 0; i < (size_t)w * h * c; i++) f[i] /= s;
}

/* motion/motion.c:644-647 (dir=+1) and :748-751 (dir=-1): uniform-range scaling of the
 * active {ad,ah,aw} corner of a {.,mh,mw}-embedded block by 2*sqrt2 / prod(sqrt2 if idx==0). */
void oracle_motion_uniform_f64(double *c, int ad, int ah, int aw, int mh, int mw, int dir)
{
	const double r2 = sqrt(2.0);
	for (int z = 0; z < ad; z++)
		for (int y = 0; y < ah; y++)
			for (int x = 0; x < aw; x++) {
				double s = 2 * r2 / ((x ? 1 : r2) * (y ? 1 : r2) * (z ? 1 : r2));
				size_t i = ((size_t)z * mh + y) * mw + x;
				c[i] = dir > 0 ? c[i] * s : c[i] * (1.0 / s);
			}
}

/* motion/motion.c:559-567,756-776 (spec==none, !linear, 8-bit output):
 * pel = c * scalefactor * normalization * normalization; u8 = clamp(lround(pel)).
 * scaled = {sd,sh,sw}; block = {bd,bh,bw}; buffer embedded in {.,mh,mw}. */
void oracle_motion_store_u8_f64(const double *c, uint8_t *pix, int sd, int sh, int sw,
                                int bd, int bh, int bw, int mh, int mw)
{
	const double scalefactor = ((double)sw * sh * sd) / ((double)bw * bh * bd);
	const double normalization = 1 / sqrt((double)sw * sh * sd * 8);
	for (int z = 0; z < sd; z++)
		for (int y = 0; y < sh; y++)
			for (int x = 0; x < sw; x++) {
				size_t i = ((size_t)z * mh + y) * mw + x;
				double pel = c[i] * scalefactor * normalization;
				pel *= normalization;
				pix[i] = pel > 255 ? 255 : pel < 0 ? 0 : (uint8_t)lround(pel);
			}
}

/* zoom/zoom.c:36-68 -- scaled cosine basis.  type: 0 interpolated, 1 centered, 2 native
 * (zoom.c:20-26 order).  basis must hold nvectors*(ncomponents-1); returns ncomponents. */
size_t oracle_zoom_basis_f64(double *basis, int type, double scale_num, double scale_den,
                             double offset, size_t nvectors, size_t len)
{
	if (len * scale_num / scale_den < 1) { scale_num = 1; scale_den = (double)len; }
	double want = round(len * scale_num / scale_den);
	size_t nc = want < (double)len ? (size_t)want : len;
	if (!basis) return nc;
	for (size_t b = 0; b < nvectors; b++)
		for (size_t n = 1; n < nc; n++) {
			double k, N;
			if (type == 2)      { k = b + offset; N = len * scale_num / scale_den; }
			else if (type == 0) { k = (b + offset) * scale_den / scale_num; N = (double)len; }
			else                { k = (b + offset) * (len - 1) * scale_den / (len * scale_num - scale_den); N = (double)len; }
			basis[b * (nc - 1) + n - 1] = cos(M_PI * (k + 0.5) * n / N);
		}
	return nc;
}

/* zoom/zoom.c:361-375 -- dense separable product on unnormalised REDFT10^2 coefficients.
 * coeffs: h*w*3 interleaved; out: vh*vw*3 interleaved. */
void oracle_zoom_product_f64(const double *coeffs, int w, int h,
                             const double *xb, size_t cw, const double *yb, size_t ch,
                             double *out, int vw, int vh)
{
	double *tmp = malloc(sizeof(double) * ch);
	for (int z = 0; z < 3; z++)
		for (int i = 0; i < vw; i++) {
			for (size_t row = 0; row < ch; row++) {
				double t = coeffs[row * w * 3 + z] / 2;
				for (size_t u = 1; u < cw; u++) t += coeffs[(row * w + u) * 3 + z] * xb[i * (cw - 1) + u - 1];
				tmp[row] = t;
			}
			for (int j = 0; j < vh; j++) {
				double s = tmp[0] / 2;
				for (size_t v = 1; v < ch; v++) s += tmp[v] * yb[j * (ch - 1) + v - 1];
				out[((size_t)j * vw + i) * 3 + z] = s / ((double)w * h);
			}
		}
	free(tmp);
}
