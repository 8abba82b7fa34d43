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

/* ---- applybasis (applybasis/applybasis.c) ---------------------------------------------------
 * Basis functions f(k, n, N, ortho) of applybasis.c:77-140 (ids in the order of the tool's -f option:
 * 0 dft, 1 idft, 2 dct1, 3 dct2, 4 dct3, 5 dct4, 6 dst1, 7 dst2, 8 dst3, 9 dst4, 10 wht, 11 dht) and the
 * forward partial sums of applybasis.c:410-431:
 *   out[k_h][k_w][n_h][n_w][j] = sum_{s_h<P_h, s_w<P_w} f(k_h+off_h, n_h P_h + s_h, H) f(k_w+off_w, n_w P_w + s_w, W) pix[..][j]
 * (the offset is added to the term index while the function is evaluated, :416-418).  Complex result as
 * (re, im) pairs; the loop order is the order of the tool's `.coeff` dump (:443). */
#include <complex.h>
static double complex ab_basis(int func, long long k, long long n, unsigned long long N, int ortho)
{
	const double r2 = sqrt(2.0);
	double c;
	switch (func) {
	case 0: return cexp((-2 * I * M_PI * k * n) / N);
	case 1: return cexp((2 * I * M_PI * k * n) / N);
	case 2: c = (n && N - 1 - n) ? cos((M_PI * (k * n)) / (N - 1)) : (n ? pow(-1, k) : 1.0) / 2; if (ortho) c *= r2; return c;
	case 3: c = cos((M_PI * (k * (2 * n + 1))) / (2 * N)); if (ortho) c *= (k ? r2 : 1); return c;
	case 4: c = n ? cos((M_PI * (n * (2 * k + 1))) / (2 * N)) : 0.5; if (ortho) c *= n ? r2 : 2; return c;
	case 5: c = cos((M_PI * ((2 * k + 1) * (2 * n + 1))) / (4 * N)); if (ortho) c *= r2; return c;
	case 6: c = sin((M_PI * ((k + 1) * (n + 1))) / (N + 1)); if (ortho) c *= r2; return c;
	case 7: c = sin((M_PI * ((k + 1) * (2 * n + 1))) / (2 * N)); if (ortho) c *= (N - 1 - k) ? r2 : 1; return c;
	case 8: c = (N - 1 - n) ? sin((M_PI * ((2 * k + 1) * (n + 1))) / (2 * N)) : pow(-1, k) / 2; if (ortho) c *= (N - 1 - n) ? r2 : 2; return c;
	case 9: c = sin((M_PI * ((2 * k + 1) * (2 * n + 1))) / (4 * N)); if (ortho) c *= r2; return c;
	case 10: {
		unsigned long long L = (unsigned long long)log2((double)N), nn = (unsigned long long)n, kk = (unsigned long long)k;
		unsigned long long sig = (nn & (kk >> (L - 1))) & 1ULL;
		for (L--, nn >>= 1; L; L--, nn >>= 1) sig += (nn & ((kk >> (L - 1)) + (kk >> L))) & 1ULL;
		return pow(-1, (double)sig);
	}
	default: return r2 * cos(2 * M_PI * n * k / N - M_PI / 4);
	}
}
void oracle_applybasis_basis_f64(double *re, double *im, int func, int ortho, long long terms, long long offset, unsigned long long N)
{
	for (long long k = 0; k < terms; k++)
		for (unsigned long long n = 0; n < N; n++) {
			double complex v = ab_basis(func, k + offset, (long long)n, N, ortho);
			re[k * N + n] = creal(v); im[k * N + n] = cimag(v);
		}
}
void oracle_applybasis_partsums_f64(double *out /* [Kh][Kw][Nh][Nw][3][2] */, const double *pix, int w, int h, int func, int ortho,
                                    int Kw, int Kh, int Pw, int Ph, long long offw, long long offh)
{
	const int Nw = w / Pw, Nh = h / Ph;
	for (int kh = 0; kh < Kh; kh++)
		for (int kw = 0; kw < Kw; kw++)
			for (int nh = 0; nh < Nh; nh++)
				for (int nw = 0; nw < Nw; nw++) {
					double complex ps[3] = {0, 0, 0};
					for (int sh = 0; sh < Ph; sh++)
						for (int sw = 0; sw < Pw; sw++) {
							double complex comp = ab_basis(func, kw + offw, (long long)nw * Pw + sw, (unsigned long long)w, ortho) *
							                      ab_basis(func, kh + offh, (long long)nh * Ph + sh, (unsigned long long)h, ortho);
							for (int j = 0; j < 3; j++) ps[j] += comp * pix[((size_t)(nh * Ph + sh) * w + (size_t)nw * Pw + sw) * 3 + j];
						}
					double *o = out + ((((size_t)kh * Kw + kw) * Nh + nh) * Nw + nw) * 6;
					for (int j = 0; j < 3; j++) { o[2 * j] = creal(ps[j]); o[2 * j + 1] = cimag(ps[j]); }
				}
}

/* the general loop of applybasis.c:370-380,410-425: K basis functions x N blocks of P pixels per axis (forward: K = terms, N = size / P;
 * --inverse: K = size, N = terms / P), complex pixels (a .coeff input; pix_im may be NULL).  The offset is added to `bi` around the
 * function call (:416-420): forward k = &bi, --inverse n = &bi -- then the function sees (n + off) P + s, the pixel index does not. */
void oracle_applybasis_partsums_ex_f64(double *out /* [Kh][Kw][Nh][Nw][3][2] */, const double *pix_re, const double *pix_im, int w, int h, int func, int ortho,
                                       int Kw, int Kh, int Nw, int Nh, int Pw, int Ph, long long offw, long long offh, int inverse)
{
	const long long kow = inverse ? 0 : offw, koh = inverse ? 0 : offh, now = inverse ? offw : 0, noh = inverse ? offh : 0;
	for (int kh = 0; kh < Kh; kh++)
		for (int kw = 0; kw < Kw; kw++)
			for (int nh = 0; nh < Nh; nh++)
				for (int nw = 0; nw < Nw; nw++) {
					double complex ps[3] = {0, 0, 0};
					for (int sh = 0; sh < Ph; sh++)
						for (int sw = 0; sw < Pw; sw++) {
							double complex comp = ab_basis(func, kw + kow, (long long)(nw + now) * Pw + sw, (unsigned long long)w, ortho) *
							                      ab_basis(func, kh + koh, (long long)(nh + noh) * Ph + sh, (unsigned long long)h, ortho);
							const size_t p = ((size_t)(nh * Ph + sh) * w + (size_t)nw * Pw + sw) * 3;
							for (int j = 0; j < 3; j++) ps[j] += comp * (pix_re[p + j] + (pix_im ? I * pix_im[p + j] : 0));
						}
					double *o = out + ((((size_t)kh * Kw + kw) * Nh + nh) * Nw + nw) * 6;
					for (int j = 0; j < 3; j++) { o[2 * j] = creal(ps[j]); o[2 * j + 1] = cimag(ps[j]); }
				}
}

/* applybasis.c:20-76 (realize / rescale / range) and :392-442 (the rendered frame) */
static double ab_rescale_f64(int type, double c, double scale)
{
	switch (type) {
	case 1: return copysign(log1p(fabs(c)) / log1p(scale), c);
	case 2: { double r = sqrt(scale); c /= r; return copysign(log1p(fabs(c)) / log1p(r), c); }
	case 3: c /= scale; return copysign(log1p(fabs(c)) / log1p(1.0), c);
	default: return c / scale;
	}
}
void oracle_applybasis_render_f64(double *frame /* fh x fw x 4, pre-filled by the caller */, const double *parts, int Kw, int Kh, int Nw, int Nh, int inverse,
                                  int scale, int padding, int plane, int rescale0, int rescale1, int range, double coeff_scale, double insize_wh)
{
	const long long tw = inverse ? Nw : Kw;
	const long long fw = (long long)Kw * Nw * scale + (long long)padding * tw + padding;
	for (int kh = 0; kh < Kh; kh++) for (int kw = 0; kw < Kw; kw++) for (int nh = 0; nh < Nh; nh++) for (int nw = 0; nw < Nw; nw++) {
		const double *p = parts + ((((size_t)kh * Kw + kw) * Nh + nh) * Nw + nw) * 6;
		double v[3];
		for (int j = 0; j < 3; j++) {
			double complex z = p[2 * j] + I * p[2 * j + 1];
			double r = plane == 1 ? cimag(z) : plane == 2 ? cabs(z) : plane == 3 ? carg(z + I * 2.220446049250313e-16) / M_PI : creal(z);
			double c0 = ab_rescale_f64(rescale0, r, coeff_scale);
			if (rescale1 >= 0) {
				double c1 = ab_rescale_f64(rescale1, r, coeff_scale), NN = sqrt(insize_wh) - 1, nn = sqrt(coeff_scale) - 1;
				c0 = ((NN - nn) * c0 + nn * c1) / NN;
			}
			v[j] = c0;
		}
		if (range == 0) for (int j = 0; j < 3; j++) v[j] = (v[j] + 1) / 2;
		else if (range == 1) for (int j = 0; j < 3; j++) v[j] = fabs(v[j]);
		else if (range == 2) for (int j = 0; j < 3; j++) v[j] += v[j] < 0;
		else if (!(v[0] >= 0 && v[1] >= 0 && v[2] >= 0)) {
			double a = fabs(v[0]), b = fabs(v[1]), c = fabs(v[2]);
			v[0] = (-a + 2 * b + 2 * c) / 3; v[1] = (2 * a - b + 2 * c) / 3; v[2] = (2 * a + 2 * b - c) / 3;
		}
		const size_t bw = inverse ? nw : kw, bh = inverse ? nh : kh, iw = inverse ? kw : nw, ih = inverse ? kh : nh;
		const size_t sw_ = inverse ? Kw : Nw, sh_ = inverse ? Kh : Nh;
		const size_t x0 = (sw_ * bw + iw) * scale + (size_t)padding * bw + padding, y0 = (sh_ * bh + ih) * scale + (size_t)padding * bh + padding;
		for (int ys = 0; ys < scale; ys++) for (int xs = 0; xs < scale; xs++) {
			double *o = frame + ((y0 + ys) * (size_t)fw + x0 + xs) * 4;
			o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = 1;
		}
	}
}

/* ---- spectrogram encoding / decoding and motion's coefficient filters, as the reference computes them with
 * COEFF_PRECISION=F, INTERMEDIATE_PRECISION=D (coeff = float, intermediate = double) ------------------------- */

/* spec/spec.c:88-139.  rangetype 0 one / 1 dc / 2 dcs; scaletype 0 log / 1 linear; signtype 0 abs / 1 shift / 2 saturate / 3 retain */
void oracle_spec_encode_f32(float *f, size_t npix, int d, double gain, int rangetype, int scaletype, int signtype)
{
	const size_t l = npix * d;
	for (size_t i = 0; i < l; i++) f[i] *= gain;
	float max[16];
	if (rangetype == 0) *max = gain;
	else if (rangetype == 1) { *max = f[0]; for (int z = 1; z < d; z++) if (f[z] > *max) *max = f[z]; }
	else for (int z = 0; z < d; z++) max[z] = f[z];
	if (rangetype != 2) for (int z = 1; z < d; z++) max[z] = max[0];
	if (scaletype == 0) {
		for (int z = 0; z < d; z++) max[z] = log1pf(max[z]);
		for (size_t i = 0; i < l; i++) f[i] = copysign(log1p(fabsf(f[i])), f[i]) / max[i % d];
	} else for (size_t i = 0; i < l; i++) f[i] /= max[i % d];
	if (signtype == 0) for (size_t i = 0; i < l; i++) f[i] = fabsf(f[i]);
	else if (signtype == 1) for (size_t i = 0; i < l; i++) f[i] = (f[i] / 2. + 0.5) * 254 / 255;
	else if (signtype == 2) for (size_t i = d; i < l; i++) f[i] = !signbit(f[i]);
}

/* spec/ispec.c:100-151 (+ the preserve_dc store of :161-163, which the fused plan scaling leaves unchanged) */
void oracle_ispec_decode_f32(float *f, size_t npix, int d, double gain, int rangetype, int scaletype, int signtype, const double *DC, int restore_dc)
{
	const size_t l = npix * d;
	if (signtype == 1) for (size_t i = 0; i < l; i++) f[i] = (f[i] * 255. / 254 - 0.5) * 2;
	else if (signtype == 2) for (size_t i = d; i < l; i++) f[i] = f[i] * 2 - 1;
	float max[16];
	if (rangetype == 0) *max = gain;
	else if (rangetype == 1) { *max = DC[0] * gain; for (int z = 1; z < d; z++) if (DC[z] * gain > *max) *max = DC[z] * gain; }
	else for (int z = 0; z < d; z++) max[z] = DC[z] * gain;
	if (rangetype != 2) for (int z = 1; z < d; z++) max[z] = max[0];
	if (scaletype == 0) {
		for (int z = 0; z < d; z++) max[z] = log1p(max[z]);
		for (size_t i = 0; i < l; i++) f[i] = copysign(expm1(fabsf(f[i] * max[i % d])), f[i]);
	} else for (size_t i = 0; i < l; i++) f[i] *= max[i % d];
	for (size_t i = 0; i < l; i++) f[i] /= gain;
	if (restore_dc) for (int z = 0; z < d; z++) f[z] = DC[z];
}

/* motion/motion.c:650,683-744 on one block (expr == NULL, coeff_limit == 0) */
unsigned long long oracle_motion_filter_f32(float *coeffs, const int active[3], const int minbuf_hw[2], const int bb[3], const int be[3],
                                            float damp, float boost, float thr_lo, float thr_hi, int preserve_dc, double grey_add, float quantizer)
{
	const int ad = active[0], ah = active[1], aw = active[2], mh = minbuf_hw[0], mw = minbuf_hw[1];
	unsigned long long coded = 0;
#define AT(z, y, x) coeffs[((size_t)(z) * mh + (y)) * mw + (x)]
	float dc = coeffs[0];
	if (damp != 1) {
		if (bb[0]) for (int z = 0; z < bb[0]; z++) for (int y = 0; y < ah; y++) for (int x = 0; x < aw; x++) AT(z, y, x) *= damp;
		if (be[0] < ad) for (int z = be[0]; z < ad; z++) for (int y = 0; y < ah; y++) for (int x = 0; x < aw; x++) AT(z, y, x) *= damp;
		if (bb[1]) for (int z = bb[0]; z < be[0]; z++) for (int y = 0; y < bb[1]; y++) for (int x = 0; x < aw; x++) AT(z, y, x) *= damp;
		if (be[1] < ah) for (int z = bb[0]; z < be[0]; z++) for (int y = be[1]; y < ah; y++) for (int x = 0; x < aw; x++) AT(z, y, x) *= damp;
		if (bb[2]) for (int z = bb[0]; z < be[0]; z++) for (int y = bb[1]; y < be[1]; y++) for (int x = 0; x < bb[2]; x++) AT(z, y, x) *= damp;
		if (be[2] < aw) for (int z = bb[0]; z < be[0]; z++) for (int y = bb[1]; y < be[1]; y++) for (int x = be[2]; x < aw; x++) AT(z, y, x) *= damp;
	}
	if (boost != 1) for (int z = bb[0]; z < be[0]; z++) for (int y = bb[1]; y < be[1]; y++) for (int x = bb[2]; x < be[2]; x++) AT(z, y, x) *= boost;
	if (thr_hi > 0) for (int z = 0; z < ad; z++) for (int y = 0; y < ah; y++) for (int x = 0; x < aw; x++) { float c = fabsf(AT(z, y, x)); if (c < thr_lo || c > thr_hi) AT(z, y, x) = 0; }
	if (preserve_dc) {
		int dcstop = bb[0] || bb[1] || bb[2];
		if (dcstop || boost != 1 || thr_hi > 0) { if (preserve_dc == 1) coeffs[0] = dc; else coeffs[0] += grey_add; }
	}
	/* (the quotient of two floats is a float division, as in motion.c:744 where both are `coeff`) */
	if (quantizer > 0) for (int z = 0; z < ad; z++) for (int y = 0; y < ah; y++) for (int x = 0; x < aw; x++) coded += !!(AT(z, y, x) = round(AT(z, y, x) / quantizer) * quantizer);
#undef AT
	return coded;
}
