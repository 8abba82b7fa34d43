/*
 * scan_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * CPU restatement of the integer scan-order generators of the reference's `scan` tool
 * (scan/scan_methods.c) and of the per-frame reconstruction step of scan/scan.c.
 * Integer work: the contract is bit-exact.
 *
 * Pinning: scan/scan_methods.c itself cannot be compiled here (it includes
 * <libavutil/eval.h>, absent from the image, and stand-in headers are not allowed), so the
 * restatement is pinned to
 *   - the two known-answer listings the reference ships for the 8x8 `diagonal` scan
 *     (scan/README.md:121-129 index format, :136-150 coordinate format), rendered through the
 *     reference's OWN serialiser (scan/scan_precomputed.c:122-153, built into oracle/_ref),
 *   - the FNV-1a-64 hashes of the zigzag order recorded from the compiled reference during the
 *     survey, committed as tests/golden/zigzag_fnv.json.
 *
 * Coordinate convention everywhere: pair[0] = y (row), pair[1] = x (column), as
 * scan/scan_context.c:44 and scan/scan.c:431 use it.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* number of elements in the first d anti-diagonals of an unbounded quadrant */
static inline size_t tri(size_t d) { return d * (d + 1) / 2; }

/* largest d with tri(d) <= i, computed as the reference does (double sqrt, truncation):
 * scan/scan_methods.c:69-71 */
static inline size_t tri_floor(size_t i) { return (size_t)(sqrt((double)(i * 2) + 0.25) - 0.5); }

/* zigzag -- scan/scan_methods.c:77-115.  One coordinate per index, limit = w*h. */
void oracle_scan_zigzag(size_t w, size_t h, size_t i, size_t yx[2])
{
	const size_t m = w < h ? w : h, head = tri(m), area = w * h;
	if (i < head) {                     /* growing anti-diagonals from the DC corner (:83-91) */
		size_t d = tri_floor(i), r = i - tri(d);
		if (d % 2 == 0) r = d - r;
		yx[0] = r; yx[1] = d - r;
		return;
	}
	if (area - i <= head) {             /* shrinking anti-diagonals towards the far corner (:92-101) */
		size_t j = area - i - 1, d = tri_floor(j), r = j - tri(d);
		if (((w + h - 1) - d - 1) % 2 == 0) r = d - r;
		yx[0] = (h - 1) - r; yx[1] = (w - 1) - (d - r);
		return;
	}
	/* constant-length band between the two triangles (:103-114) */
	size_t band = (i - head) / m;
	size_t r = m - (i - (band * m + head));
	if ((band + m) % 2 == 0) r = m - r + 1;
	if (w < h) {
		r = m - r + 1;
		yx[0] = band + r; yx[1] = w - r;
	} else {
		yx[0] = h - r; yx[1] = band + r;
	}
}

/* horizontal / vertical -- scan/scan_methods.c:59-67 */
void oracle_scan_horizontal(size_t w, size_t h, size_t i, size_t yx[2]) { (void)h; yx[0] = i / w; yx[1] = i % w; }
void oracle_scan_vertical(size_t w, size_t h, size_t i, size_t yx[2]) { (void)w; yx[0] = i % h; yx[1] = i / h; }

/* diagonal -- scan/scan_methods.c:160-165; interval :28-32; limit = w+h-1 (:23).
 * Returns the number of coordinates written. */
size_t oracle_scan_diagonal(size_t w, size_t h, size_t i, size_t (*yx)[2])
{
	size_t n = 0;
	size_t y = i < h ? i : h - 1;        /* start on the left/bottom edge, walk up-right */
	size_t x = i - y;
	for (;; y--, x++) {
		if (x >= w) break;
		yx[n][0] = y; yx[n][1] = x; n++;
		if (y == 0) break;
	}
	return n;
}

/* row / column -- scan/scan_methods.c:146-158 */
size_t oracle_scan_row(size_t w, size_t h, size_t i, size_t (*yx)[2])
{ (void)h; for (size_t x = 0; x < w; x++) { yx[x][0] = i; yx[x][1] = x; } return w; }
size_t oracle_scan_column(size_t w, size_t h, size_t i, size_t (*yx)[2])
{ (void)w; for (size_t y = 0; y < h; y++) { yx[y][0] = y; yx[y][1] = i; } return h; }

/* Whole zigzag order as linear offsets y*w+x (for hashing / device comparison). */
void oracle_zigzag_order(size_t w, size_t h, uint64_t *lin)
{
	size_t yx[2];
	for (size_t i = 0; i < w * h; i++) { oracle_scan_zigzag(w, h, i, yx); lin[i] = (uint64_t)(yx[0] * w + yx[1]); }
}

/* FNV-1a-64 over the sequence of linear offsets, one whole 64-bit word per step
 * (hash ^= value; hash *= prime) -- the variant the survey recorded from the compiled
 * reference (the 8x8 value 3429f64e9a8101d3 identifies the variant). */
uint64_t oracle_fnv1a64_u64(const uint64_t *v, size_t n)
{
	uint64_t hsh = 1469598103934665603ULL;
	for (size_t i = 0; i < n; i++) {
		hsh ^= v[i]; hsh *= 1099511628211ULL;
	}
	return hsh;
}
uint64_t oracle_zigzag_fnv(size_t w, size_t h)
{
	uint64_t hsh = 1469598103934665603ULL;
	size_t yx[2];
	for (size_t i = 0; i < w * h; i++) {
		oracle_scan_zigzag(w, h, i, yx);
		hsh ^= (uint64_t)(yx[0] * w + yx[1]); hsh *= 1099511628211ULL;
	}
	return hsh;
}

/*
 * One output frame of scan's hot loop, arithmetic only (scan/scan.c:429-432,445-459):
 *   recon = 0; recon[coords] = coeffs[coords]; recon[DC] = 0; image = REDFT01^2(recon); sum += image
 * coeffs/sum are interleaved h*w*c f64; lin = y*w+x offsets selected for this frame.
 */
int oracle_r2r_many_f64(int rank, const int *n, int howmany, const double *in, const int *inembed,
                        int istride, int idist, double *out, const int *onembed, int ostride, int odist,
                        const int *kinds);

int oracle_scan_frame_f64(int w, int h, int c, const double *coeffs, const uint64_t *lin, size_t nlin, double *sum)
{
	size_t len = (size_t)w * h * c;
	double *recon = calloc(len, sizeof(double)), *image = malloc(len * sizeof(double));
	if (!recon || !image) { free(recon); free(image); return -1; }
	for (size_t i = 0; i < nlin; i++)
		memcpy(recon + lin[i] * c, coeffs + lin[i] * c, sizeof(double) * c);
	memset(recon, 0, sizeof(double) * c);
	int n[2] = {h, w}, kinds[2] = {4, 4};
	int rc = oracle_r2r_many_f64(2, n, c, recon, NULL, c, 1, image, NULL, c, 1, kinds);
	if (!rc) for (size_t i = 0; i < len; i++) sum[i] += image[i];
	free(recon); free(image);
	return rc;
}

/*
 * Adapters with the reference's `struct scan_method` callback signatures
 * (scan/scan_methods.h:12-24) so tests can drive the reference's own context/serialiser
 * (oracle/_ref: scan_context.c, scan_precomputed.c) with the restated generators.
 */
void oracle_method_zigzag(void *o, size_t w, size_t h, size_t i, size_t (*c)[2]) { (void)o; oracle_scan_zigzag(w, h, i, *c); }
void oracle_method_horizontal(void *o, size_t w, size_t h, size_t i, size_t (*c)[2]) { (void)o; oracle_scan_horizontal(w, h, i, *c); }
void oracle_method_vertical(void *o, size_t w, size_t h, size_t i, size_t (*c)[2]) { (void)o; oracle_scan_vertical(w, h, i, *c); }
void oracle_method_diagonal(void *o, size_t w, size_t h, size_t i, size_t (*c)[2]) { (void)o; oracle_scan_diagonal(w, h, i, c); }
void oracle_method_row(void *o, size_t w, size_t h, size_t i, size_t (*c)[2]) { (void)o; oracle_scan_row(w, h, i, c); }
void oracle_method_column(void *o, size_t w, size_t h, size_t i, size_t (*c)[2]) { (void)o; oracle_scan_column(w, h, i, c); }
/* diagonal: limit = w+h-1 (scan_methods.c:23), interval (:28-32), max_interval = min(w,h) (:22) */
size_t oracle_limit_sum(void *o, size_t w, size_t h) { (void)o; return w + h - 1; }
size_t oracle_limit_min(void *o, size_t w, size_t h) { (void)o; return w < h ? w : h; }
size_t oracle_limit_width(void *o, size_t w, size_t h) { (void)o; (void)h; return w; }
size_t oracle_limit_height(void *o, size_t w, size_t h) { (void)o; (void)w; return h; }
size_t oracle_interval_diagonal(void *o, size_t w, size_t h, size_t i)
{
	(void)o;
	size_t lo = w < h ? w : h, hi = w > h ? w : h;
	if (i < lo) return i + 1;
	if (i < hi) return lo;
	return lo - (i - hi) - 1;
}

/* scan/scan.c:20-41 (generate_basis_matrix + pruned_idct) followed by the accumulate of :451-459, in the reference's
 * precisions (coeff = float storage, basis computed in double and stored as float) */
void oracle_scan_pruned_accumulate_f32(float *sum, const float *coeffs, const uint64_t *lin, size_t ncoords, int w, int h, int c)
{
	float *by = malloc(sizeof(float) * (size_t)h * h), *bx = malloc(sizeof(float) * (size_t)w * w);
	for (int k = 0; k < h; k++) { by[(size_t)k * h] = 1; for (int j = 1; j < h; j++) by[(size_t)k * h + j] = 2. * cos(M_PI * j * (k + 0.5) / h); }
	for (int k = 0; k < w; k++) { bx[(size_t)k * w] = 1; for (int j = 1; j < w; j++) bx[(size_t)k * w + j] = 2. * cos(M_PI * j * (k + 0.5) / w); }
	float *image = calloc((size_t)w * h * c, sizeof(float));
	if (ncoords) {
		for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) for (int z = 0; z < c; z++)
			image[((size_t)y * w + x) * c + z] = coeffs[lin[0] * c + z] * by[(size_t)y * h + lin[0] / w] * bx[(size_t)x * w + lin[0] % w];
		for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) for (size_t n = 1; n < ncoords; n++) for (int z = 0; z < c; z++)
			image[((size_t)y * w + x) * c + z] += coeffs[lin[n] * c + z] * by[(size_t)y * h + lin[n] / w] * bx[(size_t)x * w + lin[n] % w];
	}
	for (size_t i = 0; i < (size_t)w * h * c; i++) sum[i] += image[i];
	free(by); free(bx); free(image);
}
