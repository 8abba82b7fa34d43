// zoom_fft.hip -- zoom's basis x coefficient product (zoom/zoom.c:36-68,361-375) by fast transforms, for the scales at which the
// output samples of an axis lie on a DCT-III grid ("an O(N log N) alternative for integer scales").
//
// For the `interpolated` and `native` bases an output sample b of an axis of length `len` at scale s = num/den is
//     out[b] = sum'_n C[n] cos(pi n (alpha b + beta))        (sum' halves n = 0; n < ncomponents, zoom.c:41,364,369)
// with alpha = 1 / (len s) for both (zoom.c:49-57).  Whenever M = len s is an integer this is
//     cos(pi n (b + 1/2) / M + theta n),   theta = pi (off + (s - 1) / 2) / M  (interpolated),  pi off / M  (native)
//   = cos(theta n) cos(pi n (b + 1/2) / M) - sin(theta n) sin(pi n (b + 1/2) / M)
// and, with n' = M - n, sin(pi (b + 1/2)(M - n') / M) = (-1)^b cos(pi (b + 1/2) n' / M).  So per axis
//     out[b] = 1/2 REDFT01_M(A)[b] - (-1)^b 1/2 REDFT01_M(E)[b],   A[n] = C[n] cos(theta n),  E[n'] = C[M - n'] sin(theta (M - n'))
// (zero where n >= ncomponents): two length-M REDFT01s per axis on the engine's own row / column kernels instead of a dense
// (len s) x len product -- BASELINE config 3 (1920x1080 -> 7680x4320): 310 GFLOP of MFMA work become about 1 GB of streaming in three transform launches (4.4 GB in the first cut).
// Order of the stages (round 3): y first, on the cw columns the coefficients have, as windowed column passes (the zero-padded rows are
// neither stored nor read); the sine part's column pass alternates its output sign and accumulates, leaving T = YA - (-1)^j YE.  Then x
// as ROW passes -- whole lines loaded and stored, instead of column tiles of 32-byte row segments, in the pass that writes the full
// frame -- and both parts in ONE launch (dspfft_execute_sum2): each reads T through its own window and multiplier table
// (dspfft_plan_set_input_modulation: T[u] cos(theta u) in place, T[Mx - x] sin(theta (Mx - x)) mirrored), the cosine part's output
// line waits in registers while the sine part runs through the same LDS, and the frame is written once.  The y stage's column plans
// read the coefficients themselves the same way (window + per-row multiplier + mirror): three transform launches and one table
// kernel per frame.  Scaled widths without a listed row kernel keep the first cut (x first, column passes last, below).
// The offset only enters theta, so an animation's pan (zoom.c:323-340) re-plans nothing.  Other scales, the `centered` basis and
// viewports wider than M keep the dense product (dspfft_zoom_product, zoom_gemm.hip).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/dspfft.h"

namespace {

thread_local char g_err[256] = "";

// tables: cos(theta n), sin(theta n), evaluated in double.  both tables of a frame in one launch: y interleaved (cos, sin), x planar (xmod) or interleaved
__global__ void zf_tables_kernel(float *csy, double thy, int ny, float *csx, double thx, int nx, int x_planar, int y_planar)
{
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ny + nx; i += gridDim.x * blockDim.x) {
		double s, c;
		if (i < ny) {
			sincos(thy * (double)i, &s, &c);
			if (y_planar) { csy[i] = (float)c; csy[ny + i] = (float)s; } else { csy[2 * i] = (float)c; csy[2 * i + 1] = (float)s; }
			continue;
		}
		const int u = i - ny;
		sincos(thx * (double)u, &s, &c);
		if (x_planar) { csx[u] = (float)c; csx[nx + u] = (float)s; } else { csx[2 * u] = (float)c; csx[2 * u + 1] = (float)s; }
	}
}

// x stage inputs from the coefficients: A[v][u] = C[v][u] cos(theta u), E[v][u] = C[v][M - u] sin(theta (M - u)); rows v < ch, pitch M pixels
__global__ void zf_prep_x_kernel(float *A, float *E, const float *C, const float *cs, int w, int ch, int cw, int M)
{
	const size_t total = (size_t)ch * M;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const int v = (int)(i / M), u = (int)(i - (size_t)v * M);
		float a0 = 0, a1 = 0, a2 = 0, e0 = 0, e1 = 0, e2 = 0;
		if (u < cw) { const float *p = C + ((size_t)v * w + u) * 3; const float c = cs[2 * u]; a0 = p[0] * c; a1 = p[1] * c; a2 = p[2] * c; }
		const int n = M - u;
		if (u >= 1 && n >= 1 && n < cw) { const float *p = C + ((size_t)v * w + n) * 3; const float s = cs[2 * n + 1]; e0 = p[0] * s; e1 = p[1] * s; e2 = p[2] * s; }
		float *pa = A + i * 3, *pe = E + i * 3;
		pa[0] = a0; pa[1] = a1; pa[2] = a2; pe[0] = e0; pe[1] = e1; pe[2] = e2;
	}
}

// between the stages: T[v][b] = YA[v][b] - (-1)^b YE[v][b] (the 1/2 is the row plan's scale), then the y stage's inputs
// A2[v][b] = T[v][b] cos(theta_y v) and E2[My - v][b] = T[v][b] sin(theta_y v) (v >= 1), rows v < ch, pitch vw pixels (float4 lanes when
// vw * 3 % 4 == 0).  The other rows of A2 / E2 are zero: never read when the column plans honour the input window, cleared otherwise.
template <int VEC>
__global__ void zf_mid_kernel(float *A2, float *E2, const float *YA, const float *YE, const float *cs, int ch, int Mx, int vw, int My)
{
	const size_t rowf = (size_t)vw * 3, rowv = rowf / VEC, total = (size_t)ch * rowv;
	typedef float vec __attribute__((ext_vector_type(VEC)));
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const int v = (int)(i / rowv);
		const size_t f = (i - (size_t)v * rowv) * VEC;           // first float of this lane within the row
		const vec ya = *reinterpret_cast<const vec *>(YA + (size_t)v * Mx * 3 + f), ye = *reinterpret_cast<const vec *>(YE + (size_t)v * Mx * 3 + f);
		const float c = cs[2 * v], sn = cs[2 * v + 1];
		vec a, e;
		for (int k = 0; k < VEC; k++) {
			const int b = (int)((f + k) / 3);
			const float t = ya[k] - ((b & 1) ? -ye[k] : ye[k]);
			a[k] = t * c; e[k] = t * sn;
		}
		*reinterpret_cast<vec *>(A2 + (size_t)v * rowf + f) = a;
		if (v >= 1) *reinterpret_cast<vec *>(E2 + (size_t)(My - v) * rowf + f) = e;
	}
}

// out[j][b] = ZA[j][b] - (-1)^j ZE[j][b]   (1 / (2 w h) is the column plan's scale)
template <int VEC>
__global__ void zf_final_kernel(float *out, const float *ZA, const float *ZE, int vw, int vh)
{
	const size_t rowv = (size_t)vw * 3 / VEC, total = (size_t)vh * rowv;
	typedef float vec __attribute__((ext_vector_type(VEC)));
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const int j = (int)(i / rowv);
		const vec a = reinterpret_cast<const vec *>(ZA)[i], e = reinterpret_cast<const vec *>(ZE)[i];
		reinterpret_cast<vec *>(out)[i] = (j & 1) ? a + e : a - e;
	}
}

// ---- x stage last (rows): y stage inputs straight from the coefficients, then the x stage's compact inputs ----
// A[v][u] = C[v][u] cos(theta_y v) in rows [0, ch), E[My - v][u] = C[v][u] sin(theta_y v) (v >= 1) in rows (My - ch, My); pitch cw pixels.
// The other rows are zero: never read when the column plans honour the input window, cleared otherwise.
template <int VEC>
__global__ void zf_prep_y_kernel(float *A, float *E, const float *C, const float *cs, int w, int ch, int cw, int My)
{
	typedef float vec __attribute__((ext_vector_type(VEC)));
	const size_t rowf = (size_t)cw * 3, rowv = rowf / VEC, total = (size_t)ch * rowv;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const int v = (int)(i / rowv);
		const size_t f = (i - (size_t)v * rowv) * VEC;
		const vec c = *reinterpret_cast<const vec *>(C + (size_t)v * w * 3 + f);
		*reinterpret_cast<vec *>(A + (size_t)v * rowf + f) = c * cs[2 * v];
		if (v >= 1) *reinterpret_cast<vec *>(E + (size_t)(My - v) * rowf + f) = c * cs[2 * v + 1];
	}
}
// T[j][u] = YA[j][u] - (-1)^j YE[j][u] (the 1/2 is the column plans' scale), j < vh, u < cw; then the x stage's inputs WITHOUT their zeros:
// AX[j][u] = T[j][u] cos(theta_x u), pitch cw pixels (the row plan's window [0, cw)), and EX[j][q] = T[j][u] sin(theta_x u) at
// q = cw - 1 - u for u >= 1, pitch cw - 1 pixels (sample Mx - u of the window (Mx - cw, Mx))
__global__ void zf_mid_x_kernel(float *AX, float *EX, const float *YA, const float *YE, const float *cs, int vh, int cw)
{
	const size_t rowf = (size_t)cw * 3, total = (size_t)vh * rowf;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const int j = (int)(i / rowf);
		const size_t f = i - (size_t)j * rowf;
		const int u = (int)(f / 3), c = (int)(f - (size_t)u * 3);
		const float ya = YA[i], ye = YE[i];
		const float t = (j & 1) ? ya + ye : ya - ye;
		AX[i] = t * cs[2 * u];
		if (u >= 1) EX[((size_t)j * (cw - 1) + (cw - 1 - u)) * 3 + c] = t * cs[2 * u + 1];
	}
}

size_t r4(size_t floats) { return (floats + 3) & ~(size_t)3; }

// len * num / den as an integer M >= 1, or 0
long long grid_length(int len, double num, double den)
{
	if (!(num > 0) || !(den > 0)) return 0;
	if (len * num / den < 1) return 0;                    // zoom.c:37-40 clamps such scales to one sample: the dense path handles them
	const double m = len * num / den, r = round(m);
	return fabs(m - r) <= 1e-9 * m && r >= 1 && r < (double)(1 << 30) ? (long long)r : 0;
}

}  // namespace

struct dspfft_zoomfft_s {
	int w, h, type, vw, vh;
	long long Mx, My;
	size_t cw, ch;
	double sx, sy;                 // scales
	dspfft_plan rows, colsA, colsE;
	bool windowed;                 // the column plans skip the zero rows of their inputs (dspfft_plan_set_input_window)
	bool fused;                    // the sine part's column plan alternates its output signs and ADDS into the cosine part's result
	                               // (dspfft_plan_set_output_alternate + accumulating execution): no combine kernel
	// x stage LAST (preferred when the row plans can do it): the y stage runs first, on the cw columns the coefficients have (a quarter of
	// the output width at BASELINE config 3), and the pass that writes the full frame is a ROW pass -- whose lines are loaded and stored
	// whole -- instead of a column pass over 32-byte row segments.  Needs row plans that honour the input window (their compact inputs
	// hold only the non-zero samples) and, for the sine part, the alternating accumulating store.
	bool xlast;
	dspfft_plan ycolsA, ycolsE, rowsA, rowsE;
	bool ywindowed;
	// ... and without a kernel between the stages: the y stage's sine part adds -(-1)^j YE[j] into the cosine part's result (T, My x cw
	// pixels), and BOTH row plans read T -- the cosine part T[u] cos(theta u) in place, the sine part T[Mx - x] sin(theta (Mx - x)) mirrored
	// (dspfft_plan_set_input_modulation), the multiplier tables being the only thing a pan changes
	bool xmod;
	bool ymod;                     // the y stage's column plans read the coefficients directly (window + modulation + mirror): no input kernel
	// round 4: the x stage as ONE transform per line and channel -- cosine and sine part as the halves of a packed-FP32 pair, three phases
	// per channel (dspfft_cosrows_*, dct_duo.h) -- where the scaled width has a listed kernel; reads T like the two row plans it replaces
	dspfft_cosrows xrows;
};

extern "C" const char *dspfft_zoomfft_last_error(void) { return g_err; }

extern "C" int dspfft_zoomfft_create(dspfft_zoomfft *out, int w, int h, int type, double xnum, double xden, double ynum, double yden, int vw, int vh)
{
	if (!out || w < 1 || h < 1 || vw < 1 || vh < 1) { snprintf(g_err, sizeof g_err, "bad arguments"); return -1; }
	*out = nullptr;
	if (type != 0 && type != 2) { snprintf(g_err, sizeof g_err, "the centered basis does not sample a DCT-III grid: use dspfft_zoom_product"); return -2; }
	const long long Mx = grid_length(w, xnum, xden), My = grid_length(h, ynum, yden);
	if (!Mx || !My) { snprintf(g_err, sizeof g_err, "length x scale is not an integer: use dspfft_zoom_product"); return -2; }
	if (vw > Mx || vh > My) { snprintf(g_err, sizeof g_err, "viewport larger than the scaled image: use dspfft_zoom_product"); return -2; }
	dspfft_zoomfft z = new dspfft_zoomfft_s();
	z->w = w; z->h = h; z->type = type; z->vw = vw; z->vh = vh; z->Mx = Mx; z->My = My;
	z->cw = dspfft_zoom_ncomponents(xnum, xden, (size_t)w); z->ch = dspfft_zoom_ncomponents(ynum, yden, (size_t)h);
	z->sx = xnum / xden; z->sy = ynum / yden;
	z->rows = z->colsA = z->colsE = z->ycolsA = z->ycolsE = z->rowsA = z->rowsE = nullptr;
	z->xlast = false; z->ywindowed = false; z->xmod = false; z->ymod = false; z->xrows = nullptr;
	const int k01[1] = {DSPFFT_REDFT01};
	{
		// x stage last: rows of Mx RGB pixels read from compact lines (cw resp. cw - 1 pixels: only the window is ever read), written
		// Mx pixels apart; the y stage before it: My rows x cw RGB pixels, in place
		const char *e = getenv("DSPFFT_ZOOM_ORDER");       // "y": keep the column pass last (A/B runs)
		const long long cw = (long long)z->cw, lo = Mx - cw + 1;
		if (!(e && *e == 'y') && cw >= 1 && cw < Mx && Mx * 3 * (long long)vh < (1ll << 31) && My * cw * 3 < (1ll << 31)) {
			const dspfft_iodim xd[1] = {{(int)Mx, 3, 3}};
			const dspfft_iodim xa[2] = {{3, 1, 1}, {vh, (int)(cw * 3), (int)(Mx * 3)}}, xe[2] = {{3, 1, 1}, {vh, (int)((cw - 1) * 3), (int)(Mx * 3)}};
			const dspfft_iodim yd[1] = {{(int)My, (int)(cw * 3), (int)(cw * 3)}}, yb[1] = {{(int)(cw * 3), 1, 1}};
			bool ok = !dspfft_plan_guru_r2r(&z->rowsA, 1, xd, 2, xa, k01, 0) && !dspfft_plan_guru_r2r(&z->ycolsA, 1, yd, 1, yb, k01, 0) && !dspfft_plan_guru_r2r(&z->ycolsE, 1, yd, 1, yb, k01, 0);
			ok = ok && dspfft_plan_set_input_window(z->rowsA, 0, 0, (int)cw) == 1;
			// both row plans on the y stage's result itself?  (the table pointers are set per frame: they live in the caller's work buffer)
			const float probe = 0.f;
			const char *em = getenv("DSPFFT_ZOOM_MID");     // "1": keep the kernel between the stages (A/B runs)
			z->xmod = ok && cw > 1 && !(em && *em == '1') && dspfft_plan_set_output_alternate(z->ycolsE, 0, 1) == 1 && dspfft_plan_set_input_modulation(z->rowsA, 0, &probe, 0) == 1;
			if (ok && cw > 1) ok = !dspfft_plan_guru_r2r(&z->rowsE, 1, xd, 2, z->xmod ? xa : xe, k01, 0);
			if (ok && cw > 1) ok = dspfft_plan_set_input_window(z->rowsE, 0, (int)lo, (int)Mx) == 1 && dspfft_plan_set_output_alternate(z->rowsE, 0, 1) == 1;
			if (ok && z->xmod && dspfft_plan_set_input_modulation(z->rowsE, 0, &probe, (int)Mx) != 1) ok = false;
			if (ok) {
				dspfft_plan_set_scale(z->ycolsA, 0.5f); dspfft_plan_set_scale(z->ycolsE, z->xmod ? -0.5f : 0.5f);
				dspfft_plan_set_scale_f64(z->rowsA, 0.5 / ((double)w * (double)h));
				if (z->rowsE) dspfft_plan_set_scale_f64(z->rowsE, -0.5 / ((double)w * (double)h));
				const int wa = dspfft_plan_set_input_window(z->ycolsA, 0, 0, (int)z->ch);
				const int we = z->ch > 1 ? dspfft_plan_set_input_window(z->ycolsE, 0, (int)(My - (long long)z->ch + 1), (int)My) : 0;
				z->ywindowed = wa == 1 && (we == 1 || z->ch == 1) && z->ch < (size_t)My;
				if (!z->ywindowed) { dspfft_plan_set_input_window(z->ycolsA, 0, 0, 0); dspfft_plan_set_input_window(z->ycolsE, 0, 0, 0); }
				if (!z->xmod) dspfft_plan_set_output_alternate(z->ycolsE, 0, 0);
				// ... and no kernel in front of the y stage either: its two column plans read the coefficients themselves (rows w pixels
				// apart), the cosine part C[v] cos(theta_y v) in place, the sine part C[My - y] sin(theta_y (My - y)) mirrored
				z->ymod = false;
				const char *ep = getenv("DSPFFT_ZOOM_PREP");       // "1": keep the kernel that writes the y stage's inputs (A/B runs)
				if (z->xmod && z->ywindowed && z->ch > 1 && !(ep && *ep == '1') && My * (long long)w * 3 < (1ll << 31)) {
					const dspfft_iodim ydd[1] = {{(int)My, w * 3, (int)(cw * 3)}};
					dspfft_plan da = nullptr, de = nullptr;
					bool yok = !dspfft_plan_guru_r2r(&da, 1, ydd, 1, yb, k01, 0) && !dspfft_plan_guru_r2r(&de, 1, ydd, 1, yb, k01, 0);
					yok = yok && dspfft_plan_set_input_window(da, 0, 0, (int)z->ch) == 1 && dspfft_plan_set_input_modulation(da, 0, &probe, 0) == 1;
					yok = yok && dspfft_plan_set_input_window(de, 0, (int)(My - (long long)z->ch + 1), (int)My) == 1 && dspfft_plan_set_input_modulation(de, 0, &probe, (int)My) == 1 &&
					      dspfft_plan_set_output_alternate(de, 0, 1) == 1;
					if (yok) {
						dspfft_destroy_plan(z->ycolsA); dspfft_destroy_plan(z->ycolsE);
						z->ycolsA = da; z->ycolsE = de;
						dspfft_plan_set_scale(z->ycolsA, 0.5f); dspfft_plan_set_scale(z->ycolsE, -0.5f);
						z->ymod = true;
					} else {
						if (da) dspfft_destroy_plan(da);
						if (de) dspfft_destroy_plan(de);
					}
				}
				// the probes above left the address of a local in the plans' modulation slots: clear them (every execution sets the real tables)
				for (dspfft_plan pm : {z->rowsA, z->rowsE, z->ycolsA, z->ycolsE}) if (pm) dspfft_plan_set_input_modulation(pm, 0, nullptr, 0);
				const char *ex = getenv("DSPFFT_ZOOM_XROWS");      // "0": keep the two-transform row pass (A/B runs)
				if (z->xmod && !(ex && *ex == '0') && dspfft_cosrows_create(&z->xrows, (int)Mx, (int)cw, vw, vh)) z->xrows = nullptr;
				z->xlast = true;
				*out = z;
				return 0;
			}
			for (dspfft_plan *p : {&z->rowsA, &z->rowsE, &z->ycolsA, &z->ycolsE}) { if (*p) dspfft_destroy_plan(*p); *p = nullptr; }
		}
	}
	// x stage: A and E one after the other = 2 ch lines of Mx RGB pixels, transformed along x
	const dspfft_iodim rd[1] = {{(int)Mx, 3, 3}}, rb[2] = {{3, 1, 1}, {(int)(2 * z->ch), (int)(Mx * 3), (int)(Mx * 3)}};
	// y stage: two arrays of My rows x vw RGB pixels, each transformed along y by its own plan: the cosine part's input is non-zero in
	// rows [0, ch), the sine part's in rows (My - ch, My) -- the plans are told (input window), and then the zero rows are neither
	// written by the re-pack kernel nor read by the column pass
	const long long arr = My * (long long)vw * 3;
	if (arr >= (1ll << 31) || Mx * 3 * 2 * (long long)z->ch >= (1ll << 31)) { delete z; snprintf(g_err, sizeof g_err, "frame too large for 31-bit strides"); return -2; }
	const dspfft_iodim cd[1] = {{(int)My, vw * 3, vw * 3}}, cb[1] = {{vw * 3, 1, 1}};
	if (dspfft_plan_guru_r2r(&z->rows, 1, rd, 2, rb, k01, 0) || dspfft_plan_guru_r2r(&z->colsA, 1, cd, 1, cb, k01, 0) || dspfft_plan_guru_r2r(&z->colsE, 1, cd, 1, cb, k01, 0)) {
		snprintf(g_err, sizeof g_err, "plan: %s", dspfft_last_error());
		if (z->rows) dspfft_destroy_plan(z->rows);
		if (z->colsA) dspfft_destroy_plan(z->colsA);
		delete z;
		return -3;
	}
	dspfft_plan_set_scale(z->rows, 0.5f);
	dspfft_plan_set_scale_f64(z->colsA, 0.5 / ((double)w * (double)h));
	dspfft_plan_set_scale_f64(z->colsE, 0.5 / ((double)w * (double)h));
	const int wa = dspfft_plan_set_input_window(z->colsA, 0, 0, (int)z->ch);
	const int we = z->ch > 1 ? dspfft_plan_set_input_window(z->colsE, 0, (int)(My - (long long)z->ch + 1), (int)My) : 0;
	z->windowed = wa == 1 && (we == 1 || z->ch == 1) && z->ch < (size_t)My;
	if (!z->windowed) { dspfft_plan_set_input_window(z->colsA, 0, 0, 0); dspfft_plan_set_input_window(z->colsE, 0, 0, 0); }
	// out = ZA - (-1)^j ZE: the sine part's plan takes the sign and the minus (negative scale) and accumulates into ZA
	z->fused = dspfft_plan_set_output_alternate(z->colsE, 0, 1) == 1;
	if (z->fused) dspfft_plan_set_scale_f64(z->colsE, -0.5 / ((double)w * (double)h));
	*out = z;
	return 0;
}

extern "C" void dspfft_zoomfft_destroy(dspfft_zoomfft z)
{
	if (!z) return;
	for (dspfft_plan p : {z->rows, z->colsA, z->colsE, z->ycolsA, z->ycolsE, z->rowsA, z->rowsE}) if (p) dspfft_destroy_plan(p);
	if (z->xrows) dspfft_cosrows_destroy(z->xrows);
	delete z;
}

// layout of d_work: tables (2 cw + 2 ch floats, rounded up to 4) | AX, EX (2 ch Mx 3) | AY, EY (2 My vw 3)
extern "C" size_t dspfft_zoomfft_work_floats(dspfft_zoomfft z)
{
	if (!z) return 0;
	const size_t tab = (2 * z->cw + 2 * z->ch + 3) & ~(size_t)3;
	if (z->xlast)    // tables | AY, EY (My x cw x 3 each) | without input modulation: AX (vh x cw x 3), EX (vh x (cw - 1) x 3) | full lines when the viewport is narrower than Mx; each rounded up to 16 bytes
		return tab + 2 * r4((size_t)z->My * z->cw * 3) + (z->xmod ? 0 : r4((size_t)z->vh * z->cw * 3) + r4((size_t)z->vh * (z->cw - 1) * 3)) + (z->vw < z->Mx ? (size_t)z->vh * z->Mx * 3 : 0);
	return tab + (size_t)2 * z->ch * z->Mx * 3 + (size_t)2 * z->My * z->vw * 3;
}

extern "C" int dspfft_zoomfft_execute(dspfft_zoomfft z, const float *d_coeffs, double vx, double vy, float *d_out, float *d_work, void *stream)
{
	if (!z || !d_coeffs || !d_out || !d_work) { snprintf(g_err, sizeof g_err, "bad arguments"); return -1; }
	if (15u & ((uintptr_t)d_work | (uintptr_t)d_out)) { snprintf(g_err, sizeof g_err, "d_out and d_work must be 16-byte aligned"); return -1; }
	if (z->xlast && z->ymod && (15u & (uintptr_t)d_coeffs)) { snprintf(g_err, sizeof g_err, "d_coeffs must be 16-byte aligned (the y stage's column passes read it directly)"); return -1; }
	hipStream_t s = (hipStream_t)stream;
	const double pi = 3.14159265358979323846;
	// zoom.c:49-57: interpolated k = (b + off) / s on length len; native k = b + off on length len s
	const double thx = z->type == 0 ? pi * (vx + (z->sx - 1) / 2) / (double)z->Mx : pi * vx / (double)z->Mx;
	const double thy = z->type == 0 ? pi * (vy + (z->sy - 1) / 2) / (double)z->My : pi * vy / (double)z->My;
	float *csx = d_work, *csy = csx + 2 * z->cw;
	const size_t tab = (2 * z->cw + 2 * z->ch + 3) & ~(size_t)3;
	float *AX = d_work + tab, *EX = AX + (size_t)z->ch * z->Mx * 3;
	float *AY = EX + (size_t)z->ch * z->Mx * 3, *EY = AY + (size_t)z->My * z->vw * 3;
	hipLaunchKernelGGL(zf_tables_kernel, dim3(32), dim3(256), 0, s, csy, thy, (int)z->ch, csx, thx, (int)z->cw, z->xlast && z->xmod ? 1 : 0, z->xlast && z->ymod ? 1 : 0);
	if (z->xlast) {
		const size_t cw = z->cw, yarr = (size_t)z->My * cw * 3;
		float *AYx = d_work + tab, *EYx = AYx + r4(yarr);
		float *AXc = EYx + r4(yarr), *EXc = AXc + r4((size_t)z->vh * cw * 3);
		float *full = z->xmod ? AXc : EXc + r4((size_t)z->vh * (cw - 1) * 3);
		if (!z->ywindowed && hipMemsetAsync(AYx, 0, (r4(yarr) + yarr) * sizeof(float), s) != hipSuccess) { snprintf(g_err, sizeof g_err, "memset failed"); return -4; }
		if (z->ywindowed && z->ch == 1 && hipMemsetAsync(EYx, 0, yarr * sizeof(float), s) != hipSuccess) { snprintf(g_err, sizeof g_err, "memset failed"); return -4; }
		if (z->ymod) {
			// T = 1/2 REDFT01(C cos) - (-1)^j 1/2 REDFT01(mirrored C sin), both column plans reading d_coeffs, into AYx
			if (dspfft_plan_set_input_modulation(z->ycolsA, 0, csy, 0) != 1 || dspfft_plan_set_input_modulation(z->ycolsE, 0, csy + z->ch, (int)z->My) != 1 ||
			    dspfft_execute(z->ycolsA, d_coeffs, AYx, stream) || dspfft_execute_masked_accumulate(z->ycolsE, d_coeffs, EYx, AYx, nullptr, 0, 1, stream)) {
				snprintf(g_err, sizeof g_err, "y stage: %s", dspfft_last_error()); return -4;
			}
		} else
		// float4 lanes when every row of the three arrays starts on 16 bytes
		if ((cw * 3) % 4 == 0 && ((size_t)z->w * 3) % 4 == 0 && !(15u & (uintptr_t)d_coeffs))
			hipLaunchKernelGGL(zf_prep_y_kernel<4>, dim3(4096), dim3(256), 0, s, AYx, EYx, d_coeffs, csy, z->w, (int)z->ch, (int)cw, (int)z->My);
		else
			hipLaunchKernelGGL(zf_prep_y_kernel<1>, dim3(4096), dim3(256), 0, s, AYx, EYx, d_coeffs, csy, z->w, (int)z->ch, (int)cw, (int)z->My);
		float *dst = z->vw == z->Mx ? d_out : full;
		if (z->xmod) {
			// T = YA - (-1)^j YE in AYx (the sine part's column pass alternates, carries the minus in its scale and accumulates); the row plans
			// read T through their multiplier tables cos(theta_x u) and sin(theta_x u), the sine part mirrored about Mx
			if (!z->ymod && (dspfft_execute(z->ycolsA, AYx, AYx, stream) || dspfft_execute_masked_accumulate(z->ycolsE, EYx, EYx, AYx, nullptr, 0, 1, stream))) {
				snprintf(g_err, sizeof g_err, "y stage: %s", dspfft_last_error()); return -4;
			}
			if (z->xrows) {
				// out[j][b] = 1 / (w h) sum'_u T[j][u] cos(u (pi (b + 1/2) / Mx + theta_x)), straight into the caller's frame (a narrower viewport
				// clips in the store)
				if (dspfft_cosrows_execute(z->xrows, AYx, (long long)cw * 3, d_out, (long long)z->vw * 3, thx, 1.0 / ((double)z->w * (double)z->h), stream)) {
					snprintf(g_err, sizeof g_err, "x stage: %s", dspfft_last_error()); return -4;
				}
				dst = d_out;
			} else
			if (dspfft_plan_set_input_modulation(z->rowsA, 0, csx, 0) != 1 || dspfft_plan_set_input_modulation(z->rowsE, 0, csx + cw, (int)z->Mx) != 1 ||
			    dspfft_execute_sum2(z->rowsA, z->rowsE, AYx, AYx, dst, stream)) { snprintf(g_err, sizeof g_err, "x stage: %s", dspfft_last_error()); return -4; }
		} else {
		if (dspfft_execute(z->ycolsA, AYx, AYx, stream) || dspfft_execute(z->ycolsE, EYx, EYx, stream)) { snprintf(g_err, sizeof g_err, "y stage: %s", dspfft_last_error()); return -4; }
		hipLaunchKernelGGL(zf_mid_x_kernel, dim3(8192), dim3(256), 0, s, AXc, EXc, AYx, EYx, csx, z->vh, (int)cw);
		// the sine part's window starts at sample Mx - cw + 1: its compact lines are addressed from that many pixels before their start
		// (samples outside the window are not read)
		// (one launch that writes the frame once: dspfft_execute_sum2)
		if (z->rowsE ? dspfft_execute_sum2(z->rowsA, z->rowsE, AXc, EXc - (size_t)(z->Mx - (long long)cw + 1) * 3, dst, stream) : dspfft_execute(z->rowsA, AXc, dst, stream)) {
			snprintf(g_err, sizeof g_err, "x stage: %s", dspfft_last_error()); return -4;
		}
		}
		if (dst != d_out && hipMemcpy2DAsync(d_out, (size_t)z->vw * 3 * sizeof(float), full, (size_t)z->Mx * 3 * sizeof(float), (size_t)z->vw * 3 * sizeof(float), (size_t)z->vh,
		                                     hipMemcpyDeviceToDevice, s) != hipSuccess) { snprintf(g_err, sizeof g_err, "copy failed"); return -4; }
		if (hipGetLastError() != hipSuccess) { snprintf(g_err, sizeof g_err, "launch failed"); return -4; }
		return 0;
	}
	hipLaunchKernelGGL(zf_prep_x_kernel, dim3(4096), dim3(256), 0, s, AX, EX, d_coeffs, csx, z->w, (int)z->ch, (int)z->cw, (int)z->Mx);
	if (dspfft_execute(z->rows, AX, AX, stream)) { snprintf(g_err, sizeof g_err, "x stage: %s", dspfft_last_error()); return -4; }
	// rows the re-pack kernel does not write must read as zero: either the column plans skip them (input window) or they are cleared here
	// (ch == 1: the sine part has no non-zero row at all and no window)
	const size_t arrf = (size_t)z->My * z->vw * 3;
	if (!z->windowed && hipMemsetAsync(AY, 0, 2 * arrf * sizeof(float), s) != hipSuccess) { snprintf(g_err, sizeof g_err, "memset failed"); return -4; }
	if (z->windowed && z->ch == 1 && hipMemsetAsync(EY, 0, arrf * sizeof(float), s) != hipSuccess) { snprintf(g_err, sizeof g_err, "memset failed"); return -4; }
	if ((z->vw * 3) % 4 == 0 && (z->Mx * 3) % 4 == 0)
		hipLaunchKernelGGL(zf_mid_kernel<4>, dim3(8192), dim3(256), 0, s, AY, EY, AX, EX, csy, (int)z->ch, (int)z->Mx, z->vw, (int)z->My);
	else
		hipLaunchKernelGGL(zf_mid_kernel<1>, dim3(8192), dim3(256), 0, s, AY, EY, AX, EX, csy, (int)z->ch, (int)z->Mx, z->vw, (int)z->My);
	if (z->fused) {
		// cosine part straight into the caller's frame when the viewport is the whole scaled height (else in place, rows [0, vh) copied out);
		// the sine part adds -(-1)^j ZE[j] into it in its own store phase
		float *dst = z->vh == z->My ? d_out : AY;
		if (dspfft_execute(z->colsA, AY, dst, stream) || dspfft_execute_masked_accumulate(z->colsE, EY, EY, dst, nullptr, 0, 1, stream)) {
			snprintf(g_err, sizeof g_err, "y stage: %s", dspfft_last_error()); return -4;
		}
		if (dst != d_out && hipMemcpyAsync(d_out, AY, (size_t)z->vh * z->vw * 3 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) { snprintf(g_err, sizeof g_err, "copy failed"); return -4; }
		return hipGetLastError() == hipSuccess ? 0 : -4;
	}
	if (dspfft_execute(z->colsA, AY, AY, stream) || dspfft_execute(z->colsE, EY, EY, stream)) { snprintf(g_err, sizeof g_err, "y stage: %s", dspfft_last_error()); return -4; }
	if ((z->vw * 3) % 4 == 0)
		hipLaunchKernelGGL(zf_final_kernel<4>, dim3(8192), dim3(256), 0, s, d_out, AY, EY, z->vw, z->vh);
	else
		hipLaunchKernelGGL(zf_final_kernel<1>, dim3(8192), dim3(256), 0, s, d_out, AY, EY, z->vw, z->vh);
	if (hipGetLastError() != hipSuccess) { snprintf(g_err, sizeof g_err, "launch failed"); return -4; }
	return 0;
}

// ---- any scale, offset and basis: chirp-z transforms along both axes (round 4; dspfft_cztrows_*, dct_czt.h) ----
// zoom.c:49-61: sample b of an axis sits at k = alpha (b + offset) on a basis of N points: interpolated alpha = den / num, N = len;
// native alpha = 1, N = len num / den; centered alpha = (len - 1) den / (len num - den), N = len.  Per axis
//     out[b] = sum'_n C[n] cos(n (omega b + phi)),   omega = pi alpha / N,  phi = pi (alpha offset + 1/2) / N.
// y first, on the cw columns the coefficients have: the coefficient block is transposed so that a column is a planar line, transformed
// (ch -> vh samples), transposed back to rows of cw RGB pixels, and the x axis runs on those rows (one line per channel, the three
// lines of a row on one XCD so that their interleaved stores meet in its L2).
struct dspfft_zoomczt_s {
	int w, h, type, vw, vh;
	size_t cw, ch;
	double xnum, xden, ynum, yden;
	dspfft_cztrows rows_y, rows_x;
};
namespace {
void czt_axis(int type, double num, double den, double len, double off, double &omega, double &phi)
{
	const double pi = 3.14159265358979323846;
	if (len * num / den < 1) { num = 1; den = len; }          // zoom.c:37-40
	double alpha, N;
	if (type == 2) { alpha = 1.0; N = len * num / den; }
	else if (type == 0) { alpha = den / num; N = len; }
	else { alpha = (len - 1) * den / (len * num - den); N = len; }
	omega = pi * alpha / N;
	phi = pi * (alpha * off + 0.5) / N;
}
}  // namespace

extern "C" int dspfft_zoomczt_create(dspfft_zoomczt *out, int w, int h, int type, double xnum, double xden, double ynum, double yden, int vw, int vh)
{
	if (!out || w < 1 || h < 1 || vw < 1 || vh < 1 || type < 0 || type > 2 || !(xnum > 0) || !(xden > 0) || !(ynum > 0) || !(yden > 0)) { snprintf(g_err, sizeof g_err, "bad arguments"); return -1; }
	*out = nullptr;
	if (type == 1 && (!(w * xnum - xden > 0) || !(h * ynum - yden > 0))) { snprintf(g_err, sizeof g_err, "centered basis: len * scale must exceed 1"); return -2; }
	dspfft_zoomczt z = new dspfft_zoomczt_s();
	z->w = w; z->h = h; z->type = type; z->vw = vw; z->vh = vh; z->xnum = xnum; z->xden = xden; z->ynum = ynum; z->yden = yden;
	z->cw = dspfft_zoom_ncomponents(xnum, xden, (size_t)w); z->ch = dspfft_zoom_ncomponents(ynum, yden, (size_t)h);
	z->rows_y = z->rows_x = nullptr;
	if ((long long)z->cw * 3 * (long long)(vh > (int)z->ch ? vh : (int)z->ch) >= (1ll << 31)) { delete z; snprintf(g_err, sizeof g_err, "frame too large for 31-bit strides"); return -2; }
	const int ry = dspfft_cztrows_create(&z->rows_y, (int)z->ch, vh, (int)z->cw * 3, 1);
	const int rx = ry ? ry : dspfft_cztrows_create(&z->rows_x, (int)z->cw, vw, vh * 3, 3);
	if (ry || rx) {
		snprintf(g_err, sizeof g_err, "%s", dspfft_last_error());
		if (z->rows_y) dspfft_cztrows_destroy(z->rows_y);
		delete z;
		return (ry ? ry : rx) == -2 ? -2 : -3;
	}
	*out = z;
	return 0;
}
extern "C" void dspfft_zoomczt_destroy(dspfft_zoomczt z)
{
	if (!z) return;
	dspfft_cztrows_destroy(z->rows_y); dspfft_cztrows_destroy(z->rows_x);
	delete z;
}
// Ct (cw 3 x ch) | Yt (cw 3 x vh) | T (vh x cw 3)
extern "C" size_t dspfft_zoomczt_work_floats(dspfft_zoomczt z)
{
	if (!z) return 0;
	return r4(z->cw * 3 * z->ch) + r4(z->cw * 3 * (size_t)z->vh) + r4((size_t)z->vh * z->cw * 3);
}
extern "C" int dspfft_zoomczt_execute(dspfft_zoomczt z, const float *d_coeffs, double vx, double vy, float *d_out, float *d_work, void *stream)
{
	if (!z || !d_coeffs || !d_out || !d_work) { snprintf(g_err, sizeof g_err, "bad arguments"); return -1; }
	const long long cols = (long long)z->cw * 3, ch = (long long)z->ch;
	float *Ct = d_work, *Yt = Ct + r4((size_t)(cols * ch)), *T = Yt + r4((size_t)(cols * z->vh));
	double wx, px, wy, py;
	czt_axis(z->type, z->xnum, z->xden, (double)z->w, vx, wx, px);
	czt_axis(z->type, z->ynum, z->yden, (double)z->h, vy, wy, py);
	// columns of the coefficient block as planar lines: Ct[(u, c)][v]
	if (dspfft_transpose_f32(Ct, ch, d_coeffs, (long long)z->w * 3, (int)ch, (int)cols, stream) ||
	    // y axis: ch coefficients -> vh samples per column
	    dspfft_cztrows_execute(z->rows_y, Ct, ch, 0, 1, Yt, z->vh, 0, 1, wy, py, 1.0, stream) ||
	    // back to rows of cw RGB pixels: T[j][u][c]
	    dspfft_transpose_f32(T, cols, Yt, z->vh, (int)cols, z->vh, stream) ||
	    // x axis: channel c of row j is a line of cw coefficients (3 floats apart) -> vw samples of the caller's interleaved frame
	    dspfft_cztrows_execute(z->rows_x, T, cols, 1, 3, d_out, (long long)z->vw * 3, 1, 3, wx, px, 1.0 / ((double)z->w * (double)z->h), stream)) {
		snprintf(g_err, sizeof g_err, "%s", dspfft_last_error()); return -4;
	}
	return 0;
}
