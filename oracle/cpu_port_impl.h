/*
 * cpu_port_impl.h -- TEST INFRASTRUCTURE ONLY.  Included twice by cpu_port.c with
 * REAL = float / double and SUF = f32 / f64.
 *
 * O(N log N) CPU port of the same path dct_oracle.c states by definition: REDFT10/REDFT01
 * (FFTW 3.3 manual definitions; reference call sites spec/spec.c:63, spec/ispec.c:165,
 * scan/scan.c:292,359, motion/motion.c:535-552) through a mixed-radix complex FFT.
 * Used (a) as the f64 checker at sizes where the O(N^2) definition is too slow, after being
 * validated against it, and (b) as bench.py's `cpu_baseline` ("kind": "port").
 */
#define CAT2(a, b) a##_##b
#define CAT(a, b) CAT2(a, b)
#define FN(name) CAT(name, SUF)

typedef struct { REAL re, im; } FN(cpx);
#define CPX FN(cpx)
static double FN(now_seconds)(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

typedef struct {
	int n;            /* complex length */
	int nfac, fac[40];
	CPX *tw;          /* tw[k] = exp(-2 pi i k / n) */
} FN(fftplan);
#define FFTPLAN FN(fftplan)

static void FN(fft_plan_init)(FFTPLAN *p, int n)
{
	p->n = n; p->nfac = 0;
	int m = n;
	while (m % 4 == 0) { p->fac[p->nfac++] = 4; m /= 4; }
	while (m % 2 == 0) { p->fac[p->nfac++] = 2; m /= 2; }
	for (int f = 3; m > 1; f += 2)
		while (m % f == 0) { p->fac[p->nfac++] = f; m /= f; }
	p->tw = malloc(sizeof(CPX) * (size_t)(n > 0 ? n : 1));
	const long double tau = 6.283185307179586476925286766559005768L;
	for (int k = 0; k < n; k++) {
		p->tw[k].re = (REAL)cosl(tau * k / n);
		p->tw[k].im = (REAL)-sinl(tau * k / n);
	}
}
static void FN(fft_plan_free)(FFTPLAN *p) { free(p->tw); p->tw = NULL; }

/* Decimation-in-time recursion: out[0..n) <- DFT of in[0], in[s], in[2s] ...; tws = n_total / n. */
static void FN(fft_rec)(const FFTPLAN *P, const CPX *in, CPX *out, int n, int s, int level)
{
	if (n == 1) { *out = *in; return; }
	const int p = P->fac[level], m = n / p;
	for (int q = 0; q < p; q++)
		FN(fft_rec)(P, in + (size_t)q * s, out + (size_t)q * m, m, s * p, level + 1);
	const int tws = P->n / n;         /* twiddle stride: w_n^k = tw[k * tws] */
	const CPX *tw = P->tw;
	if (p == 2) {
		for (int k = 0; k < m; k++) {
			CPX a = out[k], b = out[k + m], w = tw[(size_t)k * tws];
			REAL br = b.re * w.re - b.im * w.im, bi = b.re * w.im + b.im * w.re;
			out[k].re = a.re + br; out[k].im = a.im + bi;
			out[k + m].re = a.re - br; out[k + m].im = a.im - bi;
		}
	} else if (p == 4) {
		for (int k = 0; k < m; k++) {
			CPX a = out[k], b = out[k + m], c = out[k + 2 * m], d = out[k + 3 * m];
			CPX w1 = tw[(size_t)k * tws], w2 = tw[(size_t)2 * k * tws], w3 = tw[(size_t)3 * k * tws];
			REAL br = b.re * w1.re - b.im * w1.im, bi = b.re * w1.im + b.im * w1.re;
			REAL cr = c.re * w2.re - c.im * w2.im, ci = c.re * w2.im + c.im * w2.re;
			REAL dr = d.re * w3.re - d.im * w3.im, di = d.re * w3.im + d.im * w3.re;
			REAL s0r = a.re + cr, s0i = a.im + ci, s1r = a.re - cr, s1i = a.im - ci;
			REAL s2r = br + dr, s2i = bi + di, s3r = br - dr, s3i = bi - di;
			out[k].re = s0r + s2r;         out[k].im = s0i + s2i;
			out[k + m].re = s1r + s3i;     out[k + m].im = s1i - s3r;     /* -i * s3 */
			out[k + 2 * m].re = s0r - s2r; out[k + 2 * m].im = s0i - s2i;
			out[k + 3 * m].re = s1r - s3i; out[k + 3 * m].im = s1i + s3r;
		}
	} else if (p == 3) {
		/* w_3 = -1/2 - i sqrt(3)/2 */
		const REAL hs = (REAL)0.86602540378443864676372317075293618L;
		for (int k = 0; k < m; k++) {
			CPX a = out[k], b = out[k + m], c = out[k + 2 * m];
			CPX w1 = tw[(size_t)k * tws], w2 = tw[(size_t)2 * k * tws];
			REAL br = b.re * w1.re - b.im * w1.im, bi = b.re * w1.im + b.im * w1.re;
			REAL cr = c.re * w2.re - c.im * w2.im, ci = c.re * w2.im + c.im * w2.re;
			REAL sr = br + cr, si = bi + ci, dr = (br - cr) * hs, di = (bi - ci) * hs;
			REAL mr = a.re - (REAL)0.5 * sr, mi = a.im - (REAL)0.5 * si;
			out[k].re = a.re + sr;         out[k].im = a.im + si;
			out[k + m].re = mr + di;       out[k + m].im = mi - dr;       /* a + w b' + w^2 c' */
			out[k + 2 * m].re = mr - di;   out[k + 2 * m].im = mi + dr;
		}
	} else if (p == 5) {
		/* c1 = cos(2 pi/5), c2 = cos(4 pi/5), s1 = sin(2 pi/5), s2 = sin(4 pi/5) */
		const REAL c1 = (REAL)0.30901699437494742410229341718281906L, c2 = (REAL)-0.80901699437494742410229341718281906L;
		const REAL s1 = (REAL)0.95105651629515357211643933337938214L, s2 = (REAL)0.58778525229247312916870595463907277L;
		for (int k = 0; k < m; k++) {
			CPX a = out[k], v[4];
			for (int q = 1; q < 5; q++) {
				CPX b = out[k + (size_t)q * m], w = tw[((size_t)q * k * tws)];
				v[q - 1].re = b.re * w.re - b.im * w.im; v[q - 1].im = b.re * w.im + b.im * w.re;
			}
			REAL p1r = v[0].re + v[3].re, p1i = v[0].im + v[3].im, m1r = v[0].re - v[3].re, m1i = v[0].im - v[3].im;
			REAL p2r = v[1].re + v[2].re, p2i = v[1].im + v[2].im, m2r = v[1].re - v[2].re, m2i = v[1].im - v[2].im;
			out[k].re = a.re + p1r + p2r; out[k].im = a.im + p1i + p2i;
			REAL ar = a.re + c1 * p1r + c2 * p2r, ai = a.im + c1 * p1i + c2 * p2i;      /* outputs 1 and 4 */
			REAL br = s1 * m1r + s2 * m2r, bi = s1 * m1i + s2 * m2i;
			out[k + m].re = ar + bi;             out[k + m].im = ai - br;                  /* e^{-2 pi i/5}: -i sin terms */
			out[k + (size_t)4 * m].re = ar - bi; out[k + (size_t)4 * m].im = ai + br;
			REAL cr = a.re + c2 * p1r + c1 * p2r, ci = a.im + c2 * p1i + c1 * p2i;      /* outputs 2 and 3 */
			REAL dr = s2 * m1r - s1 * m2r, di = s2 * m1i - s1 * m2i;
			out[k + (size_t)2 * m].re = cr + di; out[k + (size_t)2 * m].im = ci - dr;
			out[k + (size_t)3 * m].re = cr - di; out[k + (size_t)3 * m].im = ci + dr;
		}
	} else {
		/* generic odd radix: p-point DFT of the twiddled sub-results, O(p^2) */
		CPX t[64];
		CPX *tt = p <= 64 ? t : malloc(sizeof(CPX) * p);
		for (int k = 0; k < m; k++) {
			for (int q = 0; q < p; q++) {
				CPX v = out[k + (size_t)q * m], w = tw[((size_t)q * k * tws) % P->n];
				tt[q].re = v.re * w.re - v.im * w.im; tt[q].im = v.re * w.im + v.im * w.re;
			}
			for (int r = 0; r < p; r++) {
				REAL ar = tt[0].re, ai = tt[0].im;
				for (int q = 1; q < p; q++) {
					CPX w = tw[((size_t)q * r * m * tws) % P->n];   /* w_p^{qr} = w_N^{qr N/p}, N/p = m*tws */
					ar += tt[q].re * w.re - tt[q].im * w.im; ai += tt[q].re * w.im + tt[q].im * w.re;
				}
				out[k + (size_t)r * m].re = ar; out[k + (size_t)r * m].im = ai;
			}
		}
		if (tt != t) free(tt);
	}
}

/* Per-length DCT plan: half-length packing when N is even, full-length otherwise. */
typedef struct {
	int N, M;          /* M = complex FFT length (N/2 or N) */
	int even;
	FFTPLAN fft;
	CPX *t4;           /* t4[k] = exp(-i pi k / 2N), k in [0, N] */
	CPX *t1;           /* t1[k] = exp(-2 pi i k / N), k in [0, N/2] (even only) */
	CPX *a, *b;        /* scratch, length M+1 */
} FN(dctplan);
#define DCTPLAN FN(dctplan)

static void FN(dct_plan_init)(DCTPLAN *p, int N)
{
	p->N = N; p->even = (N % 2 == 0) && N >= 2; p->M = p->even ? N / 2 : N;
	FN(fft_plan_init)(&p->fft, p->M);
	const long double pi = 3.14159265358979323846264338327950288L;
	p->t4 = malloc(sizeof(CPX) * (size_t)(N + 1));
	for (int k = 0; k <= N; k++) { p->t4[k].re = (REAL)cosl(pi * k / (2.0L * N)); p->t4[k].im = (REAL)-sinl(pi * k / (2.0L * N)); }
	p->t1 = malloc(sizeof(CPX) * (size_t)(N / 2 + 1));
	for (int k = 0; k <= N / 2; k++) { p->t1[k].re = (REAL)cosl(2 * pi * k / N); p->t1[k].im = (REAL)-sinl(2 * pi * k / N); }
	p->a = malloc(sizeof(CPX) * (size_t)(p->M + 1));
	p->b = malloc(sizeof(CPX) * (size_t)(p->M + 1));
}
static void FN(dct_plan_free)(DCTPLAN *p) { FN(fft_plan_free)(&p->fft); free(p->t4); free(p->t1); free(p->a); free(p->b); }

/* REDFT10 of x[0], x[xs], ... into y[0], y[ys], ... (in-place safe: x is fully read first). */
static void FN(dct2_1d)(DCTPLAN *p, const REAL *x, ptrdiff_t xs, REAL *y, ptrdiff_t ys)
{
	const int N = p->N, M = p->M;
	CPX *a = p->a, *b = p->b;
	if (N == 1) { y[0] = 2 * x[0]; return; }
	if (p->even) {
		/* v[n]=x[2n], v[N-1-n]=x[2n+1]; z[m] = v[2m] + i v[2m+1] */
		for (int m = 0; m < M; m++) {
			int n0 = 2 * m, n1 = 2 * m + 1;
			a[m].re = x[(ptrdiff_t)(n0 < M ? 2 * n0 : 2 * (N - 1 - n0) + 1) * xs];
			a[m].im = x[(ptrdiff_t)(n1 < M ? 2 * n1 : 2 * (N - 1 - n1) + 1) * xs];
		}
		FN(fft_rec)(&p->fft, a, b, M, 1, 0);
		b[M] = b[0];
		for (int k = 0; k <= M / 2; k++) {
			/* V[k] from Z[k], conj(Z[M-k]); then the quarter-sample twiddle */
			for (int pass = 0; pass < 2; pass++) {
				int kk = pass ? M - k : k;
				if (pass && kk == k) break;
				CPX zk = b[kk], zc = b[M - kk]; zc.im = -zc.im;
				REAL er = (REAL)0.5 * (zk.re + zc.re), ei = (REAL)0.5 * (zk.im + zc.im);
				REAL dr = (REAL)0.5 * (zk.re - zc.re), di = (REAL)0.5 * (zk.im - zc.im);
				/* (d / i) = (di, -dr); times t1[kk] */
				CPX w = p->t1[kk];
				REAL orr = di * w.re + dr * w.im, oi = di * w.im - dr * w.re;
				REAL vr = er + orr, vi = ei + oi;
				CPX t = p->t4[kk];
				REAL wr = vr * t.re - vi * t.im, wi = vr * t.im + vi * t.re;
				y[(ptrdiff_t)kk * ys] = 2 * wr;
				if (kk > 0) y[(ptrdiff_t)(N - kk) * ys] = -2 * wi;
			}
		}
	} else {
		const int h = (N + 1) / 2;
		for (int n = 0; n < h; n++) { a[n].re = x[(ptrdiff_t)(2 * n) * xs]; a[n].im = 0; }
		for (int n = 0; n < N / 2; n++) { a[N - 1 - n].re = x[(ptrdiff_t)(2 * n + 1) * xs]; a[N - 1 - n].im = 0; }
		FN(fft_rec)(&p->fft, a, b, N, 1, 0);
		for (int k = 0; k < N; k++) {
			CPX t = p->t4[k];
			y[(ptrdiff_t)k * ys] = 2 * (b[k].re * t.re - b[k].im * t.im);
		}
	}
}

/* REDFT01 */
static void FN(dct3_1d)(DCTPLAN *p, const REAL *x, ptrdiff_t xs, REAL *y, ptrdiff_t ys)
{
	const int N = p->N, M = p->M;
	CPX *a = p->a, *b = p->b;
	if (N == 1) { y[0] = x[0]; return; }
	if (p->even) {
		/* V[k] = conj(t4[k]) (X[k] - i X[N-k]), k = 0..M, X[N] := 0 */
		for (int k = 0; k <= M; k++) {
			REAL xr = x[(ptrdiff_t)k * xs], xi = k ? -x[(ptrdiff_t)(N - k) * xs] : 0;
			CPX t = p->t4[k];
			b[k].re = xr * t.re + xi * t.im; b[k].im = xi * t.re - xr * t.im;
		}
		/* Z[k] = (V[k] + conj V[M-k]) + i conj(t1[k]) (V[k] - conj V[M-k]); z = conj(FFT(conj Z)) */
		for (int k = 0; k < M; k++) {
			CPX vk = b[k], vc = b[M - k]; vc.im = -vc.im;
			REAL sr = vk.re + vc.re, si = vk.im + vc.im, dr = vk.re - vc.re, di = vk.im - vc.im;
			CPX w = p->t1[k];
			REAL wr = w.re, wi = -w.im;                       /* conj(t1[k]) = exp(+2 pi i k/N) */
			REAL pr = dr * wr - di * wi, pi_ = dr * wi + di * wr;   /* (V - conjV) * e */
			a[k].re = sr - pi_; a[k].im = -(si + pr);          /* conj( s + i p ) */
		}
		FN(fft_rec)(&p->fft, a, b, M, 1, 0);
		for (int m = 0; m < M; m++) {
			REAL v0 = b[m].re, v1 = -b[m].im;
			int n0 = 2 * m, n1 = 2 * m + 1;
			y[(ptrdiff_t)(n0 < M ? 2 * n0 : 2 * (N - 1 - n0) + 1) * ys] = v0;
			y[(ptrdiff_t)(n1 < M ? 2 * n1 : 2 * (N - 1 - n1) + 1) * ys] = v1;
		}
	} else {
		/* v[n] = Re sum_k V[k] e^{+2 pi i k n/N}, V hermitian; use forward FFT of conj */
		for (int k = 0; k < N; k++) {
			REAL xr = x[(ptrdiff_t)k * xs], xi = k ? -x[(ptrdiff_t)(N - k) * xs] : 0;
			CPX t = p->t4[k];
			a[k].re = xr * t.re + xi * t.im; a[k].im = -(xi * t.re - xr * t.im);
		}
		FN(fft_rec)(&p->fft, a, b, N, 1, 0);
		const int h = (N + 1) / 2;
		for (int n = 0; n < h; n++) y[(ptrdiff_t)(2 * n) * ys] = b[n].re;
		for (int n = 0; n < N / 2; n++) y[(ptrdiff_t)(2 * n + 1) * ys] = b[N - 1 - n].re;
	}
}

/*
 * plan_many_r2r + execute semantics (see dct_oracle.c), threads = OpenMP threads to use.
 * Works axis by axis in place on `out` after an initial strided copy in->out.
 */
int FN(cpu_port_r2r_many)(int rank, const int *n, int howmany,
                          const REAL *in, const int *inembed, int istride, int idist,
                          REAL *out, const int *onembed, int ostride, int odist,
                          const int *kinds, int threads)
{
	if (rank < 1 || rank > 3) return -1;
	int dims[3] = {1, 1, 1}, ie[3] = {1, 1, 1}, oe[3] = {1, 1, 1}, kd[3] = {0, 0, 0};
	for (int a = 0; a < rank; a++) {
		int s = 3 - rank + a;
		dims[s] = n[a]; ie[s] = inembed ? inembed[a] : n[a]; oe[s] = onembed ? onembed[a] : n[a]; kd[s] = kinds[a];
		if (kd[s] != 4 && kd[s] != 5) return -1;
	}
	if (threads < 1) threads = 1;
	ptrdiff_t is[3] = {(ptrdiff_t)istride * ie[1] * ie[2], (ptrdiff_t)istride * ie[2], istride};
	ptrdiff_t os[3] = {(ptrdiff_t)ostride * oe[1] * oe[2], (ptrdiff_t)ostride * oe[2], ostride};
	int first = 1;
	for (int ax = 2; ax >= 0; ax--) {
		if (!kd[ax]) continue;
		const int N = dims[ax], o1 = (ax + 1) % 3, o2 = (ax + 2) % 3;
		const REAL *src = first ? in : out;
		const ptrdiff_t *ss = first ? is : os;
		const ptrdiff_t sdist = first ? idist : odist;
		const long per = (long)dims[o1] * dims[o2], lines = (long)howmany * per;
		/* Lines whose first samples lie next to each other in memory are transformed a BLOCK at a time through a transposed scratch
		 * tile, so a strided pass (image columns: 46 KB between samples of one line at 4K) moves whole cache lines instead of one float
		 * per line fetched:
		 *   interleaved images (dist 1, stride = howmany: spec.c:63), any axis: the `run` = howmany (x axis: the channels of one row)
		 *     or dims[2] * howmany (y axis: every column and channel of the image) consecutive lines;
		 *   planar arrays (x stride 1: motion.c:535), axes other than x: run = dims[2] consecutive lines per (batch, other index). */
		long run = 1;               /* lines per memory-consecutive run */
		int mode = 0;               /* 0 line by line, 1 interleaved x axis, 2 interleaved other axis, 3 planar non-x axis */
		if (sdist == 1 && odist == 1 && ss[2] == howmany && os[2] == howmany && howmany > 1) { mode = ax == 2 ? 1 : 2; run = ax == 2 ? howmany : (long)dims[2] * howmany; }
		else if (ax != 2 && ss[2] == 1 && os[2] == 1 && dims[2] > 1) { mode = 3; run = dims[2]; }
		if (mode == 2 && ((ax == 1 && dims[0] != 1) || ax == 0)) { mode = 0; run = 1; }      /* rank-3 interleaved: keep the plain path */
		enum { BL = 16 };
		const long nruns = mode ? lines / run : 0, bpr = mode ? (run + BL - 1) / BL : 0;   /* blocks per run */
		#pragma omp parallel num_threads(threads)
		{
			DCTPLAN P; FN(dct_plan_init)(&P, N);
			REAL *tmp = malloc(sizeof(REAL) * (size_t)N * (mode ? 2 * BL : 1));
			if (!mode) {
				#pragma omp for schedule(static)
				for (long l = 0; l < lines; l++) {
					long t = l / per, r = l % per;
					long p = r / dims[o2], q = r % dims[o2];
					const REAL *x = src + t * sdist + p * ss[o1] + q * ss[o2];
					REAL *y = out + t * odist + p * os[o1] + q * os[o2];
					if (kd[ax] == 5) FN(dct2_1d)(&P, x, ss[ax], tmp, 1); else FN(dct3_1d)(&P, x, ss[ax], tmp, 1);
					for (int k = 0; k < N; k++) y[(ptrdiff_t)k * os[ax]] = tmp[k];
				}
			} else {
				REAL *tin = tmp, *tout = tmp + (size_t)BL * N;
				#pragma omp for schedule(dynamic, 4)
				for (long blk = 0; blk < nruns * bpr; blk++) {
					const long rn = blk / bpr, j0 = (blk % bpr) * BL;
					const int nb = (int)(run - j0 < BL ? run - j0 : BL);
					/* base of the run (its line j sits at base + j) */
					ptrdiff_t bi, bo;
					if (mode == 1) {                               /* run = one row: rn = (y) for rank 2, (z, y) flattened otherwise */
						long p = rn / dims[1], q = rn % dims[1];       /* o1 = 0, o2 = 1 */
						bi = p * ss[0] + q * ss[1]; bo = p * os[0] + q * os[1];
					} else if (mode == 2) { bi = 0; bo = 0; }      /* rank 2: the whole image row direction */
					else {                                         /* planar: rn = (batch t, index of the remaining axis) */
						const int oth = ax == 0 ? 1 : 0;
						long t = rn / dims[oth], p = rn % dims[oth];
						bi = t * sdist + p * ss[oth]; bo = t * odist + p * os[oth];
					}
					const REAL *x = src + bi + j0;
					REAL *y = out + bo + j0;
					for (int k = 0; k < N; k++) { const REAL *xr = x + (ptrdiff_t)k * ss[ax]; for (int b = 0; b < nb; b++) tin[(size_t)b * N + k] = xr[b]; }
					for (int b = 0; b < nb; b++) {
						if (kd[ax] == 5) FN(dct2_1d)(&P, tin + (size_t)b * N, 1, tout + (size_t)b * N, 1);
						else FN(dct3_1d)(&P, tin + (size_t)b * N, 1, tout + (size_t)b * N, 1);
					}
					for (int k = 0; k < N; k++) { REAL *yr = y + (ptrdiff_t)k * os[ax]; for (int b = 0; b < nb; b++) yr[b] = tout[(size_t)b * N + k]; }
				}
			}
			free(tmp); FN(dct_plan_free)(&P);
		}
		first = 0;
	}
	return 0;
}

/*
 * bench.py's all-core CPU leg (`cpu_baseline`; nothing else calls it): `reps` in-place roundtrips REDFT10^2 -> REDFT01^2 / (4wh) of one interleaved
 * h x w x c frame -- the arithmetic of spec.c:63-64 + ispec.c:165-166 through the port's own dct2_1d / dct3_1d -- laid out for many cores:
 *   one thread team for the whole run (plans and scratch per thread, built before the clock starts, as FFTW plans are);
 *   each thread pinned to cpus[t] when a list is given (one logical CPU per physical core) and the frame FIRST TOUCHED by the thread that owns its
 *   rows in the row passes, so that a two-socket box does not serve every thread from the node the caller's thread happens to sit on;
 *   static row ranges / static ranges of 16-column blocks, a barrier between passes.
 * The timed region is the `reps` roundtrips (threads already running, frame resident and touched).  Returns 0; *seconds = wall time of the
 * timed region, *max_err = max |frame - x| after the last roundtrip.
 */
int FN(cpu_port_roundtrip_bench)(int h, int w, int c, const REAL *x, int threads, const int *cpus, int reps, double *seconds, double *max_err)
{
	if (h < 2 || w < 2 || c < 1 || threads < 1 || reps < 1) return -1;
	const size_t len = (size_t)h * w * c;
	REAL *f = malloc(sizeof(REAL) * len);            /* untouched pages: the threads below touch them */
	if (!f) return -2;
	enum { BL = 16 };
	const long row_len = (long)w * c, nblk = (row_len + BL - 1) / BL;
	const REAL scale = (REAL)(1.0 / (4.0 * w * h));
	double t_start = 0, t_end = 0, err = 0;
	int bad = 0;
	#pragma omp parallel num_threads(threads) reduction(max : err) reduction(+ : bad)
	{
#ifdef _OPENMP
		const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
		const int t = 0, nt = 1;
#endif
#if defined(__linux__) && defined(_GNU_SOURCE)
		if (cpus) { cpu_set_t set; CPU_ZERO(&set); CPU_SET(cpus[t], &set); if (sched_setaffinity(0, sizeof set, &set)) bad++; }
#endif
		const long y0 = (long)h * t / nt, y1 = (long)h * (t + 1) / nt;          /* my rows */
		const long b0 = nblk * t / nt, b1 = nblk * (t + 1) / nt;                /* my column blocks */
		for (long y = y0; y < y1; y++) memcpy(f + (size_t)y * row_len, x + (size_t)y * row_len, sizeof(REAL) * (size_t)row_len);
		DCTPLAN PW, PH;
		FN(dct_plan_init)(&PW, w); FN(dct_plan_init)(&PH, h);
		const int nmax = w > h ? w : h;
		REAL *tin = malloc(sizeof(REAL) * (size_t)nmax * 2 * BL), *tout = tin + (size_t)nmax * BL;
		if (!tin) bad++;
		#pragma omp barrier
		if (t == 0) t_start = FN(now_seconds)();
		for (int rep = 0; rep < reps && tin; rep++) {
			for (int dir = 0; dir < 2; dir++) {
				/* forward: rows then columns (FFTW's last-axis-first order does not matter for the result); inverse: columns then rows */
				for (int pass = 0; pass < 2; pass++) {
					const int rows = (pass == 0) == (dir == 0);
					if (rows) {
						for (long y = y0; y < y1; y++) {
							REAL *line = f + (size_t)y * row_len;
							for (int ch = 0; ch < c; ch++) {
								if (dir == 0) FN(dct2_1d)(&PW, line + ch, c, tout, 1); else FN(dct3_1d)(&PW, line + ch, c, tout, 1);
								if (dir == 1) for (int k = 0; k < w; k++) line[(size_t)k * c + ch] = tout[k] * scale;
								else for (int k = 0; k < w; k++) line[(size_t)k * c + ch] = tout[k];
							}
						}
					} else {
						for (long blk = b0; blk < b1; blk++) {
							const long j0 = blk * BL;
							const int nb = (int)(row_len - j0 < BL ? row_len - j0 : BL);
							REAL *col = f + j0;
							for (int k = 0; k < h; k++) { const REAL *xr = col + (size_t)k * row_len; for (int b = 0; b < nb; b++) tin[(size_t)b * h + k] = xr[b]; }
							for (int b = 0; b < nb; b++) {
								if (dir == 0) FN(dct2_1d)(&PH, tin + (size_t)b * h, 1, tout + (size_t)b * h, 1);
								else FN(dct3_1d)(&PH, tin + (size_t)b * h, 1, tout + (size_t)b * h, 1);
							}
							for (int k = 0; k < h; k++) { REAL *yr = col + (size_t)k * row_len; for (int b = 0; b < nb; b++) yr[b] = tout[(size_t)b * h + k]; }
						}
					}
					#pragma omp barrier
				}
			}
		}
		if (t == 0) t_end = FN(now_seconds)();
		for (long y = y0; y < y1; y++)
			for (long i = 0; i < row_len; i++) { const double d = fabs((double)f[(size_t)y * row_len + i] - (double)x[(size_t)y * row_len + i]); if (d > err) err = d; }
		free(tin); FN(dct_plan_free)(&PW); FN(dct_plan_free)(&PH);
	}
	free(f);
	*seconds = t_end - t_start; *max_err = err;
	return bad ? -3 : 0;
}

#undef CPX
#undef FFTPLAN
#undef DCTPLAN
#undef FN
#undef CAT
#undef CAT2
