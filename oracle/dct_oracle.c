/*
 * dct_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * CPU restatement, in f64, of the one arithmetic path dspfun obtains from FFTW: the
 * separable unnormalised real-even transforms REDFT10 (DCT-II) and REDFT01 (DCT-III)
 * executed through fftw(plan_many_r2r)/fftw(plan_r2r_2d) + fftw(execute).
 *
 * The arithmetic itself lives in FFTW3, a third-party system library that is NOT under
 * /root/reference and is not vendored or version-pinned there (Makefiles take whatever
 * `pkg-config fftw3{,f,l}` finds: spec/Makefile:6,10; CI installs distro libfftw3-dev,
 * i.e. FFTW 3.3.x: .github/workflows/build.yml:14,25).  This file therefore restates
 * FFTW's *published definitions* (FFTW 3.3 manual, "1d Real-even DFTs (DCTs)"):
 *     REDFT10:  Y[k] = 2 * sum_{j=0}^{N-1} X[j] cos(pi (j+1/2) k / N)
 *     REDFT01:  Y[k] = X[0] + 2 * sum_{j=1}^{N-1} X[j] cos(pi j (k+1/2) / N)
 * and the "advanced interface" addressing (howmany / stride / dist / embed) the
 * reference call sites rely on:
 *     spec/spec.c:63   ispec.c:165  zoom/zoom.c:263  scan/scan.c:292,359   (rank 2, howmany=c, stride=c, dist=1)
 *     motion/motion.c:535-538,549-552                                     (rank 3, embedded in minbuf, howmany=1)
 *     applybasis/draw.c:74                                                (plan_r2r_2d)
 *
 * Pinning.  FFTW itself was never run (absent here and on every GPU box).  Since round 3 this restatement is pinned to code the
 * REFERENCE holds: its own direct-sum statements of the same transforms -- scan/scan.c:20-41 (generate_basis_matrix + pruned_idct: the
 * 2-D REDFT01), zoom/zoom.c:36-68,361-375 and applybasis/applybasis.c:77-140 (dct2 = REDFT10's kernel, dct3 = REDFT01's) -- compiled as
 * they lie by tests/golden/make_ref_fixtures.py into tests/golden/ref_direct.npz and compared <= 1e-12 in tests/test_ref_direct.py
 * (REDFT01 directly, REDFT10 through the dct2 tables and the 4wh roundtrip identity).  The scipy / pocketfft cross-check of round 1
 * (tests/golden/make_golden.py) stays as a second, independent witness.
 *
 * Everything is computed by direct O(N^2) summation per axis with a long-double cosine
 * table indexed by the exactly reduced integer phase, so the result is the definition to
 * ~1 ulp of double; use it at sizes where O(N^2) finishes in seconds.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>

enum { ORACLE_REDFT01 = 4, ORACLE_REDFT10 = 5 };   /* same numeric values as FFTW's fftw_r2r_kind */

/* cos(pi * t / (2N)) for t in [0, 4N) */
static long double *phase_table(int N) {
	long double *tab = malloc(sizeof(*tab) * 4 * (size_t)N);
	const long double pi = 3.14159265358979323846264338327950288L;
	for (int t = 0; t < 4 * N; t++)
		tab[t] = cosl(pi * (long double)t / (2.0L * N));
	return tab;
}

/* One 1-D transform of length N on a strided vector, out-of-place into a dense temp. */
static void r2r_1d(int kind, int N, const double *x, ptrdiff_t xs, double *y, const long double *tab) {
	const long fourN = 4L * N;
	if (kind == ORACLE_REDFT10) {
		for (int k = 0; k < N; k++) {
			long double acc = 0;
			for (int j = 0; j < N; j++)
				acc += x[j * xs] * tab[((2L * j + 1) * k) % fourN];
			y[k] = (double)(2 * acc);
		}
	} else { /* REDFT01 */
		for (int k = 0; k < N; k++) {
			long double acc = 0;
			for (int j = 1; j < N; j++)
				acc += x[j * xs] * tab[((long)j * (2L * k + 1)) % fourN];
			y[k] = (double)(x[0] + 2 * acc);
		}
	}
}

/*
 * fftw_plan_many_r2r + fftw_execute semantics (FFTW 3.3 manual "Advanced Real-to-real
 * Transforms"): element (i_0..i_{r-1}) of transform t lives at
 *     in [ t*idist + istride * (((i_0*inembed[1] + i_1)*inembed[2] + i_2) ...) ]
 * embed==NULL means embed = n.  in==out is allowed (in-place).
 * Returns 0 on success, -1 on unsupported arguments.
 */
int oracle_r2r_many_f64(int rank, const int *n, int howmany,
                        const double *in, const int *inembed, int istride, int idist,
                        double *out, const int *onembed, int ostride, int odist,
                        const int *kinds)
{
	if (rank < 1 || rank > 3) return -1;
	for (int a = 0; a < rank; a++) {
		if (n[a] < 1) return -1;
		if (kinds[a] != ORACLE_REDFT10 && kinds[a] != ORACLE_REDFT01) return -1;
	}
	int dims[3] = {1, 1, 1}, ie[3] = {1, 1, 1}, oe[3] = {1, 1, 1}, kd[3] = {0, 0, 0};
	/* right-align so that axis 2 is always the fastest one */
	for (int a = 0; a < rank; a++) {
		int s = 3 - rank + a;
		dims[s] = n[a];
		ie[s] = inembed ? inembed[a] : n[a];
		oe[s] = onembed ? onembed[a] : n[a];
		kd[s] = kinds[a];
	}
	size_t total = (size_t)dims[0] * dims[1] * dims[2];
	double *work = malloc(sizeof(double) * total);
	int maxn = dims[0] > dims[1] ? dims[0] : dims[1];
	if (dims[2] > maxn) maxn = dims[2];
	double *line = malloc(sizeof(double) * maxn);
	if (!work || !line) { free(work); free(line); return -1; }

	for (int t = 0; t < howmany; t++) {
		const double *src = in + (ptrdiff_t)t * idist;
		double *dst = out + (ptrdiff_t)t * odist;
		for (int a = 0; a < dims[0]; a++)
			for (int b = 0; b < dims[1]; b++)
				for (int c = 0; c < dims[2]; c++)
					work[((size_t)a * dims[1] + b) * dims[2] + c] =
						src[(ptrdiff_t)istride * (((ptrdiff_t)a * ie[1] + b) * ie[2] + c)];
		ptrdiff_t ws[3] = {(ptrdiff_t)dims[1] * dims[2], dims[2], 1};
		for (int ax = 0; ax < 3; ax++) {
			if (!kd[ax]) continue;
			int N = dims[ax];
			long double *tab = phase_table(N);
			int o1 = (ax + 1) % 3, o2 = (ax + 2) % 3;
			for (int p = 0; p < dims[o1]; p++)
				for (int q = 0; q < dims[o2]; q++) {
					double *base = work + p * ws[o1] + q * ws[o2];
					r2r_1d(kd[ax], N, base, ws[ax], line, tab);
					for (int k = 0; k < N; k++) base[k * ws[ax]] = line[k];
				}
			free(tab);
		}
		for (int a = 0; a < dims[0]; a++)
			for (int b = 0; b < dims[1]; b++)
				for (int c = 0; c < dims[2]; c++)
					dst[(ptrdiff_t)ostride * (((ptrdiff_t)a * oe[1] + b) * oe[2] + c)] =
						work[((size_t)a * dims[1] + b) * dims[2] + c];
	}
	free(work);
	free(line);
	return 0;
}

/* f32 storage convenience: promote, transform in f64, round once to f32. */
int oracle_r2r_many_f32(int rank, const int *n, int howmany,
                        const float *in, const int *inembed, int istride, int idist,
                        float *out, const int *onembed, int ostride, int odist,
                        const int *kinds, size_t in_len, size_t out_len)
{
	double *din = malloc(sizeof(double) * in_len), *dout = malloc(sizeof(double) * out_len);
	if (!din || !dout) { free(din); free(dout); return -1; }
	for (size_t i = 0; i < in_len; i++) din[i] = in[i];
	/* untouched output elements (embedding gaps) must keep their previous contents */
	for (size_t i = 0; i < out_len; i++) dout[i] = out[i];
	int rc = oracle_r2r_many_f64(rank, n, howmany, din, inembed, istride, idist,
	                             dout, onembed, ostride, odist, kinds);
	if (!rc) for (size_t i = 0; i < out_len; i++) out[i] = (float)dout[i];
	free(din); free(dout);
	return rc;
}
