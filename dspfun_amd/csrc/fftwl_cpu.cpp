// fftwl_cpu.cpp -- the fftwl_ (long double) entry points of include/fftw3.h, for tools built with COEFF_PRECISION=L
// (reference include/precision.h:73-79,115: `fftw(call)` becomes fftwl_call).
//
// MI355X has no long double arithmetic, and SURVEY.md 8b allows this precision to be served on the CPU.  This is a HOST-ONLY
// path of the product for that one precision: it is never used by fftwf_ / fftw_ plans (those run on the GPU or fail), and it
// is its own code -- nothing under oracle/ is linked, included or called here.  Algorithm: each axis pass gathers a line,
// reorders it even/odd (Makhoul), runs one complex FFT of the line's length in long double (recursive mixed radix, any
// length: prime factors are done by their definition) and applies the quarter-sample twiddle; REDFT01 runs the same steps backwards.
// Single-threaded: fftwl_plan_with_nthreads is accepted and ignored.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <complex>
#include <map>
#include <vector>

#include "../../include/fftw3.h"

namespace {

typedef long double ld;
typedef std::complex<ld> cld;
const ld kPi = 3.14159265358979323846264338327950288L;

// w[t] = exp(-2 pi i t / n), cached per length
const std::vector<cld> &roots(size_t n)
{
	static std::map<size_t, std::vector<cld>> cache;
	auto it = cache.find(n);
	if (it != cache.end()) return it->second;
	std::vector<cld> w(n);
	for (size_t t = 0; t < n; t++) w[t] = cld(cosl(2 * kPi * t / n), -sinl(2 * kPi * t / n));
	return cache.emplace(n, std::move(w)).first->second;
}

// out[k] = sum_j in[j * stride] exp(-2 pi i j k / n)   (decimation in time over the smallest prime factor)
void fft_rec(const cld *in, size_t stride, cld *out, size_t n, const std::vector<cld> &wN, size_t wstep)
{
	if (n == 1) { out[0] = in[0]; return; }
	size_t p = n;
	for (size_t f = 2; f * f <= n; f++) if (n % f == 0) { p = f; break; }
	const size_t m = n / p;
	std::vector<cld> sub(n);
	for (size_t r = 0; r < p; r++) fft_rec(in + r * stride, stride * p, sub.data() + r * m, m, wN, wstep * p);
	// out[k + q m] = sum_r w_n^{r (k + q m)} sub_r[k]
	for (size_t k = 0; k < m; k++)
		for (size_t q = 0; q < p; q++) {
			const size_t kk = k + q * m;
			cld acc = sub[k];
			for (size_t r = 1; r < p; r++) acc += wN[((r * kk) % n) * wstep] * sub[r * m + k];
			out[kk] = acc;
		}
}

void fft(std::vector<cld> &x)
{
	const size_t n = x.size();
	if (n <= 1) return;
	std::vector<cld> y(n);
	fft_rec(x.data(), 1, y.data(), n, roots(n), 1);
	x.swap(y);
}

// REDFT10: Y[k] = 2 sum_j X[j] cos(pi (j + 1/2) k / N)
void redft10(ld *x, size_t N)
{
	std::vector<cld> v(N);
	for (size_t n = 0; n < (N + 1) / 2; n++) v[n] = x[2 * n];
	for (size_t n = 0; n < N / 2; n++) v[N - 1 - n] = x[2 * n + 1];
	fft(v);
	for (size_t k = 0; k < N; k++) {
		const cld t(cosl(kPi * k / (2 * (ld)N)), -sinl(kPi * k / (2 * (ld)N)));
		x[k] = 2 * (t * v[k]).real();
	}
}

// REDFT01: Y[k] = X[0] + 2 sum_{j >= 1} X[j] cos(pi j (k + 1/2) / N)
void redft01(ld *x, size_t N)
{
	std::vector<cld> v(N);
	v[0] = x[0];
	for (size_t k = 1; k < N; k++) {
		const cld t(cosl(kPi * k / (2 * (ld)N)), sinl(kPi * k / (2 * (ld)N)));
		v[k] = cld(x[k], -x[N - k]) * t;
	}
	// sum_k V[k] exp(+2 pi i k n / N) = conj(FFT(conj V))
	for (auto &c : v) c = std::conj(c);
	fft(v);
	for (size_t n = 0; n < (N + 1) / 2; n++) x[2 * n] = v[n].real();
	for (size_t n = 0; n < N / 2; n++) x[2 * n + 1] = v[N - 1 - n].real();
}

struct PlanL {
	int rank, howmany;
	int n[3], kinds[3];
	long long is[3], os[3], idist, odist;
	long double *in, *out;
};

PlanL *make(int rank, const int *n, int howmany, long double *in, const int *inembed, int istride, int idist,
            long double *out, const int *onembed, int ostride, int odist, const int *kinds)
{
	if (!in || !out || !n || !kinds || rank < 1 || rank > 3 || howmany < 1 || istride < 1 || ostride < 1) {
		fprintf(stderr, "dspfft: fftwl_plan_many_r2r: bad arguments\n");
		return nullptr;
	}
	PlanL *p = new PlanL();
	p->rank = rank; p->howmany = howmany; p->in = in; p->out = out; p->idist = idist; p->odist = odist;
	long long is = istride, os = ostride;
	for (int a = rank - 1; a >= 0; a--) {
		if (n[a] < 1 || (kinds[a] != FFTW_REDFT10 && kinds[a] != FFTW_REDFT01)) {
			fprintf(stderr, "dspfft: fftwl_plan_many_r2r: n[%d] = %d, kind %d unsupported (REDFT10 / REDFT01 only)\n", a, n[a], kinds[a]);
			delete p; return nullptr;
		}
		p->n[a] = n[a]; p->kinds[a] = kinds[a]; p->is[a] = is; p->os[a] = os;
		is *= inembed ? inembed[a] : n[a]; os *= onembed ? onembed[a] : n[a];
	}
	return p;
}

void run(PlanL *p)
{
	if (!p) { fprintf(stderr, "dspfft: fftwl_execute on a NULL plan\n"); return; }
	int n[3] = {1, 1, 1};
	long long is[3] = {0, 0, 0}, os[3] = {0, 0, 0};
	for (int a = 0; a < p->rank; a++) { n[3 - p->rank + a] = p->n[a]; is[3 - p->rank + a] = p->is[a]; os[3 - p->rank + a] = p->os[a]; }
	const size_t total = (size_t)n[0] * n[1] * n[2];
	std::vector<ld> w(total), line;
	for (int b = 0; b < p->howmany; b++) {
		const ld *src = p->in + (long long)b * p->idist;
		ld *dst = p->out + (long long)b * p->odist;
		// the whole logical array into a dense work buffer (input and output layouts may differ, also in place)
		for (int i = 0; i < n[0]; i++) for (int j = 0; j < n[1]; j++) for (int k = 0; k < n[2]; k++)
			w[((size_t)i * n[1] + j) * n[2] + k] = src[i * is[0] + j * is[1] + k * is[2]];
		const size_t ws[3] = {(size_t)n[1] * n[2], (size_t)n[2], 1};
		for (int a = 2; a >= 3 - p->rank; a--) {
			const int N = n[a], kind = p->kinds[a - (3 - p->rank)];
			const int o1 = (a + 1) % 3, o2 = (a + 2) % 3;
			line.resize(N);
			for (int u = 0; u < n[o1]; u++) for (int v = 0; v < n[o2]; v++) {
				ld *base = w.data() + u * ws[o1] + v * ws[o2];
				for (int t = 0; t < N; t++) line[t] = base[t * ws[a]];
				if (kind == FFTW_REDFT10) redft10(line.data(), N); else redft01(line.data(), N);
				for (int t = 0; t < N; t++) base[t * ws[a]] = line[t];
			}
		}
		for (int i = 0; i < n[0]; i++) for (int j = 0; j < n[1]; j++) for (int k = 0; k < n[2]; k++)
			dst[i * os[0] + j * os[1] + k * os[2]] = w[((size_t)i * n[1] + j) * n[2] + k];
	}
}

}  // namespace

extern "C" {

long double *fftwl_alloc_real(size_t n) { void *p = nullptr; return posix_memalign(&p, 64, (n ? n : 1) * sizeof(long double)) ? nullptr : (long double *)p; }
void fftwl_free(void *p) { free(p); }
fftwl_plan fftwl_plan_many_r2r(int rank, const int *n, int howmany, long double *in, const int *inembed, int istride, int idist,
                               long double *out, const int *onembed, int ostride, int odist, const fftwl_r2r_kind *kind, unsigned)
{
	int k[3] = {0, 0, 0};
	if (rank < 1 || rank > 3 || !kind) { fprintf(stderr, "dspfft: rank %d unsupported\n", rank); return nullptr; }
	for (int a = 0; a < rank; a++) k[a] = (int)kind[a];
	return (fftwl_plan)make(rank, n, howmany, in, inembed, istride, idist, out, onembed, ostride, odist, k);
}
fftwl_plan fftwl_plan_r2r_2d(int n0, int n1, long double *in, long double *out, fftwl_r2r_kind k0, fftwl_r2r_kind k1, unsigned)
{
	int n[2] = {n0, n1}, k[2] = {(int)k0, (int)k1};
	return (fftwl_plan)make(2, n, 1, in, nullptr, 1, 0, out, nullptr, 1, 0, k);
}
void fftwl_execute(const fftwl_plan p) { run((PlanL *)p); }
void fftwl_destroy_plan(fftwl_plan p) { delete (PlanL *)p; }
void fftwl_cleanup(void) {}
int fftwl_init_threads(void) { return 1; }
void fftwl_plan_with_nthreads(int) {}
void fftwl_cleanup_threads(void) {}
int fftwl_import_wisdom_from_filename(const char *) { return 0; }
int fftwl_export_wisdom_to_filename(const char *f) { return fftwf_export_wisdom_to_filename(f); }

}  // extern "C"
