// engine.cpp -- plan construction and execution behind include/dspfft.h.
//
// Mirrors the contract of the FFTW calls the reference makes through `fftw(call)`
// (reference include/precision.h:115): plan_many_r2r captures geometry (spec/spec.c:63,
// spec/ispec.c:165, zoom/zoom.c:263, scan/scan.c:292,359, motion/motion.c:535-552), execute runs
// it any number of times (scan/scan.c:447 runs one plan per output frame).  A plan is a list of
// 1-D axis passes over device memory; each pass is one kernel launch (see dct_core.h).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <complex>
#include <mutex>
#include <dlfcn.h>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/dspfft.h"
#include "backend.h"

using namespace dspfft;

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...)
{
	va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
	return code;
}
extern "C" const char *dspfft_last_error(void) { return g_err; }
extern "C" const char *dspfft_version(void) { return "dspfft 0.1 (gfx950)"; }

namespace {

struct Dim { int n; long long is, os; };

const int kRadices[] = {16, 15, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2};

// Smallest number of stages, then smallest sum of radices (less butterfly work / registers).
bool factor_search(int L, int depth, int maxdepth, std::vector<int> &cur, std::vector<int> &best, int &bestsum, int minr)
{
	if (L == 1) {
		int s = 0; for (int r : cur) s += r;
		if (best.empty() || cur.size() < best.size() || (cur.size() == best.size() && s < bestsum)) { best = cur; bestsum = s; }
		return true;
	}
	if (depth == maxdepth) return false;
	bool ok = false;
	for (int r : kRadices) {
		if (r > minr) continue;          // non-increasing order: each multiset visited once
		if (L % r) continue;
		cur.push_back(r);
		ok |= factor_search(L / r, depth + 1, maxdepth, cur, best, bestsum, r);
		cur.pop_back();
	}
	return ok;
}

bool factorize(int L, std::vector<int> &radices)
{
	radices.clear();
	if (L == 1) return true;
	for (int maxdepth = 1; maxdepth <= 8; maxdepth++) {
		std::vector<int> cur, best; int bestsum = 0;
		if (factor_search(L, 0, maxdepth, cur, best, bestsum, 16)) {
			// largest radix last: the last stage needs no twiddles
			std::sort(best.begin(), best.end());
			radices = best;
			return true;
		}
	}
	return false;
}

FastDiv make_div(uint32_t d)
{
	FastDiv f; f.d = d; f.mul = d >= 2 ? (uint32_t)((((uint64_t)1 << 32) + d - 1) / d) : 0; return f;
}

// Tuning overrides for experiments and A/B runs (0 / unset = the built-in choice).  plan_finish() takes a snapshot of the planner switches listed
// here (EnvScope) and everything it calls -- build_pass, build_split, build_block -- consults only that, so the passes of a plan are made under one
// consistent set of values whatever other threads do meanwhile.  What acts OUTSIDE plan_finish reads the environment where it acts: the execute-time
// switches (DSPFFT_LEAN01, DSPFFT_NO_FUSED_ROUNDTRIP, DSPFFT_ROW_CHAN, DSPFFT_RT_SLICE ...), dspfft_plan_scan_prepare (DSPFFT_SCAN_EIDS, DSPFFT_ZSKIP)
// and the plan options set after creation.
const char *const kPlanEnv[] = {"DSPFFT_JIT", "DSPFFT_JIT_TUNE", "DSPFFT_NO_TINY", "DSPFFT_ROW_LPW", "DSPFFT_ROW_THREADS", "DSPFFT_COL_K", "DSPFFT_COL_THREADS",
                                "DSPFFT_DENSE_STAGED", "DSPFFT_NO_SPLIT", "DSPFFT_FORCE_SPLIT", "DSPFFT_NO_BLOCK", "DSPFFT_BLOCK_G", "DSPFFT_NO_BLUESTEIN"};
constexpr int kNPlanEnv = (int)(sizeof kPlanEnv / sizeof kPlanEnv[0]);
thread_local const int *t_plan_env = nullptr;
struct EnvScope {
	int v[kNPlanEnv];
	const int *prev;
	EnvScope() { for (int i = 0; i < kNPlanEnv; i++) { const char *e = getenv(kPlanEnv[i]); v[i] = e ? atoi(e) : 0; } prev = t_plan_env; t_plan_env = v; }
	~EnvScope() { t_plan_env = prev; }
};
int env_int(const char *name)
{
	if (t_plan_env) for (int i = 0; i < kNPlanEnv; i++) if (!strcmp(name, kPlanEnv[i])) return t_plan_env[i];
	const char *e = getenv(name);
	return e ? atoi(e) : 0;
}

struct Tables {
	void *T = nullptr, *W = nullptr, *pos = nullptr, *cosTab = nullptr;
	void *WM = nullptr, *chirp = nullptr, *Bhat = nullptr;     // Bluestein (BLUE passes)
	void *H = nullptr;                                         // half-tile column passes: exp(-2 pi i n / N), n < N/2
	void *stage = nullptr;                                     // DENSE passes on lines longer than LDS holds: [line][N] staging copy
	void release()
	{
		be_free(T); be_free(W); be_free(pos); be_free(cosTab); be_free(WM); be_free(chirp); be_free(Bhat); be_free(H); be_free(stage);
		T = W = pos = cosTab = WM = chirp = Bhat = H = stage = nullptr;
	}
};

bool build_fft(int L, FftDesc &F, std::vector<uint32_t> &pos)
{
	std::vector<int> rad;
	if (!factorize(L, rad)) return false;
	F.L = L; F.ns = (int)rad.size();
	int Lc = L;
	for (int i = 0; i < F.ns; i++) {
		StageDesc &S = F.st[i];
		S.R = rad[i]; S.Lc = Lc; S.M1 = Lc / rad[i]; S.twstep = L / Lc; S.divM1 = make_div((uint32_t)S.M1);
		Lc = S.M1;
	}
	// digit reversal: slot p = sum d_i * (L / (R_1..R_i)) holds output k = d_1 + R_1 (d_2 + R_2 (...))
	pos.assign(L, 0);
	for (int p = 0; p < L; p++) {
		int rem = p, k = 0, mult = 1, span = L;
		for (int i = 0; i < F.ns; i++) {
			span /= rad[i];
			int d = rem / span; rem -= d * span;
			k += d * mult; mult *= rad[i];
		}
		pos[k] = (uint32_t)p;
	}
	return true;
}

struct Pass {
	enum Type { ROW, COL, DENSE, BLUE, TINY } type;   // BLUE = COL geometry, tile DFT by Bluestein's convolution; TINY = one line per thread
	TinyGeom tg;
	int blueM = 0;
	FftDesc fftM;
	int axis;
	bool first;
	PassGeom pa;               // typed pointers, tables and scales are filled in per launch (run_pass)
	DenseGeom da;
	LaunchGeom g;
	bool has_spec = false;     // a compile-time-specialised kernel also covers this pass ...
	PassGeom spa;              // ... with this geometry (used when the buffers are 16-B aligned)
	SpecInfo spec;
	int spec_nwg = 0;
	// outer-radix-2 split of a long column axis (dct_spec.h ColHalfSpec): the row pass works on row pairs, the column pass on half tiles
	bool pair = false, half = false;
	int pair_id = -1;
	SpecInfo hspec;
	PassGeom hpa;
	int half_nwg = 0;
	// a kernel compiled at plan time for this pass's geometry (jit_kernels.h; DSPFFT_JIT=1) where spec_list.h has no entry: spa /
	// spec_nwg describe it like a listed one's, jit_fn is this pass's kind
	bool jit = false;
	void *jit_fn = nullptr;
	void *jit_fn_u8 = nullptr;      // planar rows: the 8-bit variant of this pass's kind
	void *jit_fn_rt = nullptr;      // columns: the fused forward -> filter -> inverse kernel
	std::string jit_type;
	int jit_nthr = 0;
	std::vector<Dim> hostloop;
	Tables tab;
	std::string desc;
};

}  // namespace

struct dspfft_plan_s {
	int rank, howmany;
	int n[3], kinds[3];
	Dim axes[3];
	std::vector<Dim> batches;  // howmany/dist (one entry) or the guru interface's howmany_dims
	bool f64;                  // samples are double (dspfft_plan_many_r2r_f64): double tables; specialised kernels for the sizes of DSPFFT_*_SPECS_F64
	double scale, in0[3], out0[3];
	std::vector<Pass> passes;
	std::vector<Pass> split;   // alternative pass list of dspfft_execute / dspfft_execute_pass (build_split); empty when not applicable
	int split_col_axis = -1;
	// small blocks: all axes in one pass (block_core.h, build_block); used by dspfft_execute on 16-byte aligned buffers
	bool has_block = false;
	BlockGeom blk;
	int blk_kind = 0;
	int blk_axis[3];           // plan axis behind the block's x, y, z
	int blk_nwg = 0;
	size_t blk_lds = 0;
	std::string blk_desc;
	size_t alg_bytes;
	// sparse scan frames (PassGeom::zflags): per-tile flags of the masked column pass + a page of zeros, allocated on first use
	void *zflags = nullptr;
	size_t zflags_bytes = 0;
	void *zpage = nullptr;
	// dspfft_plan_scan_prepare: (min, max) owner id per column tile for the owner-id array `zr_ids` (elements per id zr_div), laid out
	// for the split passes (zr_split) or the plain ones
	void *zranges = nullptr;
	size_t zranges_bytes = 0;
	const void *zr_ids = nullptr;
	int zr_div = 0;
	bool zr_split = false;
	// ... and, for single-precision plans, the owner id of every element in the column tiles' reading order (PassGeom::eids), eid_bytes each
	void *eids = nullptr;
	size_t eids_bytes = 0;
	int eid_bytes = 0;         // 0: not built (the column pass reads zr_ids itself)
	// dspfft_plan_set_input_window: rows of `win_axis` outside [win_lo, win_hi) are zero by contract (first pass, specialised COL REDFT01)
	int win_axis = -1, win_lo = 0, win_hi = 0;
	int alt_axis = -1;         // dspfft_plan_set_output_alternate
	int mod_axis = -1, mod_rev = 0;          // dspfft_plan_set_input_modulation
	const void *mod_table = nullptr;
	bool first_axis_first = false;           // pass order (dspfft_plan_many_r2r_ordered): what a slice plan of this plan is built with
	int col_kpref = 0;                       // column tile width asked of be_find_spec first (slice plans: the narrow tile, see RtSlices)
	// dspfft_execute_roundtrip_u8 over a clip of frames: plans for a slice of the clip whose float intermediate the Infinity Cache holds
	// (roundtrip_core), built on first use and owned by the forward plan
	struct RtSlices { const dspfft_plan_s *inv_of = nullptr; int frames = 0; dspfft_plan_s *fwd = nullptr, *inv = nullptr, *fwd_rem = nullptr, *inv_rem = nullptr; void *side = nullptr, *ev_fork = nullptr, *ev_join = nullptr; };
	std::vector<RtSlices> rt_slices;
	std::mutex rt_mutex;                     // two host threads may run the same plan pair on their own buffers and streams: the slice plans are made once, and
	                                         // a call's fork / slices / join are enqueued as one piece (the library stream and its two events are shared)
};

namespace {

void merge_dims(std::vector<Dim> &d)
{
	std::sort(d.begin(), d.end(), [](const Dim &a, const Dim &b) { return a.os < b.os; });
	for (size_t i = 0; i + 1 < d.size();) {
		if (d[i + 1].is == d[i].n * d[i].is && d[i + 1].os == d[i].n * d[i].os) {
			d[i].n *= d[i + 1].n; d.erase(d.begin() + i + 1);
		} else i++;
	}
}

// T and W in the plan's sample type (computed in long double either way); the pass keeps them as void* in
// P.tab and run_pass hands them to the kernel arguments of the matching type
template <class R>
int upload_tables_t(Pass &P, int N, int L, const std::vector<uint32_t> &pos)
{
	const long double pi = 3.14159265358979323846264338327950288L;
	std::vector<cx<R>> T(N + 1), W(std::max(L, 1));
	for (int j = 0; j <= N; j++) T[j] = cmk<R>((R)cosl(pi * j / (2.0L * N)), (R)-sinl(pi * j / (2.0L * N)));
	for (int t = 0; t < L; t++) W[t] = cmk<R>((R)cosl(2 * pi * t / L), (R)-sinl(2 * pi * t / L));
	P.tab.T = be_alloc(T.size() * sizeof(cx<R>));
	P.tab.W = be_alloc(W.size() * sizeof(cx<R>));
	P.tab.pos = be_alloc(pos.size() * sizeof(uint32_t));
	if (!P.tab.T || !P.tab.W || !P.tab.pos) return -1;
	if (be_upload(P.tab.T, T.data(), T.size() * sizeof(cx<R>))) return -1;
	if (be_upload(P.tab.W, W.data(), W.size() * sizeof(cx<R>))) return -1;
	if (be_upload(P.tab.pos, pos.data(), pos.size() * sizeof(uint32_t))) return -1;
	P.pa.pos = (const uint32_t *)P.tab.pos;
	return 0;
}
int upload_tables(Pass &P, bool f64, int N, int L, const std::vector<uint32_t> &pos)
{
	return f64 ? upload_tables_t<double>(P, N, L, pos) : upload_tables_t<float>(P, N, L, pos);
}

// ---- Bluestein tables --------------------------------------------------------------------------
typedef std::complex<long double> cld;
const long double kPiL = 3.14159265358979323846264338327950288L;

// in-place forward DFT of any length whose prime factors are small (recursive decimation in time; host, plan time)
void host_fft(std::vector<cld> &x)
{
	const size_t n = x.size();
	if (n <= 1) return;
	size_t r = n;
	for (size_t f = 2; f * f <= n; f++) if (n % f == 0) { r = f; break; }
	const size_t m = n / r;
	std::vector<std::vector<cld>> sub(r, std::vector<cld>(m));
	for (size_t j = 0; j < r; j++) for (size_t i = 0; i < m; i++) sub[j][i] = x[i * r + j];
	for (size_t j = 0; j < r; j++) host_fft(sub[j]);
	std::vector<cld> w(n);
	for (size_t t = 0; t < n; t++) w[t] = cld(cosl(2 * kPiL * t / n), -sinl(2 * kPiL * t / n));
	for (size_t k = 0; k < n; k++) {
		cld acc = 0;
		for (size_t j = 0; j < r; j++) acc += w[(j * k) % n] * sub[j][k % m];
		x[k] = acc;
	}
}

// convolution length: 7-smooth M in [lo, 1.3 lo] minimising (stages x M) -- every stage is one pass over the M-row
// LDS buffer -- e.g. lo = 2731: 2880 = 12x15x16 (3 stages) beats 2744 = 7x7x7x8 (4 stages of costlier butterflies).
// DSPFFT_NO_BLUESTEIN=1 (plan time) disables the path (the O(N^2) DENSE kernel takes over; used by its tests).
int blue_length(int lo, FftDesc &F, std::vector<uint32_t> &pos)
{
	if (env_int("DSPFFT_NO_BLUESTEIN") == 1) return 0;
	int best = 0; long long bestcost = 0;
	for (int M = lo; M <= lo + lo / 3 + 16; M++) {
		int r = M;
		for (int p : {2, 3, 5, 7}) while (r % p == 0) r /= p;
		if (r != 1) continue;
		FftDesc f; std::vector<uint32_t> q;
		if (!build_fft(M, f, q)) continue;
		const long long cost = (long long)f.ns * M;
		if (!best || cost < bestcost) { best = M; bestcost = cost; }
	}
	if (best) build_fft(best, F, pos);
	return best;
}

template <class R>
int upload_blue_tables_t(Pass &P, int N, int M, const std::vector<uint32_t> &posM)
{
	std::vector<cld> c(N), b(M, cld(0));
	for (long long n = 0; n < N; n++) {
		const long long r = (n * n) % (2LL * N);                    // exact phase reduction
		c[n] = cld(cosl(kPiL * r / N), sinl(kPiL * r / N));
	}
	b[0] = c[0];
	for (int m = 1; m < N; m++) { b[m] = c[m]; b[M - m] = c[m]; }
	host_fft(b);
	std::vector<cx<R>> T(N + 1), WM(M), chirp(N), Bhat(M);
	std::vector<uint32_t> ident(N);
	for (int j = 0; j <= N; j++) T[j] = cmk<R>((R)cosl(kPiL * j / (2.0L * N)), (R)-sinl(kPiL * j / (2.0L * N)));
	for (int t = 0; t < M; t++) WM[t] = cmk<R>((R)cosl(2 * kPiL * t / M), (R)-sinl(2 * kPiL * t / M));
	for (int n = 0; n < N; n++) { chirp[n] = cmk<R>((R)c[n].real(), (R)c[n].imag()); ident[n] = (uint32_t)n; }
	for (int k = 0; k < M; k++) Bhat[posM[k]] = cmk<R>((R)(b[k].real() / M), (R)(b[k].imag() / M));   // digit-reversed, like the spectrum it multiplies
	struct { void **dst; const void *src; size_t bytes; } up[] = {
		{&P.tab.T, T.data(), T.size() * sizeof(cx<R>)}, {&P.tab.WM, WM.data(), WM.size() * sizeof(cx<R>)},
		{&P.tab.chirp, chirp.data(), chirp.size() * sizeof(cx<R>)}, {&P.tab.Bhat, Bhat.data(), Bhat.size() * sizeof(cx<R>)},
		{&P.tab.pos, ident.data(), ident.size() * sizeof(uint32_t)}};
	for (auto &u : up) {
		*u.dst = be_alloc(u.bytes);
		if (!*u.dst || be_upload(*u.dst, u.src, u.bytes)) return -1;
	}
	P.pa.pos = (const uint32_t *)P.tab.pos;
	return 0;
}

std::string radix_string(const FftDesc &F)
{
	std::string s;
	for (int i = 0; i < F.ns; i++) { if (i) s += "x"; s += std::to_string(F.st[i].R); }
	return s.empty() ? "1" : s;
}

// Build the pass along transformed axis `a`.  `first` => reads the user's input strides.
// ---- plan-time specialisation (DSPFFT_JIT=1) ----
// Radix order and thread count for a length without an entry in spec_list.h, by the rules the listed entries came out of
// (tools/kbench*, colbench, wavebench): largest radix first, an odd radix last when there is one (conflict-free last stage);
// about 11 samples (ROW, at most 512 threads while two lines fit a CU) / 34 samples (COL) of the tile per thread, rounded to a power of two.
std::string jit_radices(const FftDesc &F)
{
	std::vector<int> r;
	for (int i = 0; i < F.ns; i++) r.push_back(F.st[i].R);
	std::sort(r.begin(), r.end(), [](int x, int y) { return x > y; });
	int odd = -1;
	for (int i = (int)r.size() - 1; i >= 0; i--) if (r[i] % 2) { odd = i; break; }
	if (odd >= 0) { const int v = r[odd]; r.erase(r.begin() + odd); r.push_back(v); }
	std::string s;
	for (int v : r) s += ", " + std::to_string(v);
	return s;
}
int jit_threads(double want)
{
	int t = 64;
	while (t < 1024 && (double)t * 1.41 < want) t *= 2;
	return t;
}
std::string library_dir()
{
	Dl_info info;
	if (!dladdr((const void *)&library_dir, &info) || !info.dli_fname) return ".";
	std::string p = info.dli_fname;
	const size_t k = p.rfind('/');
	return k == std::string::npos ? "." : p.substr(0, k);
}
// DSPFFT_JIT=1 forces it on, =2 off; otherwise the planning effort the caller asked for decides (dspfft_set_plan_effort: the FFTW
// shim passes FFTW_MEASURE / PATIENT / EXHAUSTIVE on as "this plan will be executed many times", FFTW_ESTIMATE as "plan fast")
// planning effort of the plans made next: a PROCESS-WIDE value (dspfft_set_plan_effort: what one thread sets, plans made on any thread see -- worker
// pools, Python threads) that a thread may override for itself (dspfft_set_thread_plan_effort; < 0: no override).  The FFTW shim's set / plan /
// restore around one fftw(plan_many_r2r) call (fftw_shim.hip make_plan) uses the thread's override, so it cannot leak into a plan another thread
// is making.  (Round 5 made the one setter thread-local, which silently dropped plans created on other threads to effort 0: ADVICE r05.)
std::atomic<int> g_plan_effort_default{0};
thread_local int t_plan_effort = -1;
#define g_plan_effort (t_plan_effort >= 0 ? t_plan_effort : g_plan_effort_default.load(std::memory_order_relaxed))
bool jit_enabled()
{
	const int e = env_int("DSPFFT_JIT");
	return be_jit_available() && (e == 1 || (e != 2 && g_plan_effort > 0));
}

// Planning effort 2 (FFTW_PATIENT / FFTW_EXHAUSTIVE, DSPFFT_JIT_TUNE=1): compile a handful of candidates -- the rule's choice, the
// planner's own radix order, half and twice the threads, the next narrower column tile -- run each on a scratch buffer laid out like
// the plan's data and keep the fastest.  Every candidate lands in the disk cache; only the timing (milliseconds) is repeated later.
struct JitCand { std::string type; int T, K; };
bool jit_tune() { return env_int("DSPFFT_JIT_TUNE") == 1 || (env_int("DSPFFT_JIT_TUNE") != 2 && g_plan_effort >= 2); }
std::string planner_radices(const FftDesc &F)
{
	std::string s;
	for (int i = 0; i < F.ns; i++) s += ", " + std::to_string(F.st[i].R);
	return s;
}
size_t plan_span(const dspfft_plan_s *pl)
{
	long long span = 1;
	for (int a = 0; a < pl->rank; a++) span += (long long)(pl->n[a] - 1) * std::max(pl->axes[a].is, pl->axes[a].os);
	for (const Dim &b : pl->batches) span += (long long)(b.n - 1) * std::max(b.is, b.os);
	return (size_t)span;
}
template <class R> void fill_args(PassArgsT<R> &a, const PassGeom &g, const dspfft_plan_s *pl, const Pass &P, const R *in, R *out, double scale, const struct Fuse &fz);
template <class R>
float jit_time_t(const dspfft_plan_s *pl, const Pass &P, const PassGeom &g, void *fn, int nwg, int nthr, void *scratch);

// fills P.jit* with the (fastest) candidate that compiles; geometry of candidate i through `geom(i, g, nwg)`
template <class G>
bool jit_pick(const dspfft_plan_s *pl, Pass &P, int is_col, const std::vector<JitCand> &cands, int kind, G geom, JitCand &chosen)
{
	const std::string dir = library_dir();
	const bool tune = jit_tune() && cands.size() > 1;
	void *scratch = nullptr;
	if (tune) {
		const size_t bytes = plan_span(pl) * (pl->f64 ? 8 : 4);
		scratch = be_alloc(bytes);
		if (scratch && be_zero(scratch, bytes, nullptr)) { be_free(scratch); scratch = nullptr; }
	}
	float best = 0.f;
	bool have = false;
	// single precision: columns also get the fused roundtrip, planar rows of a multiple of four samples the 8-bit variants
	const int extras = !pl->f64 && (is_col || (P.pa.C == 1 && P.pa.N % 4 == 0));
	for (size_t i = 0; i < cands.size(); i++) {
		void *fn[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
		char log[2048] = "";
		const int nfn = be_jit_build(cands[i].type.c_str(), is_col, extras, dir.c_str(), fn, log, sizeof log);
		if (nfn < 2) {
			if (i == 0) fprintf(stderr, "dspfft: plan-time compilation of %s failed\n%s\n", cands[i].type.c_str(), log);
			continue;
		}
		PassGeom g = P.pa;
		int nwg = 0;
		geom(cands[i], g, nwg);
		float ms = 0.f;
		if (tune && scratch) {
			ms = pl->f64 ? jit_time_t<double>(pl, P, g, fn[kind], nwg, cands[i].T, scratch) : jit_time_t<float>(pl, P, g, fn[kind], nwg, cands[i].T, scratch);
			if (ms <= 0.f) continue;
		}
		if (!have || (tune && scratch && ms < best)) {
			have = true; best = ms; chosen = cands[i];
			P.jit = true; P.jit_fn = fn[kind]; P.jit_nthr = cands[i].T; P.spa = g; P.spec_nwg = nwg; P.jit_type = cands[i].type;
			P.jit_fn_u8 = (!is_col && nfn == 4) ? fn[2 + kind] : nullptr;
			P.jit_fn_rt = (is_col && nfn == 3) ? fn[2] : nullptr;
		}
		if (!(tune && scratch)) break;
	}
	be_free(scratch);
	return have;
}

// ROW: exact (N, C).  Returns true and fills P.jit* on success.
bool jit_row(const dspfft_plan_s *pl, Pass &P, int N, int C, long long nlines, const FftDesc &F, int kind)
{
	const size_t es = pl->f64 ? 8 : 4;
	const size_t lds = (size_t)C * (N / 2 + 16) * 2 * es;
	if (F.ns < 1 || lds > be_max_lds()) return false;
	// (the timed search of planning effort 2 kept picking twice the threads of the first rule, N C / 22, for RGB rows of 1500-2400
	// pixels and never more than 512 below 4000 pixels)
	int T0 = jit_threads((double)N * C / 11.0);
	if (T0 > 512 && lds <= 80 * 1024) T0 = 512;
	const std::string head = std::string("RowSpecT<") + (pl->f64 ? "double" : "float") + ", " + std::to_string(N) + ", " + std::to_string(C) + ", ";
	std::vector<JitCand> cands;
	auto add = [&](int T, const std::string &rad) {
		if (T < 64 || T > 1024) return;
		JitCand c{head + std::to_string(T) + rad + ">", T, 0};
		for (const JitCand &o : cands) if (o.type == c.type) return;
		cands.push_back(c);
	};
	add(T0, jit_radices(F)); add(T0, planner_radices(F)); add(T0 * 2, jit_radices(F)); add(T0 / 2, jit_radices(F)); add(T0 * 2, planner_radices(F));
	JitCand c;
	if (!jit_pick(pl, P, 0, cands, kind, [&](const JitCand &, PassGeom &, int &nwg) { nwg = (int)nlines; }, c)) return false;
	char buf[320];
	snprintf(buf, sizeof buf, "axis %d: ROW+%s N=%d C=%d compiled at plan time: %s, lines=%lld", P.axis, pl->f64 ? " f64" : "", N, C, c.type.c_str(), nlines);
	P.desc = buf;
	return true;
}
// COL: tile width by the LDS it takes (16 samples up to 80 KB, then 8, then 4; doubles half of that), the inner extent a multiple of it
bool jit_col(const dspfft_plan_s *pl, Pass &P, int N, int inner, const FftDesc &F, int kind)
{
	const size_t es = pl->f64 ? 8 : 4;
	const int Kmax = pl->f64 ? 8 : 16, Kmin = pl->f64 ? 2 : 4;
	int K = Kmax;
	while (K >= Kmin && ((size_t)(N + 16) * K * es > (K == Kmax ? 80u * 1024 : 150u * 1024) || inner % K)) K /= 2;
	if (K < Kmin || F.ns < 1) return false;
	const std::string head = std::string("ColSpecT<") + (pl->f64 ? "double" : "float") + ", " + std::to_string(N) + ", ";
	std::vector<JitCand> cands;
	auto add = [&](int k, int T, const std::string &rad) {
		if (k < Kmin || inner % k || T < 64 || T > 1024 || (size_t)(N + 16) * k * es > 150u * 1024) return;
		JitCand c{head + std::to_string(k) + ", " + std::to_string(T) + rad + ">", T, k};
		for (const JitCand &o : cands) if (o.type == c.type) return;
		cands.push_back(c);
	};
	auto threads = [&](int k) { return jit_threads((double)N * k * (pl->f64 ? 2 : 1) / 34.0); };
	add(K, threads(K), jit_radices(F)); add(K, threads(K), planner_radices(F)); add(K, threads(K) * 2, jit_radices(F)); add(K, threads(K) / 2, jit_radices(F));
	add(K / 2, threads(K / 2), jit_radices(F)); add(K * 2, threads(K * 2), jit_radices(F));
	JitCand c;
	if (!jit_pick(pl, P, 1, cands, kind, [&](const JitCand &cd, PassGeom &g, int &nwg) { g.K = cd.K; g.B = cd.K / 2; g.ntiles = inner / cd.K; nwg = g.ntiles * g.nb0 * g.nb1; }, c)) return false;
	char buf[320];
	snprintf(buf, sizeof buf, "axis %d: COL+%s N=%d K=%d compiled at plan time: %s, tiles=%d wgs=%d", P.axis, pl->f64 ? " f64" : "", N, c.K, c.type.c_str(), P.spa.ntiles, P.spec_nwg);
	P.desc = buf;
	return true;
}

int build_pass(dspfft_plan_s *pl, int a, bool first, Pass &P)
{
	memset(&P.pa, 0, sizeof P.pa); memset(&P.da, 0, sizeof P.da); memset(&P.g, 0, sizeof P.g);
	P.axis = a; P.first = first;
	const int N = pl->n[a];
	const int kind = pl->kinds[a] == DSPFFT_REDFT10 ? KIND_REDFT10 : KIND_REDFT01;
	auto eff = [&](const Dim &d) { Dim e = d; if (!first) e.is = d.os; return e; };
	const Dim ax = eff(pl->axes[a]);
	std::vector<Dim> others;
	for (int b = 0; b < pl->rank; b++) if (b != a && pl->n[b] > 1) others.push_back(eff(pl->axes[b]));
	for (const Dim &b : pl->batches) if (b.n > 1) others.push_back(eff(b));
	const size_t maxlds = be_max_lds();
	const size_t es = pl->f64 ? 8 : 4;         // bytes per sample; a complex LDS slot is 2*es
	const char *tag = pl->f64 ? " f64" : "";
	char buf[256];

	// ---------------- TINY ----------------
	if (N <= 32 && !(env_int("DSPFFT_NO_TINY") == 1)) {
		std::vector<Dim> d = others;
		merge_dims(d);
		long long nlines = 1;
		for (const Dim &x : d) nlines *= x.n;
		if (d.size() <= 6 && nlines < (1ll << 31)) {
			TinyGeom &tg = P.tg;
			memset(&tg, 0, sizeof tg);
			tg.N = N; tg.kind = kind; tg.es_in = ax.is; tg.es_out = ax.os; tg.nd = (int)d.size(); tg.nlines = nlines;
			for (size_t i = 0; i < d.size(); i++) { tg.bn[i] = d[i].n; tg.bis[i] = d[i].is; tg.bos[i] = d[i].os; tg.bdiv[i] = make_div((uint32_t)d[i].n); }
			// contiguous lines packed back to back: element-contiguous global access through LDS chunks
			tg.packed = d.size() >= 1 && ax.is == 1 && ax.os == 1 && d[0].is == N && d[0].os == N;
			tg.chunk_div = make_div((uint32_t)(tg.packed ? (d[0].n + TINY_CHUNK - 1) / TINY_CHUNK : 1));
			P.type = Pass::TINY;
			P.g.nwg = (int)((nlines + 255) / 256); P.g.nthr = 256; P.g.lds_bytes = 0;
			snprintf(buf, sizeof buf, "axis %d: TINY%s N=%d lines=%lld dims=%d%s", a, tag, N, nlines, tg.nd, tg.packed ? " packed" : "");
			P.desc = buf;
			return 0;
		}
	}
	// ---------------- ROW ----------------
	{
		bool ok = (N % 2 == 0) && ax.is == ax.os && ax.is >= 1 && ax.is <= 4;
		int C = (int)ax.is, cdim = -1;
		if (ok && C > 1) {
			ok = false;
			for (size_t i = 0; i < others.size(); i++)
				if (others[i].is == 1 && others[i].os == 1 && others[i].n == C) { cdim = (int)i; ok = true; }
		}
		FftDesc F; std::vector<uint32_t> pos;
		if (ok) ok = build_fft(N / 2, F, pos);
		if (ok) {
			const int L = N / 2;
			// the line's C signals sit in LDS together; longer lines go to the column pass -- unless the line has listed channel-line kernels
			// (one channel per workgroup: 7680 x 3 doubles, spec_list.h)
			if ((size_t)L * C * 2 * es > maxlds) {
				SpecInfo si;
				ok = pl->f64 && C > 1 && be_find_spec_f64(0, N, C, &si) && si.chan;
			}
			if (ok) {
				std::vector<Dim> lines;
				for (size_t i = 0; i < others.size(); i++) if ((int)i != cdim) lines.push_back(others[i]);
				merge_dims(lines);
				PassGeom &pa = P.pa;
				pa.N = N; pa.kind = kind; pa.C = C; pa.fft = F;
				pa.nb0 = lines.size() > 0 ? lines[0].n : 1; pa.sb0_in = lines.size() > 0 ? lines[0].is : 0; pa.sb0_out = lines.size() > 0 ? lines[0].os : 0;
				pa.nb1 = lines.size() > 1 ? lines[1].n : 1; pa.sb1_in = lines.size() > 1 ? lines[1].is : 0; pa.sb1_out = lines.size() > 1 ? lines[1].os : 0;
				const long long nb2 = lines.size() > 2 ? lines[2].n : 1;
				pa.sb2_in = lines.size() > 2 ? lines[2].is : 0; pa.sb2_out = lines.size() > 2 ? lines[2].os : 0;
				for (size_t i = 3; i < lines.size(); i++) P.hostloop.push_back(lines[i]);
				P.type = Pass::ROW;
				const long long nlines = (long long)pa.nb0 * pa.nb1 * nb2;
				pa.nlines = nlines;
				if (nlines > 0x7fffffffLL) return fail(-2, "too many lines for one launch");
				// short lines: several per workgroup (about 2048 samples), as long as the launch keeps well over a workgroup per CU
				int LPW = 1;
				while (LPW < 16 && (long long)N * C * LPW < 2048 && nlines / (2 * LPW) >= 1024 && (size_t)L * C * 2 * LPW * 2 * es <= maxlds) LPW *= 2;
				if (env_int("DSPFFT_ROW_LPW")) LPW = std::min(16, std::max(1, env_int("DSPFFT_ROW_LPW")));
				pa.LPW = LPW; pa.divB = make_div((uint32_t)(C * LPW)); pa.divC = make_div((uint32_t)C);
				pa.divNC = make_div((uint32_t)(N * C)); pa.divKC = make_div((uint32_t)((L / 2 + 1) * C));
				P.g.nwg = (int)((nlines + LPW - 1) / LPW); P.g.lds_bytes = (size_t)L * C * LPW * 2 * es;
				// about 16 waves per CU: one big-LDS workgroup of 1024 threads, two of 512, otherwise 256 (measured, tools/sweep_generic.py)
				P.g.nthr = P.g.lds_bytes > 80 * 1024 ? 1024 : P.g.lds_bytes > 24 * 1024 ? 512 : 256;
				if (env_int("DSPFFT_ROW_THREADS")) P.g.nthr = env_int("DSPFFT_ROW_THREADS");
				if (upload_tables(P, pl->f64, N, L, pos)) return fail(-3, "table upload failed");
				snprintf(buf, sizeof buf, "axis %d: ROW%s  N=%d C=%d fft=%d(%s) lines=%lld x%d lds=%zu", a, tag, N, C, L, radix_string(F).c_str(), nlines, LPW, P.g.lds_bytes);
				P.desc = buf;
				// vector pixel access needs the line starts aligned to the pixel vector (C=2: 8 B, C=4: 16 B)
				const int al = (C == 2 || C == 4) ? C : 1;
				bool aligned = true;
				for (const Dim &d : lines) aligned = aligned && (d.is % al == 0) && (d.os % al == 0);
				if (aligned && (pl->f64 ? be_find_spec_f64(0, N, C, &P.spec) : be_find_spec(0, N, C, &P.spec))) {
					P.has_spec = true; P.spa = pa; P.spec_nwg = (int)nlines;
					if (P.spec.chan) snprintf(buf, sizeof buf, "axis %d: ROW*%s N=%d C=%d spec#%d as %d channel lines, threads=%d lines=%lld lds=%zu", a, tag, N, C, P.spec.id, P.spec.chan, P.spec.nthr, nlines, P.spec.lds);
					else snprintf(buf, sizeof buf, "axis %d: ROW*%s N=%d C=%d spec#%d threads=%d lines=%lld lds=%zu", a, tag, N, C, P.spec.id, P.spec.nthr, nlines, P.spec.lds);
					P.desc = buf;
				} else if (aligned && jit_enabled() && lines.size() <= 2) {
					jit_row(pl, P, N, C, nlines, F, kind);
				}
				return 0;
			}
		}
	}
	// ---------------- COL (and its Bluestein variant) ----------------
	{
		FftDesc F, FM; std::vector<uint32_t> pos, posM;
		const bool direct = build_fft(N, F, pos);
		const int M = direct ? 0 : blue_length(2 * N - 1, FM, posM);
		bool ok = direct || M > 0;
		int idim = -1;
		for (size_t i = 0; i < others.size(); i++) if (others[i].is == 1 && others[i].os == 1) idim = (int)i;
		if (ok) {
			// merge dims that continue the inner run contiguously (dense embeddings)
			std::vector<Dim> rest;
			// no unit-stride dimension beside the axis (planar lines): a one-sample-wide "inner" run, i.e. tiles of one
			// real signal whose partner column is empty
			Dim inner = idim >= 0 ? others[idim] : Dim{1, 1, 1};
			for (size_t i = 0; i < others.size(); i++) if ((int)i != idim) rest.push_back(others[i]);
			std::sort(rest.begin(), rest.end(), [](const Dim &x, const Dim &y) { return x.os < y.os; });
			for (size_t i = 0; idim >= 0 && i < rest.size();) {
				if (rest[i].is == inner.n && rest[i].os == inner.n && (long long)inner.n * rest[i].n < (1 << 30)) { inner.n *= rest[i].n; rest.erase(rest.begin() + i); i = 0; }
				else i++;
			}
			merge_dims(rest);
			const size_t rows = direct ? (size_t)N : (size_t)M;           // complex LDS rows per tile column pair
			// tile width: 16 floats (64 B row segments); short axes take wider tiles so a workgroup still owns a few
			// thousand samples (a 64-row tile of width 16 is 1024 samples for 256 threads)
			int Kwant = 16;
			long long nbatch = 1;
			for (const Dim &d : rest) nbatch *= d.n;
			// ... as long as the launch still has well over a workgroup per CU (a single small frame keeps its narrow tiles)
			while (Kwant < 64 && (long long)rows * Kwant < 4096 && ((inner.n + 2 * Kwant - 1) / (2 * Kwant)) * nbatch >= 1024) Kwant *= 2;
			int K = std::min(env_int("DSPFFT_COL_K") ? env_int("DSPFFT_COL_K") : Kwant, (inner.n + 1) & ~1);
			while (K >= 2 && rows * (K / 2) * 2 * es > maxlds) K -= 2;
			if (K >= 2) {
				PassGeom &pa = P.pa;
				pa.N = N; pa.kind = kind; pa.K = K; pa.B = K / 2; pa.ninner = inner.n; pa.ntiles = (inner.n + K - 1) / K;
				pa.es_in = ax.is; pa.es_out = ax.os; pa.fft = F; pa.divB = make_div((uint32_t)pa.B);
				pa.nb0 = rest.size() > 0 ? rest[0].n : 1; pa.sb0_in = rest.size() > 0 ? rest[0].is : 0; pa.sb0_out = rest.size() > 0 ? rest[0].os : 0;
				pa.nb1 = rest.size() > 1 ? rest[1].n : 1; pa.sb1_in = rest.size() > 1 ? rest[1].is : 0; pa.sb1_out = rest.size() > 1 ? rest[1].os : 0;
				for (size_t i = 2; i < rest.size(); i++) P.hostloop.push_back(rest[i]);
				long long nwg = (long long)pa.ntiles * pa.nb0 * pa.nb1;
				if (nwg > 0x7fffffff) return fail(-2, "too many tiles for one launch");
				P.g.nwg = (int)nwg;
				P.g.lds_bytes = rows * pa.B * 2 * es;
				const long long slots = (long long)rows * pa.B;
				P.g.nthr = (slots >= 12000 || P.g.lds_bytes > 80 * 1024) ? 1024 : slots >= 4096 ? 512 : 256;   // measured, tools/sweep_generic.py
				if (env_int("DSPFFT_COL_THREADS")) P.g.nthr = env_int("DSPFFT_COL_THREADS");
				if (!direct) {
					P.type = Pass::BLUE; P.blueM = M; P.fftM = FM;
					memset(&pa.fft, 0, sizeof pa.fft);
					if (pl->f64 ? upload_blue_tables_t<double>(P, N, M, posM) : upload_blue_tables_t<float>(P, N, M, posM)) return fail(-3, "table upload failed");
					snprintf(buf, sizeof buf, "axis %d: BLUE%s N=%d K=%d inner=%d tiles=%d conv=%d(%s) wgs=%d lds=%zu", a, tag, N, K, inner.n, pa.ntiles, M, radix_string(FM).c_str(), P.g.nwg, P.g.lds_bytes);
					P.desc = buf;
					return 0;
				}
				P.type = Pass::COL;
				if (upload_tables(P, pl->f64, N, N, pos)) return fail(-3, "table upload failed");
				snprintf(buf, sizeof buf, "axis %d: COL%s  N=%d K=%d inner=%d tiles=%d fft=%d(%s) wgs=%d lds=%zu", a, tag, N, K, inner.n, pa.ntiles, N, radix_string(F).c_str(), P.g.nwg, P.g.lds_bytes);
				P.desc = buf;
				// float4 tile rows: every stride that positions a tile must keep 16-B alignment
				bool aligned = (ax.is % 4 == 0) && (ax.os % 4 == 0);
				for (const Dim &d : rest) aligned = aligned && (d.is % 4 == 0) && (d.os % 4 == 0);
				if (aligned && (pl->f64 ? be_find_spec_f64(1, N, inner.n, &P.spec) : be_find_spec(1, N, inner.n, &P.spec, pl->col_kpref))) {
					P.has_spec = true; P.spa = pa;
					P.spa.K = P.spec.P; P.spa.B = P.spec.P / 2; P.spa.ntiles = inner.n / P.spec.P;
					P.spec_nwg = P.spa.ntiles * pa.nb0 * pa.nb1;
					snprintf(buf, sizeof buf, "axis %d: COL*%s N=%d K=%d spec#%d threads=%d inner=%d tiles=%d wgs=%d lds=%zu (generic fallback: K=%d)", a, tag, N, P.spec.P, P.spec.id, P.spec.nthr, inner.n, P.spa.ntiles, P.spec_nwg, P.spec.lds, K);
					P.desc = buf;
				} else if (aligned && jit_enabled() && rest.size() <= 2) {
					jit_col(pl, P, N, inner.n, F, kind);
				}
				return 0;
			}
		}
	}
	// ---------------- DENSE ----------------
	{
		// a line longer than LDS holds (beyond Bluestein's range too: a prime factor > 13 in a length of tens of thousands) is staged in a
		// device array of the plan instead: the reference never checks for a NULL plan (spec/spec.c:63-64), so every length gets one
		const bool staged = (size_t)N * es > maxlds || env_int("DSPFFT_DENSE_STAGED") == 1;
		merge_dims(others);
		DenseGeom &da = P.da;
		da.N = N; da.kind = kind; da.es_in = ax.is; da.es_out = ax.os;
		da.nb0 = others.size() > 0 ? others[0].n : 1; da.sb0_in = others.size() > 0 ? others[0].is : 0; da.sb0_out = others.size() > 0 ? others[0].os : 0;
		da.nb1 = others.size() > 1 ? others[1].n : 1; da.sb1_in = others.size() > 1 ? others[1].is : 0; da.sb1_out = others.size() > 1 ? others[1].os : 0;
		da.nb2 = others.size() > 2 ? others[2].n : 1; da.sb2_in = others.size() > 2 ? others[2].is : 0; da.sb2_out = others.size() > 2 ? others[2].os : 0;
		for (size_t i = 3; i < others.size(); i++) P.hostloop.push_back(others[i]);
		const long double pi = 3.14159265358979323846264338327950288L;
		P.tab.cosTab = be_alloc((size_t)4 * N * es);
		if (!P.tab.cosTab) return fail(-3, "table upload failed");
		int urc;
		if (pl->f64) {
			std::vector<double> ct((size_t)4 * N);
			for (int t = 0; t < 4 * N; t++) ct[t] = (double)cosl(pi * t / (2.0L * N));
			urc = be_upload(P.tab.cosTab, ct.data(), ct.size() * 8);
		} else {
			std::vector<float> ct((size_t)4 * N);
			for (int t = 0; t < 4 * N; t++) ct[t] = (float)cosl(pi * t / (2.0L * N));
			urc = be_upload(P.tab.cosTab, ct.data(), ct.size() * 4);
		}
		if (urc) return fail(-3, "table upload failed");
		P.type = Pass::DENSE;
		long long lines = (long long)da.nb0 * da.nb1 * da.nb2;
		if (lines > 0x7fffffff) return fail(-2, "too many lines for the dense path");
		P.g.nwg = (int)lines; P.g.nthr = 256; P.g.lds_bytes = staged ? 0 : (size_t)N * es;
		if (staged) {
			// one staging array serves every host-loop iteration (they run one after the other); executions of this plan on two streams
			// at once would share it: dspfft_execute_many[_repeat] refuses that (staged_plan_on_two_streams)
			P.tab.stage = be_alloc((size_t)lines * N * es);
			if (!P.tab.stage) return fail(-3, "axis %d: no memory for the staging copy of %lld lines of %d samples", a, lines, N);
		}
		snprintf(buf, sizeof buf, "axis %d: DENSE%s N=%d lines=%lld lds=%zu%s", a, tag, N, lines, P.g.lds_bytes, staged ? " (lines staged in device memory)" : "");
		P.desc = buf;
		return 0;
	}
}

struct Fuse { const void *eids = nullptr; int eid_bytes = 0; const uint32_t *mask = nullptr; uint32_t id = 0; int div = 1; bool accumulate = false; uint8_t *zflags = nullptr; int zshift = 0, zhalf = 0; const void *zpage = nullptr; const uint32_t *zranges = nullptr; };
FastDiv make_div(uint32_t d);

template <class R>
void fill_args(PassArgsT<R> &a, const PassGeom &g, const dspfft_plan_s *pl, const Pass &P, const R *in, R *out, double scale, const Fuse &fz)
{
	static_cast<PassGeom &>(a) = g;
	a.in = in; a.out = out; a.T = (const cx<R> *)P.tab.T; a.W = (const cx<R> *)P.tab.W; a.H = (const cx<R> *)P.tab.H;
	a.scale = (R)scale; a.in_scale0 = (R)pl->in0[P.axis]; a.out_scale0 = (R)pl->out0[P.axis];
	a.mask = fz.mask; a.mask_id = fz.id; a.mask_div = make_div((uint32_t)fz.div); a.accumulate = fz.accumulate;
	a.zflags = fz.zflags; a.zshift = fz.zshift; a.zhalf = fz.zhalf; a.zpage = fz.zpage; a.zranges = fz.zranges;
	// masked column tiles take their owner ids from the plan's element-order table where one was prepared for this id array and the frame id fits it
	a.mask_mode = 0; a.eids = nullptr;
	if (fz.mask && fz.zpage && fz.eids && fz.eid_bytes && fz.id < (fz.eid_bytes == 1 ? 0xffu : 0xffffu)) { a.mask_mode = fz.eid_bytes; a.eids = fz.eids; }
	a.alt_out = (pl->alt_axis == P.axis && pl->alt_axis >= 0) ? 1 : 0;
	static const int lean_off = []() { const char *e = getenv("DSPFFT_LEAN01"); return e && *e == '0' ? 1 : 0; }();
	a.lean_off = lean_off;
	a.win_lo = a.win_hi = 0;
	if (pl->win_axis == P.axis && P.first && !fz.mask && pl->zpage) { a.win_lo = pl->win_lo; a.win_hi = pl->win_hi; a.zpage = pl->zpage; }
	a.in_mul = nullptr; a.in_rev = 0;
	if (pl->mod_axis == P.axis && P.first && !fz.mask && a.win_hi > 0) { a.in_mul = pl->mod_table; a.in_rev = pl->mod_rev; }
}

// one candidate of the planning-effort-2 search: mean of three launches on the scratch buffer, in milliseconds (<= 0: failed)
template <class R>
float jit_time_t(const dspfft_plan_s *pl, const Pass &P, const PassGeom &g, void *fn, int nwg, int nthr, void *scratch)
{
	PassArgsT<R> a;
	fill_args(a, g, pl, P, (const R *)scratch, (R *)scratch, 1.0, Fuse());
	void *e0 = be_event_create(), *e1 = be_event_create();
	float ms = -1.f;
	if (e0 && e1 && !be_jit_launch(fn, &a, nwg, nthr, nullptr) && !be_event_record(e0, nullptr)) {
		bool ok = true;
		for (int i = 0; i < 3 && ok; i++) ok = !be_jit_launch(fn, &a, nwg, nthr, nullptr);
		if (ok && !be_event_record(e1, nullptr) && !be_event_synchronize(e1) && !be_event_elapsed_ms(e0, e1, &ms)) ms /= 3.f; else ms = -1.f;
	}
	be_event_destroy(e0); be_event_destroy(e1);
	return ms;
}
template float jit_time_t<float>(const dspfft_plan_s *, const Pass &, const PassGeom &, void *, int, int, void *);
template float jit_time_t<double>(const dspfft_plan_s *, const Pass &, const PassGeom &, void *, int, int, void *);

template <class R>
int run_pass(const dspfft_plan_s *pl, const Pass &P, const R *in, R *out, bool last, void *stream, const Fuse &fz = Fuse())
{
	const double scale = last ? pl->scale : 1.0;
	// iterate the host-side batch dims (rare: more batch levels than a kernel takes)
	std::vector<int> idx(P.hostloop.size(), 0);
	for (;;) {
		long long oin = 0, oout = 0;
		for (size_t i = 0; i < idx.size(); i++) { oin += idx[i] * P.hostloop[i].is; oout += idx[i] * P.hostloop[i].os; }
		int rc = 0;
		if (P.type == Pass::TINY) {
			TinyArgsT<R> a;
			static_cast<TinyGeom &>(a) = P.tg;
			a.in = in + oin; a.out = out + oout;
			a.scale = (R)scale; a.in_scale0 = (R)pl->in0[P.axis]; a.out_scale0 = (R)pl->out0[P.axis];
			a.mask = fz.mask; a.mask_id = fz.id; a.mask_div = make_div((uint32_t)fz.div); a.accumulate = fz.accumulate;
			rc = be_launch_tiny(a, stream);
		} else if (P.type == Pass::DENSE) {
			DenseArgsT<R> a;
			static_cast<DenseGeom &>(a) = P.da;
			a.in = in + oin; a.out = out + oout; a.cosTab = (const R *)P.tab.cosTab; a.stage = (R *)P.tab.stage;
			a.scale = (R)scale; a.in_scale0 = (R)pl->in0[P.axis]; a.out_scale0 = (R)pl->out0[P.axis];
			a.mask = fz.mask; a.mask_id = fz.id; a.mask_div = make_div((uint32_t)fz.div); a.accumulate = fz.accumulate;
			rc = be_launch_dense(a, P.g, stream);
		} else {
			bool use_spec = false;
			if constexpr (std::is_same<R, float>::value) {
				if (P.pair || P.half) {          // only reached through pick_passes(), which has checked alignment and scales
					PassArgs a;
					fill_args(a, P.pair ? P.spa : P.hpa, pl, P, in + oin, out + oout, scale, fz);
					rc = P.pair ? be_launch_row_pair(P.pair_id, a, P.spec_nwg / 2, stream) : be_launch_col_half(P.hspec.id, a, P.half_nwg, stream);
					if (rc) return fail(-4, "kernel launch failed (%s): backend code %d", P.desc.c_str(), rc);
					return 0;
				}
			}
			{
				// the specialised kernels move whole pixels (ROW: C samples) or four samples (COL) per access
				const uintptr_t al = sizeof(R) * (P.type == Pass::ROW ? (P.pa.C == 2 || P.pa.C == 4 ? P.pa.C : 1) : 4);
				const bool ptr_ok = ((al - 1) & ((uintptr_t)(in + oin) | (uintptr_t)(out + oout))) == 0;
				use_spec = (P.has_spec || P.jit) && ptr_ok;
				// the plan options that let a caller leave zeros unstored (input window), or that change values (modulation, alternating output
				// sign), live in the listed kernels only: a run that cannot take them must fail, not fall through to a kernel that ignores them
				const bool windowed = pl->win_axis == P.axis && P.first && pl->win_hi > 0;
				const bool alternating = pl->alt_axis >= 0 && pl->alt_axis == P.axis;
				if ((windowed || alternating) && !(P.has_spec && !P.jit && ptr_ok))
					return fail(-1, "%s: the plan's input window / modulation / alternating output need 16-byte aligned buffers (in %p, out %p): the listed kernel cannot run and no other honours them",
					            P.desc.c_str(), (const void *)(in + oin), (const void *)(out + oout));
				if (windowed && fz.mask)
					return fail(-1, "%s: an owner-id mask cannot be combined with an input window (the masked loads would read the rows the window leaves unstored)", P.desc.c_str());
				if (use_spec) {
					PassArgsT<R> a;
					fill_args(a, P.spa, pl, P, in + oin, out + oout, scale, fz);
					rc = P.jit ? be_jit_launch(P.jit_fn, &a, P.spec_nwg, P.jit_nthr, stream) : be_launch_spec(P.type == Pass::COL, P.spec.id, a, P.spec_nwg, stream);
				}
			}
			if (P.type == Pass::BLUE) {
				BlueArgsT<R> a;
				fill_args(a, P.pa, pl, P, in + oin, out + oout, scale, fz);
				a.M = P.blueM; a.fftM = P.fftM;
				a.WM = (const cx<R> *)P.tab.WM; a.chirp = (const cx<R> *)P.tab.chirp; a.Bhat = (const cx<R> *)P.tab.Bhat;
				rc = be_launch_blue(a, P.g, stream);
			} else if (!use_spec) {
				PassArgsT<R> a;
				fill_args(a, P.pa, pl, P, in + oin, out + oout, scale, fz);
				rc = P.type == Pass::ROW ? be_launch_row(a, P.g, stream) : be_launch_col(a, P.g, stream);
			}
		}
		if (rc) return fail(-4, "kernel launch failed (%s): backend code %d", P.desc.c_str(), rc);
		size_t i = 0;
		for (; i < idx.size(); i++) { if (++idx[i] < P.hostloop[i].n) break; idx[i] = 0; }
		if (i == idx.size()) break;
	}
	return 0;
}

}  // namespace

namespace {
// Outer radix-2 split of a long column axis for 2-D f32 plans whose two passes both have specialised kernels: the row pass
// butterflies row pairs on its input, the column pass transforms half tiles of N/2 rows and twice the width (dct_spec.h,
// ColHalfSpec).  REDFT10 along the columns runs row pass -> column pass (decimation in frequency), REDFT01 column pass ->
// row pass (decimation in time), whatever order the plain plan uses.  DSPFFT_NO_SPLIT=1 disables it; DSPFFT_FORCE_SPLIT=1
// applies it wherever the kernels exist (tests on small frames).
void build_split(dspfft_plan_s *pl)
{
	if (pl->f64 || pl->rank != 2 || pl->passes.size() != 2 || env_int("DSPFFT_NO_SPLIT") == 1) return;
	const Pass *R = nullptr, *Cc = nullptr;
	for (const Pass &P : pl->passes) {
		if (!P.has_spec || !P.hostloop.empty() || P.pa.sb2_in || P.pa.sb2_out) return;
		if (P.type == Pass::ROW) R = &P; else if (P.type == Pass::COL) Cc = &P;
	}
	if (!R || !Cc) return;
	const int N = Cc->pa.N, ca = Cc->axis, ra = R->axis;
	const bool force = env_int("DSPFFT_FORCE_SPLIT") == 1;
	if (N % 4) return;
	SpecInfo hs;
	if (!be_find_half_spec(N, Cc->pa.ninner, &hs)) return;
	// Used when the full-length tile fills a CU's LDS on its own (8K: 4320 rows x 8 floats = 138 KB, one workgroup per CU) and the
	// half tiles fit twice: measured 935 vs 1027 us per 7680x4320x3 roundtrip.  At 3840x2160 the half tiles are wider instead
	// (1080 x 16 floats, 64-B row segments: column passes 42-45 us instead of 46-49) but the paired row pass loses more than that
	// (2 rows per workgroup in sequence, 2 workgroups per CU: 52-55 us instead of 37-39) -- 58.3K vs 58.6K Mpix/s on two streams,
	// 41.4K vs 48.9K on one (tools/sbench.hip, tools/bench_8k_split.py; profiles/r02_split.txt) -- so 4K keeps the plain passes.
	// The half tile may also take the whole LDS again with twice the width (4320: 2160 rows x 16 floats): an 8K frame does not fit the
	// Infinity Cache, and HBM serves 64-byte row segments so much better than 32-byte ones that one workgroup per CU wins (882 vs 922 us).
	if (!force && !(Cc->spec.lds > 80 * 1024 && hs.lds <= Cc->spec.lds + 4096)) return;
	const int pid = be_find_row_pair(R->pa.N, R->pa.C);
	if (pid < 0) return;
	const bool col_first = pl->kinds[ca] == DSPFFT_REDFT01;
	std::vector<Pass> sp(2);
	Pass &PR = sp[col_first ? 1 : 0], &PC = sp[col_first ? 0 : 1];
	auto drop = [&]() { for (Pass &P : sp) P.tab.release(); };
	if (build_pass(pl, ra, !col_first, PR) || build_pass(pl, ca, col_first, PC)) { drop(); return; }
	// the row pass's first batch dimension must be the split axis, line for line
	const long long cis = col_first ? pl->axes[ca].os : pl->axes[ca].is, cos_ = pl->axes[ca].os;
	if (PR.type != Pass::ROW || !PR.has_spec || !PR.hostloop.empty() || PR.spa.nb0 != N || PR.spa.sb0_in != cis || PR.spa.sb0_out != cos_ ||
	    PC.type != Pass::COL || !PC.has_spec || !PC.hostloop.empty()) { drop(); return; }
	char buf[256];
	PR.pair = true; PR.pair_id = pid;
	snprintf(buf, sizeof buf, "axis %d: ROW*2 N=%d C=%d row pairs of axis %d, spec#%d threads=%d pairs=%d lds=%zu", ra, PR.pa.N, PR.pa.C, ca, PR.spec.id, be_row_pair_threads(pid), PR.spec_nwg / 2, PR.spec.lds);
	PR.desc = buf;
	// column pass: FFT of length N/2; T stays the table of the full length, W becomes the half length's, H = exp(-2 pi i n / N)
	const int M = N / 2;
	{
		const long double pi = 3.14159265358979323846264338327950288L;
		std::vector<cf> W(M), H(M);
		for (int t = 0; t < M; t++) { W[t] = cmk<float>((float)cosl(2 * pi * t / M), (float)-sinl(2 * pi * t / M)); H[t] = cmk<float>((float)cosl(2 * pi * t / N), (float)-sinl(2 * pi * t / N)); }
		be_free(PC.tab.W);
		PC.tab.W = be_alloc(W.size() * sizeof(cf)); PC.tab.H = be_alloc(H.size() * sizeof(cf));
		if (!PC.tab.W || !PC.tab.H || be_upload(PC.tab.W, W.data(), W.size() * sizeof(cf)) || be_upload(PC.tab.H, H.data(), H.size() * sizeof(cf))) { drop(); return; }
	}
	PC.half = true; PC.hspec = hs; PC.hpa = PC.spa;
	PC.hpa.K = hs.P; PC.hpa.B = hs.P / 2; PC.hpa.ntiles = PC.pa.ninner / hs.P;
	PC.half_nwg = 2 * PC.hpa.ntiles * PC.pa.nb0 * PC.pa.nb1;
	snprintf(buf, sizeof buf, "axis %d: COL*/2 N=%d as 2 x %d, K=%d half#%d threads=%d inner=%d tiles=%d wgs=%d lds=%zu", ca, N, M, hs.P, hs.id, hs.nthr, PC.pa.ninner, 2 * PC.hpa.ntiles, PC.half_nwg, hs.lds);
	PC.desc = buf;
	pl->split = std::move(sp);
	pl->split_col_axis = ca;
}

// which pass list an execution uses: the split one unless the buffers or the plan's per-index scales rule it out
// Blocks of 4, 8 or 16 samples a side (motion's --blocksize 8x8x8 and the like): one pass over the data instead of one per axis
// (block_core.h).  Needs the x axis contiguous, a batch dimension to take a workgroup's blocks from (the blocks of a row of blocks in
// a [D][H][W] volume, or whole blocks of a block-major stack) and 16-byte aligned strides.
void build_block(dspfft_plan_s *pl)
{
	if (pl->f64 || pl->rank < 2 || env_int("DSPFFT_NO_BLOCK") == 1) return;
	for (const Pass &P : pl->passes) if (P.type != Pass::TINY || !P.hostloop.empty()) return;
	for (int a = 1; a < pl->rank; a++) if (pl->kinds[a] != pl->kinds[0]) return;
	int ax = -1;
	for (int a = 0; a < pl->rank; a++) if (pl->axes[a].is == 1 && pl->axes[a].os == 1) ax = a;
	if (ax < 0) return;
	std::vector<int> rest;
	for (int a = 0; a < pl->rank; a++) if (a != ax) rest.push_back(a);
	std::sort(rest.begin(), rest.end(), [&](int p, int q) { return pl->axes[p].is < pl->axes[q].is; });
	const int ay = rest[0], az = rest.size() > 1 ? rest[1] : -1;
	const int nx = pl->n[ax], ny = pl->n[ay], nz = az >= 0 ? pl->n[az] : 1;
	if (!be_block_supported(nx, ny, nz)) return;
	// the dimension a workgroup takes its G blocks from: the batch dimension of smallest stride
	std::vector<Dim> dims;
	for (const Dim &b : pl->batches) if (b.n > 1) dims.push_back(b);
	if (dims.empty()) return;
	size_t gi = 0;
	for (size_t i = 1; i < dims.size(); i++) if (dims[i].is < dims[gi].is) gi = i;
	const Dim grp = dims[gi];
	dims.erase(dims.begin() + gi);
	merge_dims(dims);
	if (dims.size() > BLOCK_MAX_DIMS) return;
	bool aligned = grp.is % 4 == 0 && grp.os % 4 == 0 && pl->axes[ay].is % 4 == 0 && pl->axes[ay].os % 4 == 0 &&
	               (az < 0 || (pl->axes[az].is % 4 == 0 && pl->axes[az].os % 4 == 0));
	long long nrest = 1;
	for (const Dim &d : dims) { aligned = aligned && d.is % 4 == 0 && d.os % 4 == 0; nrest *= d.n; }
	if (!aligned) return;
	BlockGeom &g = pl->blk;
	memset(&g, 0, sizeof g);
	g.nx = nx; g.ny = ny; g.nz = nz; pl->blk_kind = pl->kinds[0] == DSPFFT_REDFT10 ? KIND_REDFT10 : KIND_REDFT01;
	g.sy_in = pl->axes[ay].is; g.sy_out = pl->axes[ay].os; g.sz_in = az >= 0 ? pl->axes[az].is : 0; g.sz_out = az >= 0 ? pl->axes[az].os : 0;
	g.sxb_in = grp.is; g.sxb_out = grp.os;
	g.rows_fast = !(grp.is == nx && grp.os == nx);
	// tiles of about 4096 samples = 16 KB (measured on 1920x1080x256, tools/bench_blocks.py: 8x8x8 blocks 2.44 ms per roundtrip with 64 KB
	// tiles, 1.77 ms with 16 KB ones -- eight and more workgroups per CU), rows of at most 1 KB
	int G = std::min(256 / nx, 4096 / (nx * ny * nz));
	if (env_int("DSPFFT_BLOCK_G")) G = env_int("DSPFFT_BLOCK_G");      // experiments (tools/bench_blocks.py)
	G = std::max(1, std::min(G, grp.n));
	if ((size_t)nz * ny * G * nx * sizeof(float) > 64 * 1024) return;
	g.G = G; g.nxb = grp.n; g.ngroups = (grp.n + G - 1) / G; g.pitch = G * nx; g.gdiv = make_div((uint32_t)g.ngroups);
	g.nd = (int)dims.size();
	for (int d = 0; d < g.nd; d++) { g.bn[d] = dims[d].n; g.bis[d] = dims[d].is; g.bos[d] = dims[d].os; g.bdiv[d] = make_div((uint32_t)dims[d].n); }
	const long long nwg = (long long)g.ngroups * nrest;
	if (nwg > 0x7fffffffLL) return;
	pl->blk_nwg = (int)nwg; pl->blk_lds = (size_t)nz * ny * g.pitch * sizeof(float);
	pl->blk_axis[0] = ax; pl->blk_axis[1] = ay; pl->blk_axis[2] = az;
	char buf[256];
	snprintf(buf, sizeof buf, "axes %d,%d%s: BLOCK %dx%dx%d (x,y,z) all axes in one pass, %d blocks per workgroup (%s), wgs=%d lds=%zu",
	         ax, ay, az >= 0 ? (std::string(",") + std::to_string(az)).c_str() : "", nx, ny, nz, G, g.rows_fast ? "block-major" : "side by side along x", pl->blk_nwg, pl->blk_lds);
	pl->blk_desc = buf;
	pl->has_block = true;
}

void block_scales(const dspfft_plan_s *pl, BlockScales &s)
{
	s.scale = (float)pl->scale;
	for (int i = 0; i < 3; i++) { const int ax = pl->blk_axis[i]; s.in0[i] = ax >= 0 ? (float)pl->in0[ax] : 1.f; s.out0[i] = ax >= 0 ? (float)pl->out0[ax] : 1.f; }
}

int run_block(const dspfft_plan_s *pl, const float *in, float *out, void *stream)
{
	BlockArgs a;
	static_cast<BlockGeom &>(a) = pl->blk;
	a.kind = pl->blk_kind; a.in = in; a.out = out;
	block_scales(pl, a.s);
	if (int rc = be_launch_block(a, pl->blk_nwg, pl->blk_lds, stream)) return fail(-4, "kernel launch failed (%s): backend code %d", pl->blk_desc.c_str(), rc);
	return 0;
}

// motion's per-block pipeline as one pass (block_core.h): both plans fuse their blocks the same way
bool block_roundtrip_ok(const dspfft_plan_s *f, const dspfft_plan_s *i)
{
	if (!f->has_block || !i->has_block || f->blk_kind != KIND_REDFT10 || i->blk_kind != KIND_REDFT01) return false;
	const BlockGeom &a = f->blk, &b = i->blk;
	bool ok = a.nx == b.nx && a.ny == b.ny && a.nz == b.nz && a.G == b.G && a.nxb == b.nxb && a.nd == b.nd && a.rows_fast == b.rows_fast &&
	          a.sy_out == b.sy_in && a.sz_out == b.sz_in && a.sxb_out == b.sxb_in && f->blk_nwg == i->blk_nwg;
	for (int k = 0; k < 3; k++) ok = ok && f->blk_axis[k] == i->blk_axis[k];
	for (int d = 0; ok && d < a.nd; d++) ok = a.bn[d] == b.bn[d] && a.bos[d] == b.bis[d];
	return ok;
}

const std::vector<Pass> &pick_passes(const dspfft_plan_s *pl, const void *in, const void *out)
{
	if (pl->split.empty()) return pl->passes;
	if (15u & ((uintptr_t)in | (uintptr_t)out)) return pl->passes;
	const int ca = pl->split_col_axis;
	// the row-pair butterfly mixes index 0 of the split axis with index N-1: scales on that index need the plain passes
	if (pl->kinds[ca] == DSPFFT_REDFT10 ? pl->in0[ca] != 1.0 : pl->out0[ca] != 1.0) return pl->passes;
	return pl->split;
}
}  // namespace

static int plan_finish(dspfft_plan_s *pl, dspfft_plan *plan, bool first_axis_first)
{
	EnvScope env;                 // the planner's switches, read once for this plan
	size_t samples = 1;
	pl->howmany = 1;
	for (const Dim &b : pl->batches) { samples *= (size_t)b.n; pl->howmany *= b.n; }
	for (int a = 0; a < pl->rank; a++) samples *= (size_t)pl->n[a];
	pl->alg_bytes = samples * (pl->f64 ? 16 : 8);
	pl->first_axis_first = first_axis_first;
	bool first = true;
	for (int i = 0; i < pl->rank; i++) {
		const int a = first_axis_first ? i : pl->rank - 1 - i;
		pl->passes.emplace_back();
		int rc = build_pass(pl, a, first, pl->passes.back());
		if (rc) { dspfft_destroy_plan(pl); return rc; }
		first = false;
	}
	build_split(pl);
	build_block(pl);
	*plan = pl;
	return 0;
}

static int plan_many(dspfft_plan *plan, int rank, const int *n, int howmany,
                     const int *inembed, int istride, int idist,
                     const int *onembed, int ostride, int odist, const int *kinds, bool f64, bool first_axis_first = false)
{
	if (!plan) return fail(-1, "null plan pointer");
	*plan = nullptr;
	if (rank < 1 || rank > 3) return fail(-1, "rank %d unsupported (1..3)", rank);
	if (howmany < 1 || istride < 1 || ostride < 1) return fail(-1, "howmany/stride must be >= 1");
	for (int a = 0; a < rank; a++) {
		if (n[a] < 1) return fail(-1, "n[%d] = %d", a, n[a]);
		if (kinds[a] != DSPFFT_REDFT10 && kinds[a] != DSPFFT_REDFT01) return fail(-1, "kind %d unsupported (REDFT10/REDFT01 only)", kinds[a]);
		if ((inembed && inembed[a] < n[a] && a > 0) || (onembed && onembed[a] < n[a] && a > 0)) return fail(-1, "embed smaller than n");
	}
	dspfft_plan_s *pl = new dspfft_plan_s();
	pl->rank = rank; pl->scale = 1.0; pl->f64 = f64;
	long long is = istride, os = ostride;
	for (int a = rank - 1; a >= 0; a--) {
		pl->n[a] = n[a]; pl->kinds[a] = kinds[a]; pl->in0[a] = pl->out0[a] = 1.0;
		pl->axes[a].n = n[a]; pl->axes[a].is = is; pl->axes[a].os = os;
		is *= inembed ? inembed[a] : n[a]; os *= onembed ? onembed[a] : n[a];
	}
	if (howmany > 1) pl->batches.push_back(Dim{howmany, idist, odist});
	return plan_finish(pl, plan, first_axis_first);
}

extern "C" int dspfft_plan_many_r2r(dspfft_plan *plan, int rank, const int *n, int howmany,
                                    const int *inembed, int istride, int idist,
                                    const int *onembed, int ostride, int odist, const int *kinds)
{
	return plan_many(plan, rank, n, howmany, inembed, istride, idist, onembed, ostride, odist, kinds, false);
}
extern "C" int dspfft_plan_many_r2r_f64(dspfft_plan *plan, int rank, const int *n, int howmany,
                                        const int *inembed, int istride, int idist,
                                        const int *onembed, int ostride, int odist, const int *kinds)
{
	return plan_many(plan, rank, n, howmany, inembed, istride, idist, onembed, ostride, odist, kinds, true);
}

// FFTW's guru interface shape (fftw_plan_guru_r2r): every transformed and every batch dimension has its own extent and
// input / output strides, so e.g. all 8x8x8 blocks of a volume (motion --blocksize 8x8x8) are ONE plan
extern "C" int dspfft_plan_guru_r2r(dspfft_plan *plan, int rank, const dspfft_iodim *dims, int howmany_rank, const dspfft_iodim *howmany_dims,
                                    const int *kinds, int f64)
{
	if (!plan) return fail(-1, "null plan pointer");
	*plan = nullptr;
	if (rank < 1 || rank > 3 || !dims || !kinds) return fail(-1, "rank %d unsupported (1..3)", rank);
	if (howmany_rank < 0 || howmany_rank > 6 || (howmany_rank && !howmany_dims)) return fail(-1, "howmany_rank %d unsupported (0..6)", howmany_rank);
	for (int a = 0; a < rank; a++) {
		if (dims[a].n < 1 || dims[a].is < 1 || dims[a].os < 1) return fail(-1, "dims[%d] must have n, is, os >= 1", a);
		if (kinds[a] != DSPFFT_REDFT10 && kinds[a] != DSPFFT_REDFT01) return fail(-1, "kind %d unsupported (REDFT10/REDFT01 only)", kinds[a]);
	}
	for (int b = 0; b < howmany_rank; b++)
		if (howmany_dims[b].n < 1 || howmany_dims[b].is < 0 || howmany_dims[b].os < 0) return fail(-1, "howmany_dims[%d] must have n >= 1 and non-negative strides", b);
	dspfft_plan_s *pl = new dspfft_plan_s();
	pl->rank = rank; pl->scale = 1.0; pl->f64 = f64 != 0;
	for (int a = 0; a < rank; a++) {
		pl->n[a] = dims[a].n; pl->kinds[a] = kinds[a]; pl->in0[a] = pl->out0[a] = 1.0;
		pl->axes[a].n = dims[a].n; pl->axes[a].is = dims[a].is; pl->axes[a].os = dims[a].os;
	}
	for (int b = 0; b < howmany_rank; b++) if (howmany_dims[b].n > 1) pl->batches.push_back(Dim{howmany_dims[b].n, howmany_dims[b].is, howmany_dims[b].os});
	return plan_finish(pl, plan, false);
}

extern "C" int dspfft_plan_many_r2r_ordered(dspfft_plan *plan, int rank, const int *n, int howmany,
                                            const int *inembed, int istride, int idist,
                                            const int *onembed, int ostride, int odist, const int *kinds, int first_axis_first)
{
	return plan_many(plan, rank, n, howmany, inembed, istride, idist, onembed, ostride, odist, kinds, false, first_axis_first != 0);
}

extern "C" int dspfft_plan_r2r_2d(dspfft_plan *plan, int n0, int n1, int kind0, int kind1)
{
	int n[2] = {n0, n1}, k[2] = {kind0, kind1};
	return dspfft_plan_many_r2r(plan, 2, n, 1, nullptr, 1, 0, nullptr, 1, 0, k);
}

extern "C" int dspfft_plan_set_scale(dspfft_plan pl, float scale)
{
	if (!pl) return fail(-1, "null plan");
	pl->scale = scale; return 0;
}
extern "C" int dspfft_plan_set_axis_scale0(dspfft_plan pl, int axis, float in_scale0, float out_scale0)
{
	if (!pl || axis < 0 || axis >= pl->rank) return fail(-1, "bad plan/axis");
	pl->in0[axis] = in_scale0; pl->out0[axis] = out_scale0; return 0;
}
// Input rows of `axis` outside [lo, hi) are zero by contract and need not be read.  Honoured (return 1) only where it is implemented:
// f32 plans whose FIRST pass is a listed specialised column REDFT01 pass along `axis` (zoom's y stage); returns 0 -- nothing changes, the
// caller must really store those zeros -- everywhere else.  lo = hi = 0 turns it off.
extern "C" int dspfft_plan_set_input_window(dspfft_plan pl, int axis, int lo, int hi)
{
	if (!pl || axis < 0 || axis >= pl->rank || lo < 0 || hi < lo || hi > pl->n[axis]) return fail(-1, "bad plan / axis / window");
	pl->win_axis = -1; pl->win_lo = pl->win_hi = 0;
	pl->mod_axis = -1; pl->mod_table = nullptr; pl->mod_rev = 0;      // a modulation lives on its window
	if (hi == 0) return 0;
	if (pl->f64 || pl->passes.empty() || !pl->split.empty() || pl->has_block) return 0;
	const Pass &P = pl->passes[0];
	if (P.axis != axis || (P.type != Pass::COL && P.type != Pass::ROW) || !P.has_spec || P.jit || pl->kinds[axis] != DSPFFT_REDFT01 || !P.hostloop.empty()) return 0;
	if (!pl->zpage) { pl->zpage = be_alloc(64); if (pl->zpage) { const char z[64] = {0}; if (be_upload(pl->zpage, z, 64)) { be_free(pl->zpage); pl->zpage = nullptr; } } }
	if (!pl->zpage) return 0;
	pl->win_axis = axis; pl->win_lo = lo; pl->win_hi = hi;
	return 1;
}
// Input sample x of `axis` is read from position p = reversed_from > 0 ? reversed_from - x : x of its line and multiplied by d_mul[p]
// (floats, device memory, kept by the caller) as it is loaded.  Only together with an input window (set it first): samples outside the
// window stay zero, and with reversed_from every sample inside it must map to a position >= 0.  Honoured (return 1) for f32 plans whose
// FIRST pass is a listed specialised ROW or COL REDFT01 pass along `axis`; d_mul = NULL turns it off.
extern "C" int dspfft_plan_set_input_modulation(dspfft_plan pl, int axis, const float *d_mul, int reversed_from)
{
	if (!pl || axis < 0 || axis >= pl->rank || reversed_from < 0) return fail(-1, "bad plan / axis / reversal");
	pl->mod_axis = -1; pl->mod_table = nullptr; pl->mod_rev = 0;
	if (!d_mul) return 0;
	if (pl->win_axis != axis || pl->win_hi <= 0) return 0;
	if (reversed_from > 0 && pl->win_hi - 1 > reversed_from) return fail(-1, "input window [%d, %d) reaches beyond the reversal point %d", pl->win_lo, pl->win_hi, reversed_from);
	if (pl->passes.empty() || (pl->passes[0].type != Pass::ROW && pl->passes[0].type != Pass::COL)) return 0;
	pl->mod_axis = axis; pl->mod_table = d_mul; pl->mod_rev = reversed_from;
	return 1;
}
// Output sample j of `axis` times (-1)^j, fused into the pass of that axis.  Honoured (return 1) only where it is implemented: f32 plans
// whose pass along `axis` is a listed specialised column or row REDFT01 pass and the plan's LAST one; 0 otherwise (nothing changes).
extern "C" int dspfft_plan_set_output_alternate(dspfft_plan pl, int axis, int on)
{
	if (!pl || axis < 0 || axis >= pl->rank) return fail(-1, "bad plan / axis");
	pl->alt_axis = -1;
	if (!on) return 0;
	if (pl->f64 || pl->passes.empty() || !pl->split.empty() || pl->has_block) return 0;
	const Pass &P = pl->passes.back();
	if (P.axis != axis || (P.type != Pass::COL && P.type != Pass::ROW) || !P.has_spec || P.jit || pl->kinds[axis] != DSPFFT_REDFT01 || !P.hostloop.empty()) return 0;
	pl->alt_axis = axis;
	return 1;
}
extern "C" int dspfft_plan_set_scale_f64(dspfft_plan pl, double scale)
{
	if (!pl) return fail(-1, "null plan");
	pl->scale = scale; return 0;
}
extern "C" int dspfft_plan_set_axis_scale0_f64(dspfft_plan pl, int axis, double in_scale0, double out_scale0)
{
	if (!pl || axis < 0 || axis >= pl->rank) return fail(-1, "bad plan/axis");
	pl->in0[axis] = in_scale0; pl->out0[axis] = out_scale0; return 0;
}

namespace {
template <class R>
int execute_t(dspfft_plan pl, const R *d_in, R *d_out, void *stream)
{
	if (!pl || !d_in || !d_out) return fail(-1, "null plan or buffer");
	if (pl->f64 != std::is_same<R, double>::value) return fail(-1, "plan and buffers differ in sample type (f32 plan <-> dspfft_execute, f64 plan <-> dspfft_execute_f64)");
	if constexpr (std::is_same<R, float>::value) {
		if (pl->has_block && !(15u & ((uintptr_t)d_in | (uintptr_t)d_out))) return run_block(pl, d_in, d_out, stream);
	}
	const std::vector<Pass> &passes = pick_passes(pl, d_in, d_out);
	for (size_t i = 0; i < passes.size(); i++) {
		const Pass &P = passes[i];
		int rc = run_pass<R>(pl, P, P.first ? d_in : d_out, d_out, i + 1 == passes.size(), stream);
		if (rc) return rc;
	}
	return 0;
}
template <class R>
int execute_masked_accumulate_t(dspfft_plan pl, const R *d_in, R *d_work, R *d_acc, const uint32_t *d_ids, uint32_t id, int elems_per_id, void *stream);
}  // namespace

extern "C" int dspfft_execute(dspfft_plan pl, const float *d_in, float *d_out, void *stream) { return execute_t<float>(pl, d_in, d_out, stream); }
extern "C" int dspfft_execute_f64(dspfft_plan pl, const double *d_in, double *d_out, void *stream) { return execute_t<double>(pl, d_in, d_out, stream); }
extern "C" int dspfft_execute_masked_accumulate(dspfft_plan pl, const float *d_in, float *d_work, float *d_acc,
                                                const uint32_t *d_ids, uint32_t id, int elems_per_id, void *stream)
{
	return execute_masked_accumulate_t<float>(pl, d_in, d_work, d_acc, d_ids, id, elems_per_id, stream);
}
extern "C" int dspfft_execute_masked_accumulate_f64(dspfft_plan pl, const double *d_in, double *d_work, double *d_acc,
                                                    const uint32_t *d_ids, uint32_t id, int elems_per_id, void *stream)
{
	return execute_masked_accumulate_t<double>(pl, d_in, d_work, d_acc, d_ids, id, elems_per_id, stream);
}

namespace {
// Sparse frames of the fused scan step: one image, a specialised column pass first and a specialised row pass over the same dense rows
// second.  The column pass flags the tiles none of whose coefficients belongs to the frame and leaves them alone; the row pass reads
// zeros there.  At BASELINE config 4 (zigzag, 32 frames) a frame touches about half of the columns.  The 1-D passes commute; a plain
// plan lists ROW before COL, and a frame of a wide image is sparser in columns than in rows, so the column pass goes first when input
// and output strides agree (a pass's geometry is built for its place in the list).  Returns false when the plan does not qualify.
bool sparse_order(const dspfft_plan_s *pl, const std::vector<Pass> &passes, bool split, size_t order[2])
{
	order[0] = 0; order[1] = 1;
	if (passes.size() != 2 || env_int("DSPFFT_NO_ZSKIP") == 1) return false;
	// measured (tools/bench_scan_step.py): 7680x4320x3 on the split passes 630 -> 513 us per frame; 3840x2160x3 on the plain passes
	// 119 -> 122 us (the frame lives in the Infinity Cache and the flag lookups cost what the skipped reads save), so plain plans
	// take part only on request (DSPFFT_ZSKIP=1: the tests of the mechanism on small frames)
	if (!split && env_int("DSPFFT_ZSKIP") != 1) return false;
	if (!split && passes[0].type == Pass::ROW && passes[1].type == Pass::COL) {
		bool same = true;
		for (int b = 0; b < pl->rank; b++) same = same && pl->axes[b].is == pl->axes[b].os;
		for (const Dim &b : pl->batches) same = same && b.is == b.os;
		if (same) { order[0] = 1; order[1] = 0; }
	}
	const Pass &PC = passes[order[0]], &PR = passes[order[1]];
	const bool col_ok = PC.type == Pass::COL && (split ? PC.half : PC.has_spec), row_ok = PR.type == Pass::ROW && (split ? PR.pair : PR.has_spec);
	const PassGeom &gc = split ? PC.hpa : PC.spa, &gr = PR.spa;
	const bool ok = col_ok && row_ok && PC.hostloop.empty() && PR.hostloop.empty() && !gr.sb2_in && !gr.sb2_out && gc.nb0 == 1 && gc.nb1 == 1 && gr.nb1 == 1 && gr.nb0 == gc.N &&
	                gr.N * gr.C == gc.ninner && gr.sb0_in == gc.ninner && gc.es_in == gc.ninner && gc.es_out == gc.ninner && gc.ninner % gc.K == 0 && (gc.K & (gc.K - 1)) == 0;
	if (!ok) { order[0] = 0; order[1] = 1; }
	return ok;
}
}  // namespace

extern "C" int dspfft_plan_scan_prepare(dspfft_plan pl, const uint32_t *d_ids, int elems_per_id, void *stream)
{
	if (!pl) return fail(-1, "null plan");
	pl->zr_ids = nullptr;                       // forget what was prepared before
	if (!d_ids) return 0;
	if (elems_per_id < 1) return fail(-1, "elems_per_id must be >= 1");
	const bool split = !pl->split.empty();
	const std::vector<Pass> &passes = split ? pl->split : pl->passes;
	size_t order[2];
	if (!sparse_order(pl, passes, split, order)) return 0;      // nothing to prepare for this plan: executes read every owner id
	const PassGeom &gc = split ? passes[order[0]].hpa : passes[order[0]].spa;
	const int halves = split ? 2 : 1;
	const size_t need = (size_t)2 * halves * gc.ntiles * sizeof(uint32_t);
	if (pl->zranges_bytes < need) { be_free(pl->zranges); pl->zranges = be_alloc(need); pl->zranges_bytes = pl->zranges ? need : 0; }
	if (!pl->zranges) return fail(-3, "allocation failed");
	TileRangeGeom g;
	g.K = gc.K; g.ntiles = gc.ntiles; g.nrows = gc.N / halves; g.row_start = 0; g.row_step = 1; g.es = gc.es_in; g.div = make_div((uint32_t)elems_per_id);
	if (be_scan_tile_ranges((uint32_t *)pl->zranges, d_ids, g, halves, stream)) return fail(-4, "launch failed");
	pl->zr_ids = d_ids; pl->zr_div = elems_per_id; pl->zr_split = split;
	// the element-order id table (dct_spec.h, masked_two_step): one byte per id while every frame id is below 255, else two.
	// DSPFFT_SCAN_EIDS=0: not built (A/B runs)
	pl->eid_bytes = 0;
	const char *env_eids = getenv("DSPFFT_SCAN_EIDS");
	const bool no_eids = env_eids && *env_eids == '0';
	if (!pl->f64 && !no_eids && gc.K % 4 == 0 && gc.es_in % elems_per_id == 0 && (long long)gc.ntiles * gc.K == gc.ninner) {
		std::vector<uint32_t> rg((size_t)2 * halves * gc.ntiles);
		if (be_download(rg.data(), pl->zranges, rg.size() * sizeof(uint32_t), stream)) return fail(-4, "download failed");
		uint32_t top = 0;
		for (size_t i = 0; i < rg.size(); i += 2) if (rg[i] <= rg[i + 1] && rg[i + 1] > top) top = rg[i + 1];      // (an empty tile holds lo > hi)
		const int eb = top < 0xffu ? 1 : top < 0xffffu ? 2 : 0;
		const size_t bytes = (size_t)gc.ntiles * gc.N * gc.K * eb;
		if (eb) {
			if (pl->eids_bytes < bytes) { be_free(pl->eids); pl->eids = be_alloc(bytes); pl->eids_bytes = pl->eids ? bytes : 0; }
			if (pl->eids) {
				TileEidGeom e;
				e.K = gc.K; e.ntiles = gc.ntiles; e.N = gc.N; e.halves = halves; e.bytes = eb; e.es = gc.es_in; e.div = make_div((uint32_t)elems_per_id);
				if (be_scan_tile_eids(pl->eids, d_ids, e, stream)) return fail(-4, "launch failed");
				pl->eid_bytes = eb;
			}
		}
	}
	return 0;
}

namespace {
template <class R>
int execute_masked_accumulate_t(dspfft_plan pl, const R *d_in, R *d_work, R *d_acc, const uint32_t *d_ids, uint32_t id, int elems_per_id, void *stream)
{
	if (!pl || !d_in || !d_work || !d_acc) return fail(-1, "null plan or buffer");
	if (pl->f64 != std::is_same<R, double>::value) return fail(-1, "plan and buffers differ in sample type");
	if (d_ids && elems_per_id < 1) return fail(-1, "elems_per_id must be >= 1");
	if (d_ids && (unsigned long long)pl->alg_bytes / (2 * sizeof(R)) * (unsigned)elems_per_id >= (1ull << 32))
		return fail(-2, "masked execution addresses elements with 32-bit offsets: plan too large");
	// the split passes (row pairs + half tiles) carry the mask and the accumulation like the plain ones: the row-pair butterfly
	// runs on the masked inputs, which is what masking the coefficients first means
	const bool split_ok = !pl->split.empty() && !(15u & ((uintptr_t)d_in | (uintptr_t)d_work | (uintptr_t)d_acc)) && &pick_passes(pl, d_in, d_acc) == &pl->split;
	const std::vector<Pass> &passes = split_ok ? pl->split : pl->passes;
	const size_t np = passes.size();
	for (const Pass &P : passes)
		if (!P.hostloop.empty() || P.pa.sb2_in || P.pa.sb2_out) return fail(-2, "masked/accumulating execution is not available for plans with more than two batch levels");
	// Sparse frames: one image, a specialised column pass first and a specialised row pass over the same dense rows second.  The
	// column pass flags the tiles none of whose coefficients belongs to this frame and leaves them alone; the row pass reads zeros
	// there.  At BASELINE config 4 (zigzag, 32 frames) a frame touches about half of the columns.
	uint8_t *zflags = nullptr;
	int zshift = 0, zhalf = 0;
	const uint32_t *zranges = nullptr;
	const void *eids = nullptr;
	int eid_bytes = 0;
	size_t order[2] = {0, 1};
	if (d_ids && np == 2 && sparse_order(pl, passes, split_ok, order)) {
		const PassGeom &gc = split_ok ? passes[order[0]].hpa : passes[order[0]].spa;
		const size_t al = sizeof(R) * 4;
		if (!((al - 1) & ((uintptr_t)d_in | (uintptr_t)d_work | (uintptr_t)d_acc))) {
			const size_t need = (size_t)2 * gc.ntiles;
			if (pl->zflags_bytes < need) { be_free(pl->zflags); pl->zflags = be_alloc(need); pl->zflags_bytes = pl->zflags ? need : 0; }
			if (!pl->zpage) { pl->zpage = be_alloc(64); if (pl->zpage) { const char z[64] = {0}; if (be_upload(pl->zpage, z, 64)) { be_free(pl->zpage); pl->zpage = nullptr; } } }
			if (pl->zflags && pl->zpage) {
				zflags = (uint8_t *)pl->zflags; zhalf = split_ok ? gc.ntiles : 0;
				while ((1 << zshift) < gc.K) zshift++;
				if (pl->zranges && pl->zr_ids == (const void *)d_ids && pl->zr_div == elems_per_id && pl->zr_split == split_ok) {
					zranges = (const uint32_t *)pl->zranges;
					if (pl->eid_bytes) { eids = pl->eids; eid_bytes = pl->eid_bytes; }
				}
			}
		}
	}
	if (!zflags) { order[0] = 0; order[1] = 1; }
	for (size_t i = 0; i < np; i++) {
		const Pass &P = passes[np == 2 ? order[i] : i];
		const bool firstp = i == 0, lastp = i + 1 == np;
		Fuse fz;
		if (firstp && d_ids) { fz.mask = d_ids; fz.id = id; fz.div = elems_per_id; }
		fz.accumulate = lastp;
		fz.zflags = zflags; fz.zshift = zshift; fz.zhalf = zhalf; fz.zpage = pl->zpage; fz.zranges = zranges;
		if (firstp && std::is_same<R, float>::value) { fz.eids = eids; fz.eid_bytes = eid_bytes; }
		const R *src = firstp ? d_in : d_work;
		R *dst = lastp ? d_acc : d_work;
		int rc = run_pass<R>(pl, P, src, dst, lastp, stream, fz);
		if (rc) return rc;
	}
	return 0;
}
}  // namespace

// out = a(in_a) + b(in_b).  One launch when both are f32 single-pass plans on the same listed row REDFT01 kernel writing the same lines
// (zoom's x stage: the cosine and the sine part of a shifted cosine series); otherwise a's execution followed by b's accumulating one.
extern "C" int dspfft_execute_sum2(dspfft_plan pa, dspfft_plan pb, const float *d_in_a, const float *d_in_b, float *d_out, void *stream)
{
	if (!pa || !pb || !d_in_a || !d_in_b || !d_out) return fail(-1, "null plan or buffer");
	if (pa->f64 || pb->f64) return fail(-1, "dspfft_execute_sum2 takes f32 plans");
	auto one_row = [](const dspfft_plan_s *pl) {
		if (pl->passes.size() != 1 || !pl->split.empty() || pl->has_block) return false;
		const Pass &P = pl->passes[0];
		return P.type == Pass::ROW && P.has_spec && !P.jit && P.hostloop.empty() && !P.pa.sb2_in && !P.pa.sb2_out && pl->kinds[P.axis] == DSPFFT_REDFT01;
	};
	if (one_row(pa) && one_row(pb)) {
		const Pass &A = pa->passes[0], &B = pb->passes[0];
		const PassGeom &ga = A.spa, &gb = B.spa;
		if (A.spec.id == B.spec.id && A.spec_nwg == B.spec_nwg && ga.nb0 == gb.nb0 && ga.nb1 == gb.nb1 && ga.sb0_out == gb.sb0_out && ga.sb1_out == gb.sb1_out) {
			// same pointer rule as run_pass (whole pixels per access); the two-launch path below applies it by itself
			const uintptr_t al = sizeof(float) * (ga.C == 2 || ga.C == 4 ? ga.C : 1);
			if ((al - 1) & ((uintptr_t)d_in_a | (uintptr_t)d_in_b | (uintptr_t)d_out)) return fail(-1, "dspfft_execute_sum2: buffers must be aligned to a pixel of %d floats", ga.C);
			PassArgs a, b;
			fill_args(a, ga, pa, A, d_in_a, d_out, pa->scale, Fuse());
			fill_args(b, gb, pb, B, d_in_b, d_out, pb->scale, Fuse());
			if (int rc = be_launch_row_sum2(A.spec.id, a, b, A.spec_nwg, stream)) return fail(-4, "kernel launch failed (%s, sum of two): backend code %d", A.desc.c_str(), rc);
			return 0;
		}
	}
	// one after the other: b's accumulating execution has no work buffer of its own here, so b must be a one-pass plan
	if (pb->passes.size() != 1 || pb->has_block) return fail(-2, "dspfft_execute_sum2: the second plan must be a one-pass plan (its result is added in that pass's store)");
	if (int rc = dspfft_execute(pa, d_in_a, d_out, stream)) return rc;
	return dspfft_execute_masked_accumulate(pb, d_in_b, d_out, d_out, nullptr, 0, 1, stream);
}

// ---- rows of a phase-shifted cosine series (include/dspfft.h dspfft_cosrows_*; dct_duo.h) ----
struct dspfft_cosrows_s {
	int M, cw, vw, lines, id, nsrc;
	void *W;        // exp(-2 pi i t / (M/2)), t < M/2
	float *tab;     // [nsrc][M/2] slots of four floats, rebuilt per execution
};
extern "C" int dspfft_cosrows_create(dspfft_cosrows *out, int M, int cw, int vw, int lines)
{
	if (!out) return fail(-1, "null plan pointer");
	*out = nullptr;
	if (M < 4 || cw < 1 || cw > M || vw < 1 || vw > M || lines < 1) return fail(-1, "bad arguments (M %d, cw %d, vw %d, lines %d)", M, cw, vw, lines);
	const int id = be_find_zoomx(M);
	if (id < 0) return fail(-2, "no cosine-series row kernel for lines of %d samples", M);
	dspfft_cosrows p = new dspfft_cosrows_s();
	p->M = M; p->cw = cw; p->vw = vw; p->lines = lines; p->id = id;
	p->nsrc = 4 * cw <= M ? 1 : 2 * cw <= M ? 2 : 4;
	const int L = M / 2;
	std::vector<cf> w((size_t)L);
	const long double pi = 3.14159265358979323846264338327950288L;
	for (int t = 0; t < L; t++) w[t] = cmk<float>((float)cosl(2 * pi * t / L), (float)-sinl(2 * pi * t / L));
	p->W = be_alloc((size_t)L * sizeof(cf));
	p->tab = (float *)be_alloc((size_t)p->nsrc * L * 4 * sizeof(float));
	if (!p->W || !p->tab || be_upload(p->W, w.data(), w.size() * sizeof(cf))) {
		if (p->W) be_free(p->W);
		if (p->tab) be_free(p->tab);
		delete p;
		return fail(-3, "no device memory for the tables");
	}
	*out = p;
	return 0;
}
extern "C" int dspfft_cosrows_execute(dspfft_cosrows p, const float *d_in, long long in_pitch, float *d_out, long long out_pitch, double theta, double scale, void *stream)
{
	if (!p || !d_in || !d_out) return fail(-1, "null plan or buffer");
	if (in_pitch < 0 || out_pitch < 0 || (p->lines > 1 && (in_pitch < (long long)p->cw * 3 || out_pitch < (long long)p->vw * 3))) return fail(-1, "line pitch shorter than a line");
	if (3u & ((uintptr_t)d_in | (uintptr_t)d_out)) return fail(-1, "buffers must be 4-byte aligned");
	// 32-bit offsets inside a line
	if ((long long)p->M * 3 >= (1ll << 30)) return fail(-2, "line too long");
	if (int rc = be_zoomx_tables(p->tab, p->M, p->cw, p->nsrc, theta, scale, stream)) return fail(-4, "table kernel launch failed: backend code %d", rc);
	ZoomXArgs a;
	a.in = d_in; a.out = d_out; a.tab = p->tab; a.W = (const cf *)p->W; a.in_pitch = in_pitch; a.out_pitch = out_pitch; a.cw = p->cw; a.vw = p->vw; a.lines = p->lines;
	if (int rc = be_launch_zoomx(p->id, a, p->nsrc, p->vw < p->M, stream)) return fail(-4, "kernel launch failed (cosine-series rows, M = %d): backend code %d", p->M, rc);
	return 0;
}
extern "C" void dspfft_cosrows_destroy(dspfft_cosrows p)
{
	if (!p) return;
	be_free(p->W); be_free(p->tab);
	delete p;
}

// ---- chirp-z rows (include/dspfft.h dspfft_cztrows_*; dct_czt.h) ----
struct dspfft_cztrows_s {
	int nc, nout, lines, group, id, P;
	void *W;                    // exp(-2 pi i t / P)
	cf *atab, *etab, *htab, *hspec;
	double omega_of_spectrum;   // the omega hspec was made for (NaN: none yet)
};
extern "C" int dspfft_cztrows_create(dspfft_cztrows *out, int nc, int nout, int lines, int group)
{
	if (!out) return fail(-1, "null plan pointer");
	*out = nullptr;
	if (nc < 1 || nout < 1 || lines < 1 || group < 1 || lines % group) return fail(-1, "bad arguments (nc %d, nout %d, lines %d, group %d)", nc, nout, lines, group);
	int P = 0;
	const int id = be_find_czt(nc + nout - 1, &P);
	if (id < 0) return fail(-2, "no chirp-z row kernel for a convolution of %d points", nc + nout - 1);
	dspfft_cztrows p = new dspfft_cztrows_s();
	p->nc = nc; p->nout = nout; p->lines = lines; p->group = group; p->id = id; p->P = P;
	p->omega_of_spectrum = std::nan("");
	std::vector<cf> w((size_t)P);
	const long double pi = 3.14159265358979323846264338327950288L;
	for (int t = 0; t < P; t++) w[t] = cmk<float>((float)cosl(2 * pi * t / P), (float)-sinl(2 * pi * t / P));
	p->W = be_alloc((size_t)P * sizeof(cf));
	p->atab = (cf *)be_alloc(((size_t)nc + nout + 2 * (size_t)P) * sizeof(cf));
	if (!p->W || !p->atab || be_upload(p->W, w.data(), w.size() * sizeof(cf))) {
		if (p->W) be_free(p->W);
		if (p->atab) be_free(p->atab);
		delete p;
		return fail(-3, "no device memory for the tables");
	}
	p->etab = p->atab + nc; p->htab = p->etab + nout; p->hspec = p->htab + P;
	*out = p;
	return 0;
}
extern "C" int dspfft_cztrows_length(dspfft_cztrows p) { return p ? p->P : 0; }
extern "C" int dspfft_cztrows_execute(dspfft_cztrows p, const float *d_in, long long in_group, long long in_pitch, int es_in,
                                      float *d_out, long long out_group, long long out_pitch, int es_out, double omega, double phi, double scale, void *stream)
{
	if (!p || !d_in || !d_out || es_in < 1 || es_out < 1) return fail(-1, "null plan or buffer, or a stride below 1");
	if (3u & ((uintptr_t)d_in | (uintptr_t)d_out)) return fail(-1, "buffers must be 4-byte aligned");
	const bool fresh = !(omega == p->omega_of_spectrum);
	if (int rc = be_czt_tables(p->atab, p->etab, fresh ? p->htab : nullptr, p->nc, p->nout, p->P, omega, phi, scale, stream)) return fail(-4, "table kernel launch failed: backend code %d", rc);
	CztArgs a;
	a.in = nullptr; a.out = nullptr; a.atab = p->htab; a.hspec = nullptr; a.etab = nullptr; a.W = (const cf *)p->W;
	a.in_pitch = a.out_pitch = a.in_group = a.out_group = 0; a.es_in = a.es_out = 1; a.nc = p->P; a.nout = 0; a.lines = 1; a.group = 1;
	if (fresh) {
		if (int rc = be_launch_czt_spectrum(p->id, a, p->hspec, stream)) return fail(-4, "kernel launch failed (chirp spectrum, P = %d): backend code %d", p->P, rc);
		p->omega_of_spectrum = omega;
	}
	a.in = d_in; a.out = d_out; a.atab = p->atab; a.hspec = p->hspec; a.etab = p->etab;
	a.in_pitch = in_pitch; a.out_pitch = out_pitch; a.in_group = in_group; a.out_group = out_group; a.es_in = es_in; a.es_out = es_out;
	a.nc = p->nc; a.nout = p->nout; a.lines = p->lines; a.group = p->group;
	if (int rc = be_launch_czt_rows(p->id, a, stream)) return fail(-4, "kernel launch failed (chirp-z rows, P = %d): backend code %d", p->P, rc);
	return 0;
}
extern "C" void dspfft_cztrows_destroy(dspfft_cztrows p)
{
	if (!p) return;
	be_free(p->W); be_free(p->atab);
	delete p;
}
extern "C" int dspfft_transpose_f32(float *d_out, long long out_pitch, const float *d_in, long long in_pitch, int rows, int cols, void *stream)
{
	if (!d_out || !d_in || rows < 1 || cols < 1 || out_pitch < rows || in_pitch < cols) return fail(-1, "bad arguments");
	if (int rc = be_transpose(d_out, out_pitch, d_in, in_pitch, rows, cols, stream)) return fail(-4, "kernel launch failed (transpose): backend code %d", rc);
	return 0;
}

namespace {
// spec_fused.h is_plain, for the host side: none of the fused scan step's or zoom's extras asked for
bool is_plain_args(const PassArgs &a) { return !a.mask && !a.zflags && !a.accumulate && a.win_hi <= 0 && !a.alt_out && !a.in_mul && !a.in_rev; }
// a planar row pass that can take / produce 8-bit samples itself
bool pass_has_u8(const dspfft_plan_s *pl, const Pass &P)
{
	if (P.type != Pass::ROW || P.pa.C != 1 || !P.hostloop.empty()) return false;
	// the 8-bit kernels are plain instantiations (spec_kernels.h): a plan with zoom's window / modulation / alternating sign on this axis converts separately
	if (pl->alt_axis == P.axis || (pl->win_axis == P.axis && P.first && pl->zpage)) return false;
	return (P.has_spec && be_spec_has_u8(P.spec.id)) || (P.jit && P.jit_fn_u8);
}

int run_pass_u8(const dspfft_plan_s *pl, const Pass &P, const float *in, float *out, bool last, const U8IO &io, void *stream)
{
	PassArgs a;
	fill_args(a, P.spa, pl, P, in, out, last ? pl->scale : 1.0, Fuse());
	if (P.jit && !P.has_spec) {
		U8IO io2 = io;
		void *args[2] = {&a, &io2};
		if (int rc = be_jit_launch_n(P.jit_fn_u8, args, P.spec_nwg, P.jit_nthr, stream)) return fail(-4, "kernel launch failed (%s, u8): backend code %d", P.desc.c_str(), rc);
		return 0;
	}
	if (int rc = be_launch_spec_u8(P.spec.id, a, io, P.spec_nwg, stream)) return fail(-4, "kernel launch failed (%s, u8): backend code %d", P.desc.c_str(), rc);
	return 0;
}

int roundtrip_core(dspfft_plan fwd, dspfft_plan inv, const float *d_in, float *d_out, const uint8_t *d_in8, uint8_t *d_out8, double mul8,
                   const dspfft_motion_filter_params *fp, unsigned long long *d_coeffs_coded, void *stream, bool may_slice = true);

// ---- a clip of frames in slices (motion's per-frame blocks, motion/motion.c:591,613-615: the frames are independent) ----
// The three launches of the 8-bit roundtrip move the clip's float intermediate through HBM four times (2.1 GB for config 5's luma plane).  Walked
// in slices of S frames -- 8-bit rows -> fused column roundtrip -> 8-bit rows per slice, every slice through the SAME S frames of the work
// buffer -- the intermediate stays in the 256 MB Infinity Cache and only the 8-bit ends touch HBM.  What that buys is bounded by the kernels,
// which are issue- and latency-bound rather than HBM-bound (profiles/r06_motion_slices.txt): the column kernel gains 5-10 % on a resident slice
// with the narrow K = 8 tile (four workgroups per CU; on the HBM-resident clip K = 16's 64-byte segments win), the row kernels lose a few per
// cent to the launch tails of eleven launches instead of one.  Slices alternate between the caller's stream and one of the library's own with a
// work area each, so that one slice's launch tails fill with the other's kernels.
// Which plans: 2-D f32 per-frame plans with ONE batch level whose column pass has a narrow-tile entry (spec_list.h) and whose clip is larger
// than two slices.  DSPFFT_RT_SLICE=frames forces a slice size (0: never), DSPFFT_RT_STREAMS=1 keeps every slice on the caller's stream.
constexpr size_t kSliceBytes = 100u << 20;        // per work area (two areas in flight: 200 MB of the cache's 256)
constexpr int kSliceTileK = 8;

int slice_frames(const dspfft_plan_s *fwd, const dspfft_plan_s *inv, const dspfft_motion_filter_params *fp, int nstreams)
{
	static const int forced = []() { const char *e = getenv("DSPFFT_RT_SLICE"); return e ? atoi(e) : -1; }();
	if (forced == 0 || fwd->rank != 2 || fwd->batches.size() != 1 || inv->batches.size() != 1 || fwd->passes.size() != 2 || inv->passes.size() != 2) return 0;
	if (fwd->col_kpref || inv->col_kpref) return 0;                     // (a slice plan itself)
	const Pass &F = fwd->passes[1], &I = inv->passes[0];
	if (F.type != Pass::COL || I.type != Pass::COL || !F.has_spec || !I.has_spec || F.jit || I.jit) return 0;
	SpecInfo narrow;
	if (!be_find_spec(1, F.pa.N, F.pa.ninner, &narrow, kSliceTileK) || narrow.P != kSliceTileK || narrow.P == F.spec.P) return 0;
	const Dim &b = fwd->batches[0];
	const long long frame = b.os;                                       // floats between frames of the work layout
	if (frame <= 0 || inv->batches[0].is != b.os || inv->batches[0].os != b.os) return 0;
	long long S = forced > 0 ? forced : (long long)(kSliceBytes * (nstreams > 1 ? 1 : 2) / ((size_t)frame * sizeof(float)));
	if (forced <= 0 && b.n < 4 * S) S = (b.n + 3) / 4;                  // a short clip (a rank's share of a clip split over eight GPUs): four slices,
	const int bd = fp ? fp->block_depth : 1;                            // the filter finds a frame's place in its block from its offset in the work area
	if (bd > 1) S -= S % bd;
	if (S < 1 || (forced <= 0 && S < 8)) return 0;                      // of at least eight frames each (32 frames: 0.369 -> 0.356 ms; fewer: launch tails take over)
	return (int)std::min<long long>(S, b.n);
}

dspfft_plan_s *slice_plan(const dspfft_plan_s *of, int frames)
{
	dspfft_plan_s *pl = new dspfft_plan_s();
	pl->rank = of->rank; pl->scale = of->scale; pl->f64 = of->f64;
	for (int a = 0; a < of->rank; a++) { pl->n[a] = of->n[a]; pl->kinds[a] = of->kinds[a]; pl->in0[a] = of->in0[a]; pl->out0[a] = of->out0[a]; pl->axes[a] = of->axes[a]; }
	if (frames > 1) pl->batches.push_back(Dim{frames, of->batches[0].is, of->batches[0].os});
	pl->col_kpref = kSliceTileK;
	dspfft_plan out = nullptr;
	return plan_finish(pl, &out, of->first_axis_first) ? nullptr : out;
}

// returns 1 when the clip was run in slices, 0 when this call does not qualify (the caller runs it whole), < 0 on error
int roundtrip_sliced(dspfft_plan fwd, dspfft_plan inv, float *d_work, const uint8_t *d_in8, uint8_t *d_out8, double mul8,
                     const dspfft_motion_filter_params *fp, unsigned long long *d_coeffs_coded, void *stream)
{
	static const int nstreams = []() { const char *e = getenv("DSPFFT_RT_STREAMS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : v > 2 ? 2 : v; }();
	const int S = slice_frames(fwd, inv, fp, nstreams);
	if (!S) return 0;
	std::lock_guard<std::mutex> lock(fwd->rt_mutex);
	dspfft_plan_s::RtSlices *r = nullptr;
	for (dspfft_plan_s::RtSlices &c : fwd->rt_slices) if (c.inv_of == inv && c.frames == S) r = &c;
	if (!r) {
		dspfft_plan_s::RtSlices c;
		const int total = fwd->batches[0].n, rem = total % S;
		c.inv_of = inv; c.frames = S;
		c.fwd = slice_plan(fwd, S); c.inv = slice_plan(inv, S);
		if (rem) { c.fwd_rem = slice_plan(fwd, rem); c.inv_rem = slice_plan(inv, rem); }
		if (nstreams > 1) { c.side = be_stream_create(); c.ev_fork = be_order_event_create(); c.ev_join = be_order_event_create(); }
		const bool ok = c.fwd && c.inv && (!rem || (c.fwd_rem && c.inv_rem)) && (nstreams == 1 || (c.side && c.ev_fork && c.ev_join));
		if (!ok) {
			dspfft_destroy_plan(c.fwd); dspfft_destroy_plan(c.inv); dspfft_destroy_plan(c.fwd_rem); dspfft_destroy_plan(c.inv_rem);
			if (c.side) be_stream_destroy(c.side);
			if (c.ev_fork) be_event_destroy(c.ev_fork);
			if (c.ev_join) be_event_destroy(c.ev_join);
			return 0;
		}
		fwd->rt_slices.push_back(c);
		r = &fwd->rt_slices.back();
	}
	// the parents' scales may have been set since (dspfft_plan_set_scale / _set_axis_scale0)
	for (dspfft_plan_s *q : {r->fwd, r->fwd_rem}) if (q) { q->scale = fwd->scale; for (int a = 0; a < 2; a++) { q->in0[a] = fwd->in0[a]; q->out0[a] = fwd->out0[a]; } }
	for (dspfft_plan_s *q : {r->inv, r->inv_rem}) if (q) { q->scale = inv->scale; for (int a = 0; a < 2; a++) { q->in0[a] = inv->in0[a]; q->out0[a] = inv->out0[a]; } }
	const int total = fwd->batches[0].n;
	const long long fin = fwd->batches[0].is, fwk = fwd->batches[0].os, fout = inv->batches[0].os;
	const bool two = nstreams > 1 && r->side;
	if (two && (be_event_record(r->ev_fork, stream) || be_stream_wait_event(r->side, r->ev_fork))) return fail(-4, "stream fork failed");
	int k = 0;
	for (int f0 = 0; f0 < total; f0 += S, k++) {
		const bool last = total - f0 < S;
		void *st = (two && (k & 1)) ? r->side : stream;
		float *work = d_work + ((two && (k & 1)) ? (long long)S * fwk : 0);
		if (int rc = roundtrip_core(last ? r->fwd_rem : r->fwd, last ? r->inv_rem : r->inv, nullptr, work, d_in8 + (long long)f0 * fin, d_out8 + (long long)f0 * fout, mul8,
		                            fp, d_coeffs_coded, st, false)) return rc;
	}
	if (two && (be_event_record(r->ev_join, r->side) || be_stream_wait_event(stream, r->ev_join))) return fail(-4, "stream join failed");
	return 1;
}

// d_in8 / d_out8 non-NULL: 8-bit samples at the two ends (dspfft_execute_roundtrip_u8), d_out is then the float work buffer
int roundtrip_core(dspfft_plan fwd, dspfft_plan inv, const float *d_in, float *d_out, const uint8_t *d_in8, uint8_t *d_out8, double mul8,
                   const dspfft_motion_filter_params *fp, unsigned long long *d_coeffs_coded, void *stream, bool may_slice)
{
	if (!fwd || !inv || !(d_in || d_in8) || !d_out) return fail(-1, "null plan or buffer");
	if (fwd->f64 || inv->f64) return fail(-1, "the fused roundtrip takes f32 plans");
	const size_t nf = fwd->passes.size(), ni = inv->passes.size();
	const Pass &F = fwd->passes[nf - 1], &I = inv->passes[0];
	bool differs = fwd->rank != inv->rank;
	for (int a = 0; !differs && a < fwd->rank; a++) differs = fwd->n[a] != inv->n[a];
	// (plans whose blocks go through the fused block pass have no pass order to agree on)
	if (fwd->rank != inv->rank || fwd->howmany != inv->howmany || (!differs && F.axis != I.axis && !block_roundtrip_ok(fwd, inv)))
		return fail(-1, "the forward plan's last pass and the inverse plan's first pass must run along the same axis (create the inverse with dspfft_plan_many_r2r_ordered(..., 1))");
	// motion's `scaled != block` (motion.c:535-552): the inverse runs over DIFFERENT extents inside the same embedding -- larger:
	// the spectrum is zero-padded (band-limited upscale), smaller: truncated (downscale)
	bool rescale = false;
	for (int a = 0; a < fwd->rank; a++) {
		if (fwd->axes[a].os != inv->axes[a].is || inv->axes[a].is != inv->axes[a].os || fwd->kinds[a] != DSPFFT_REDFT10 || inv->kinds[a] != DSPFFT_REDFT01)
			return fail(-1, "roundtrip: the inverse must be REDFT01, in place, on the forward (REDFT10) plan's output layout");
		rescale = rescale || fwd->n[a] != inv->n[a];
	}
	if (rescale && (!fwd->batches.empty() || !inv->batches.empty())) return fail(-2, "roundtrip with different forward / inverse extents takes one block per call (howmany = 1)");
	if (rescale)
		for (int a = 0; a < fwd->rank; a++)
			if (fwd->axes[a].is != fwd->axes[a].os) return fail(-2, "roundtrip with different forward / inverse extents needs ONE embedding for input, work and output (motion.c:535-552)");
	if (fwd->batches.size() != inv->batches.size()) return fail(-1, "roundtrip: batch layouts differ");
	for (size_t b = 0; b < fwd->batches.size(); b++)
		if (fwd->batches[b].n != inv->batches[b].n || fwd->batches[b].os != inv->batches[b].is || inv->batches[b].is != inv->batches[b].os) return fail(-1, "roundtrip: batch layouts differ");
	MotionFilter mf;
	memset(&mf, 0, sizeof mf);
	if (fp) {
		if (fp->preserve_dc < 0 || fp->preserve_dc > 2 || fp->minbuf_hw[0] < 1 || fp->minbuf_hw[1] < 1 || fp->block_depth < 1) return fail(-1, "bad filter parameters");
		mf.ad = fp->active[0]; mf.ah = fp->active[1]; mf.aw = fp->active[2]; mf.mh = fp->minbuf_hw[0]; mf.mw = fp->minbuf_hw[1];
		mf.b0d = fp->band_begin[0]; mf.b0h = fp->band_begin[1]; mf.b0w = fp->band_begin[2];
		mf.b1d = fp->band_end[0]; mf.b1h = fp->band_end[1]; mf.b1w = fp->band_end[2];
		mf.damp = fp->damp; mf.boost = fp->boost; mf.thr_lo = fp->threshold_lo; mf.thr_hi = fp->threshold_hi;
		mf.preserve_dc = fp->preserve_dc; mf.grey_add = fp->grey_add; mf.quantizer = fp->quantizer; mf.enabled = 1;
		motion_filter_set_divs(mf, fp->block_depth);
	}
	// extent of the working buffer in elements (the filter addresses it with 32-bit offsets)
	long long span = 1;
	for (int a = 0; a < fwd->rank; a++) span += (long long)(std::max(fwd->n[a], inv->n[a]) - 1) * fwd->axes[a].os;
	for (const Dim &b : fwd->batches) span += (long long)(b.n - 1) * b.os;
	if (fp && span >= (1ll << 31)) return fail(-2, "filtered roundtrip addresses the buffer with 31-bit offsets: buffer too large");
	// region geometry of the two ends (rank padded to 3 with unit extents)
	int nf3[3] = {1, 1, 1}, ni3[3] = {1, 1, 1};
	long long sw3[3] = {0, 0, 0}, si3[3] = {0, 0, 0};
	for (int a = 0; a < fwd->rank; a++) {
		const int k = 3 - fwd->rank + a;
		nf3[k] = fwd->n[a]; ni3[k] = inv->n[a]; sw3[k] = fwd->axes[a].os; si3[k] = fwd->axes[a].is;
	}
	// small blocks: everything in one pass over the data
	if (!rescale && block_roundtrip_ok(fwd, inv)) {
		const uintptr_t pin = d_in8 ? (uintptr_t)d_in8 : (uintptr_t)d_in, pout = d_out8 ? (uintptr_t)d_out8 : (uintptr_t)d_out;
		if (!((d_in8 ? 3u : 15u) & pin) && !((d_out8 ? 3u : 15u) & pout)) {
			BlockRtArgs a;
			static_cast<BlockGeom &>(a) = fwd->blk;
			const BlockGeom &o = inv->blk;
			a.sy_out = o.sy_out; a.sz_out = o.sz_out; a.sxb_out = o.sxb_out;
			for (int d = 0; d < o.nd; d++) a.bos[d] = o.bos[d];
			a.in = d_in8 ? nullptr : d_in; a.out = d_out8 ? nullptr : d_out; a.in8 = d_in8; a.out8 = d_out8; a.mul8 = mul8;
			block_scales(fwd, a.f); block_scales(inv, a.i);
			a.filt = mf; a.coded = d_coeffs_coded;
			if (int rc = be_launch_block_roundtrip(a, fwd->blk_nwg, fwd->blk_lds, stream)) return fail(-4, "kernel launch failed (fused block roundtrip): backend code %d", rc);
			return 0;
		}
	}
	// the standalone filter finds a coefficient's position from its offset in a block-major embedding (minbuf_hw, block_depth); the
	// blocks of a volume lying side by side are filtered by the fused block pass only, which knows each block's own coordinates
	if (fp && fwd->has_block && !fwd->blk.rows_fast)
		return fail(-2, "filtered roundtrip over the blocks of a volume needs the fused block pass: matching forward / inverse plans and 16-byte (8-bit: 4-byte) aligned buffers");
	if (rescale) {
		// one block, unfused: zero the working buffer (motion.c:619), load the block region, forward, filter, inverse over the
		// scaled region, store it.  In place on a float buffer the caller has zeroed everything outside the block itself.
		if (d_in8) {
			if (be_zero(d_out, (size_t)span * sizeof(float), stream) || be_region_u8_to_f32(d_out, d_in8, nf3, sw3, si3, stream)) return fail(-4, "launch failed");
			d_in = d_out;
		} else if (d_in != d_out) {
			if (be_zero(d_out, (size_t)span * sizeof(float), stream)) return fail(-4, "launch failed");
		}
		for (size_t i = 0; i < nf; i++) {
			const Pass &P = fwd->passes[i];
			if (int rc = run_pass<float>(fwd, P, (P.first && !d_in8) ? d_in : d_out, d_out, i + 1 == nf, stream)) return rc;
		}
		if (fp && be_motion_filter(d_out, mf, (uint64_t)span, d_coeffs_coded, stream)) return fail(-4, "filter launch failed");
		for (size_t i = 0; i < ni; i++)
			if (int rc = run_pass<float>(inv, inv->passes[i], (const float *)d_out, d_out, i + 1 == ni, stream)) return rc;
		if (d_out8 && be_region_f32_to_u8(d_out8, d_out, mul8, ni3, sw3, sw3, stream)) return fail(-4, "launch failed");
		return 0;
	}
	if (d_in8 || d_out8) {
		// the 8-bit buffers share the plans' element layout; the unfused conversions below walk whole spans
		if (nf < 2 || ni < 2) return fail(-2, "8-bit roundtrip needs at least two transformed axes");
		long long ispan = 1;
		for (int a = 0; a < fwd->rank; a++) ispan += (long long)(fwd->n[a] - 1) * fwd->axes[a].is;
		for (const Dim &b : fwd->batches) ispan += (long long)(b.n - 1) * b.is;
		if (d_out8 && !pass_has_u8(inv, inv->passes[ni - 1])) {
			// the output conversion will be one sweep over the whole span, which must then hold nothing but samples
			long long dense = 1;
			for (int a = 0; a < fwd->rank; a++) dense *= fwd->n[a];
			for (const Dim &b : fwd->batches) dense *= b.n;
			if (dense != span) return fail(-2, "8-bit output without a planar specialised row pass needs a dense work layout");
		}
		if (d_in8 && !pass_has_u8(fwd, fwd->passes[0])) {
			if (ispan != span) return fail(-2, "8-bit input without a planar specialised row pass needs identical input and work layouts");
			if (be_u8_to_f32(d_out, d_in8, (uint64_t)span, stream)) return fail(-4, "launch failed");
			d_in = d_out; d_in8 = nullptr;
		}
	}
	if (may_slice && d_in8 && d_out8 && pass_has_u8(fwd, fwd->passes[0]) && pass_has_u8(inv, inv->passes[ni - 1]) && F.axis == I.axis &&
	    (15u & (uintptr_t)d_out) == 0 && !(getenv("DSPFFT_NO_FUSED_ROUNDTRIP") && *getenv("DSPFFT_NO_FUSED_ROUNDTRIP") == '1')) {
		const int rc = roundtrip_sliced(fwd, inv, d_out, d_in8, d_out8, mul8, fp, d_coeffs_coded, stream);
		if (rc) return rc < 0 ? rc : 0;
	}
	for (size_t i = 0; i + 1 < nf; i++) {
		const Pass &P = fwd->passes[i];
		if (i == 0 && d_in8) {
			U8IO io; io.in = d_in8; io.out = nullptr; io.mul = 1.0;
			if (int rc = run_pass_u8(fwd, P, d_out, d_out, false, io, stream)) return rc;
			continue;
		}
		if (int rc = run_pass<float>(fwd, P, P.first ? d_in : d_out, d_out, false, stream)) return rc;
	}
	const float *src = nf == 1 ? d_in : d_out;
	const bool listed = F.has_spec && I.has_spec && F.spec.id == I.spec.id;
	const bool compiled = !listed && F.jit && I.jit && F.jit_fn_rt && F.jit_type == I.jit_type;      // kernels compiled at plan time (jit_kernels.h)
	const bool fusable = F.type == Pass::COL && I.type == Pass::COL && (listed || compiled) && F.spec_nwg == I.spec_nwg &&
	                     F.hostloop.empty() && I.hostloop.empty() && (15u & ((uintptr_t)src | (uintptr_t)d_out)) == 0 &&
	                     !(getenv("DSPFFT_NO_FUSED_ROUNDTRIP") && *getenv("DSPFFT_NO_FUSED_ROUNDTRIP") == '1');
	PassArgs af, ai;
	if (fusable) {
		fill_args(af, F.spa, fwd, F, src, d_out, fwd->scale, Fuse());
		fill_args(ai, I.spa, inv, I, (const float *)d_out, d_out, ni == 1 ? inv->scale : 1.0, Fuse());
	}
	if (fusable && is_plain_args(af) && is_plain_args(ai)) {      // (the fused kernel is the plain instantiation of both passes)
		if (compiled) {
			// parameters: (PassArgs af, PassArgs ai, FilterOp filt, unsigned long long *coded); FilterOp is the MotionFilter, nothing else
			void *args[4] = {&af, &ai, &mf, &d_coeffs_coded};
			if (int rc = be_jit_launch_n(F.jit_fn_rt, args, F.spec_nwg, F.jit_nthr, stream)) return fail(-4, "kernel launch failed (fused roundtrip, compiled at plan time): backend code %d", rc);
		} else if (int rc = be_launch_roundtrip(F.spec.id, af, ai, mf, d_coeffs_coded, F.spec_nwg, stream)) return fail(-4, "kernel launch failed (fused roundtrip): backend code %d", rc);
	} else {
		if (int rc = run_pass<float>(fwd, F, src, d_out, true, stream)) return rc;
		if (fp && be_motion_filter(d_out, mf, (uint64_t)span, d_coeffs_coded, stream)) return fail(-4, "filter launch failed");
		if (int rc = run_pass<float>(inv, I, (const float *)d_out, d_out, ni == 1, stream)) return rc;
	}
	for (size_t i = 1; i < ni; i++) {
		const Pass &P = inv->passes[i];
		if (i + 1 == ni && d_out8 && pass_has_u8(inv, P)) {
			U8IO io; io.in = nullptr; io.out = d_out8; io.mul = mul8;
			if (int rc = run_pass_u8(inv, P, d_out, d_out, true, io, stream)) return rc;
			return 0;
		}
		if (int rc = run_pass<float>(inv, P, (const float *)d_out, d_out, i + 1 == ni, stream)) return rc;
	}
	if (d_out8 && be_f32_to_u8(d_out8, d_out, mul8, (uint64_t)span, stream)) return fail(-4, "launch failed");    // no fused store: one sweep
	return 0;
}
}  // namespace

extern "C" int dspfft_execute_roundtrip(dspfft_plan fwd, dspfft_plan inv, const float *d_in, float *d_out,
                                        const dspfft_motion_filter_params *fp, unsigned long long *d_coeffs_coded, void *stream)
{
	if (!d_in) return fail(-1, "null plan or buffer");
	return roundtrip_core(fwd, inv, d_in, d_out, nullptr, nullptr, 1.0, fp, d_coeffs_coded, stream);
}

extern "C" int dspfft_execute_roundtrip_u8(dspfft_plan fwd, dspfft_plan inv, const uint8_t *d_in, uint8_t *d_out, float *d_work, double out_mul,
                                           const dspfft_motion_filter_params *fp, unsigned long long *d_coeffs_coded, void *stream)
{
	if (!d_in || !d_out || !d_work) return fail(-1, "null plan or buffer");
	return roundtrip_core(fwd, inv, nullptr, d_work, d_in, d_out, out_mul, fp, d_coeffs_coded, stream);
}

extern "C" int dspfft_scan_zigzag_frame_ids(uint32_t *d_ids, uint32_t w, uint32_t h, uint64_t step, void *s)
{
	if (!d_ids || !w || !h || !step) return fail(-1, "bad arguments");
	if ((uint64_t)w * h > 0xffffffffull) return fail(-1, "image too large for 32-bit offsets");
	return be_scan_zigzag_frame_ids(d_ids, w, h, step, s) ? fail(-4, "launch failed") : 0;
}

extern "C" int dspfft_plan_num_passes(dspfft_plan pl) { return pl ? (int)pl->passes.size() : 0; }

extern "C" int dspfft_execute_pass(dspfft_plan pl, int index, const float *d_in, float *d_out, void *stream)
{
	if (!pl || !d_in || !d_out || index < 0 || index >= (int)pl->passes.size()) return fail(-1, "bad plan, buffer or pass index");
	if (pl->f64) return fail(-1, "dspfft_execute_pass takes f32 plans");
	const std::vector<Pass> &passes = pick_passes(pl, d_in, d_out);
	const Pass &P = passes[index];
	return run_pass<float>(pl, P, P.first ? d_in : d_out, d_out, index + 1 == (int)passes.size(), stream);
}

// one pass over the items; items [timed_item, timed_item + timed_count) bracket each of their passes with pass_events (2 per pass).
// Buffers are float or double according to each item's plan.
template <class R>
static int execute_item(dspfft_plan pl, const void *in, void *out, void *st, bool timed, void *const *pass_events, int &ev)
{
	const R *d_in = (const R *)in;
	R *d_out = (R *)out;
	if (!timed) return execute_t<R>(pl, d_in, d_out, st);
	const std::vector<Pass> &passes = pick_passes(pl, d_in, d_out);
	for (size_t p = 0; p < passes.size(); p++) {
		const Pass &P = passes[p];
		if (be_event_record(pass_events[ev++], st)) return fail(-4, "event record failed");
		if (int rc = run_pass<R>(pl, P, P.first ? d_in : d_out, d_out, p + 1 == passes.size(), st)) return rc;
		if (be_event_record(pass_events[ev++], st)) return fail(-4, "event record failed");
	}
	return 0;
}
// double frames on more than one stream: the 3840 x 3 row pass goes back to one workgroup per line for the batch (backend.h)
struct ChanLinesGuard {
	bool on = false;
	ChanLinesGuard(int count, const dspfft_plan *plans, void *const *streams)
	{
		void *first = nullptr; bool have = false;
		for (int i = 0; streams && i < count && !on; i++) {
			if (!plans[i] || !plans[i]->f64) continue;
			if (!have) { first = streams[i]; have = true; } else if (streams[i] != first) on = true;
		}
		if (on) g_chan_lines_suspended++;
	}
	~ChanLinesGuard() { if (on) g_chan_lines_suspended--; }
};
static int execute_many_once(int count, const dspfft_plan *plans, const void *const *d_in, void *const *d_out, void *const *streams,
                             int timed_item, int timed_count, void *const *pass_events)
{
	ChanLinesGuard guard(count, plans, streams);
	int ev = 0;
	for (int i = 0; i < count; i++) {
		dspfft_plan pl = plans[i];
		void *st = streams ? streams[i] : nullptr;
		const bool timed = pass_events && i >= timed_item && i < timed_item + timed_count;
		const int rc = pl->f64 ? execute_item<double>(pl, d_in[i], d_out[i], st, timed, pass_events, ev) : execute_item<float>(pl, d_in[i], d_out[i], st, timed, pass_events, ev);
		if (rc) return rc;
	}
	return 0;
}
static bool plan_is_staged(const dspfft_plan_s *pl)
{
	for (const Pass &P : pl->passes) if (P.tab.stage) return true;
	return false;
}
static int check_many(int count, const dspfft_plan *plans, const void *const *d_in, void *const *d_out, void *const *streams, int timed_item, int timed_count, bool events, bool f64_ok)
{
	if (count < 0 || (count && (!plans || !d_in || !d_out))) return fail(-1, "bad arguments");
	for (int i = 0; i < count; i++) {
		if (!plans[i] || !d_in[i] || !d_out[i]) return fail(-1, "item %d: null plan or buffer", i);
		if (plans[i]->f64 && !f64_ok) return fail(-1, "dspfft_execute_many takes f32 plans");
		// a staged DENSE pass (lines beyond what LDS holds) copies through ONE array owned by the plan: the same plan on two streams at once
		// would race on it
		if (streams && plan_is_staged(plans[i]))
			for (int j = 0; j < i; j++)
				if (plans[j] == plans[i] && streams[j] != streams[i]) return fail(-1, "items %d and %d: a plan with a staged pass (its lines pass through a device array of the plan) cannot run on two streams at once", j, i);
	}
	if (events) {
		if (timed_item < 0 || timed_count < 0 || timed_item + timed_count > count) return fail(-1, "timed items [%d, %d) lie outside the batch of %d", timed_item, timed_item + timed_count, count);
		// a one-pass small-block plan runs ONE fused kernel in dspfft_execute, not its axis passes: bracketing "passes" would time other kernels
		for (int i = timed_item; i < timed_item + timed_count; i++)
			if (plans[i]->has_block) return fail(-1, "item %d: per-pass events are not available for one-pass block plans", i);
	}
	return 0;
}

extern "C" int dspfft_execute_many(int count, const dspfft_plan *plans, const float *const *d_in, float *const *d_out, void *const *streams,
                                   int timed_item, int timed_count, void *const *pass_events)
{
	if (int rc = check_many(count, plans, (const void *const *)d_in, (void *const *)d_out, streams, timed_item, timed_count, pass_events != nullptr, false)) return rc;
	return execute_many_once(count, plans, (const void *const *)d_in, (void *const *)d_out, streams, timed_item, timed_count, pass_events);
}

// The frame loop of a clip inside the library: the batch `repeats` times (motion/motion.c:613-753 runs its per-frame plans once per
// frame of the clip, scan/scan.c:421-447 its inverse plan once per output frame).  See include/dspfft.h.
extern "C" int dspfft_execute_many_repeat(int count, const dspfft_plan *plans, const void *const *d_in, void *const *d_out, void *const *streams,
                                          int repeats, int rejoin_every, int timed_every, int timed_count, void *const *pass_events)
{
	if (repeats < 0 || rejoin_every < 0 || timed_every < 0) return fail(-1, "bad arguments");
	const bool ev = pass_events && timed_every > 0 && timed_count > 0;
	if (ev && count % timed_count) return fail(-1, "the timed window (%d items) must divide the batch (%d items)", timed_count, count);
	if (int rc = check_many(count, plans, d_in, d_out, streams, 0, ev ? count : 0, ev, true)) return rc;
	// distinct streams of the batch, in order of first use
	std::vector<void *> uniq;
	for (int i = 0; i < count; i++) { void *st = streams ? streams[i] : nullptr; if (std::find(uniq.begin(), uniq.end(), st) == uniq.end()) uniq.push_back(st); }
	std::vector<void *> join;
	const bool rejoin = rejoin_every > 0 && uniq.size() > 1;
	if (rejoin) for (size_t i = 0; i < uniq.size(); i++) { void *e = be_order_event_create(); if (!e) { for (void *x : join) be_event_destroy(x); return fail(-4, "event creation failed"); } join.push_back(e); }
	int rc = 0, windows = 0;
	size_t ev_off = 0;
	for (int k = 0; k < repeats && !rc; k++) {
		if (rejoin && k && k % rejoin_every == 0) {
			// every stream waits for where every other stream stood at the end of the previous repeat: their relative phase cannot drift
			for (size_t a = 0; a < uniq.size() && !rc; a++) if (be_event_record(join[a], uniq[a])) rc = fail(-4, "event record failed");
			for (size_t a = 0; a < uniq.size() && !rc; a++)
				for (size_t b = 0; b < uniq.size() && !rc; b++)
					if (a != b && be_stream_wait_event(uniq[a], join[b])) rc = fail(-4, "stream wait failed");
			if (rc) break;
		}
		if (ev && k % timed_every == 0) {
			const int first = (int)(((long long)windows * timed_count) % count);
			rc = execute_many_once(count, plans, d_in, d_out, streams, first, timed_count, pass_events + ev_off);
			for (int i = first; i < first + timed_count; i++) ev_off += 2 * pick_passes(plans[i], d_in[i], d_out[i]).size();    // (alignment of the pointers only)
			windows++;
		} else rc = execute_many_once(count, plans, d_in, d_out, streams, 0, 0, nullptr);
	}
	for (void *x : join) be_event_destroy(x);
	return rc;
}
extern "C" void *dspfft_stream_create(void) { return be_stream_create(); }
extern "C" void dspfft_stream_destroy(void *s) { be_stream_destroy(s); }
extern "C" int dspfft_stream_synchronize(void *s) { return be_stream_synchronize(s) ? fail(-4, "stream synchronise failed") : 0; }
extern "C" void *dspfft_event_create(void) { return be_event_create(); }
extern "C" void dspfft_event_destroy(void *e) { be_event_destroy(e); }
extern "C" int dspfft_event_synchronize(void *e) { return e && !be_event_synchronize(e) ? 0 : fail(-4, "event synchronise failed"); }
extern "C" int dspfft_event_elapsed_ms(void *a, void *b, float *ms)
{
	if (!a || !b || !ms) return fail(-1, "bad arguments");
	return be_event_elapsed_ms(a, b, ms) ? fail(-4, "event query failed") : 0;
}

extern "C" void dspfft_destroy_plan(dspfft_plan pl)
{
	if (!pl) return;
	for (Pass &P : pl->passes) P.tab.release();
	for (Pass &P : pl->split) P.tab.release();
	be_free(pl->zflags); be_free(pl->zpage); be_free(pl->zranges); be_free(pl->eids);
	for (dspfft_plan_s::RtSlices &r : pl->rt_slices) {
		dspfft_destroy_plan(r.fwd); dspfft_destroy_plan(r.inv); dspfft_destroy_plan(r.fwd_rem); dspfft_destroy_plan(r.inv_rem);
		if (r.side) be_stream_destroy(r.side);
		if (r.ev_fork) be_event_destroy(r.ev_fork);
		if (r.ev_join) be_event_destroy(r.ev_join);
	}
	delete pl;
}

extern "C" int dspfft_plan_describe(dspfft_plan pl, char *buf, size_t buflen)
{
	if (!pl || !buf || !buflen) return fail(-1, "bad arguments");
	std::string s = std::string("backend ") + be_name() + "\n";
	// a split plan lists the passes dspfft_execute runs; the plain ones (masked / fused executions, unaligned buffers) follow
	for (const Pass &P : pl->split) { s += P.desc; s += "\n"; }
	if (pl->has_block) { s += pl->blk_desc; s += "\n"; }
	for (const Pass &P : pl->passes) { if (!pl->split.empty() || pl->has_block) s += "plain "; s += P.desc; if (!P.hostloop.empty()) s += " +hostloop"; s += "\n"; }
	for (const dspfft_plan_s::RtSlices &r : pl->rt_slices) {          // (present once dspfft_execute_roundtrip_u8 has walked a clip in slices)
		char b[256];
		snprintf(b, sizeof b, "roundtrip_u8 in slices of %d frames (last: %d) on %d stream(s): ", r.frames, r.fwd_rem ? r.fwd_rem->howmany : r.frames, r.side ? 2 : 1);
		s += b; s += r.fwd->passes.back().desc; s += "\n";
	}
	snprintf(buf, buflen, "%s", s.c_str());
	return 0;
}

extern "C" size_t dspfft_plan_algorithmic_bytes(dspfft_plan pl) { return pl ? pl->alg_bytes : 0; }
#undef g_plan_effort
extern "C" void dspfft_set_plan_effort(int effort) { g_plan_effort_default.store(effort < 0 ? 0 : effort, std::memory_order_relaxed); }
extern "C" int dspfft_get_plan_effort(void) { return t_plan_effort >= 0 ? t_plan_effort : g_plan_effort_default.load(std::memory_order_relaxed); }
extern "C" void dspfft_set_thread_plan_effort(int effort) { t_plan_effort = effort < 0 ? -1 : effort; }
extern "C" int dspfft_get_thread_plan_effort(void) { return t_plan_effort; }

extern "C" int dspfft_scan_zigzag(uint32_t *d_lin, uint32_t w, uint32_t h, uint64_t first, uint64_t count, void *s)
{
	if (!d_lin || !w || !h || first + count > (uint64_t)w * h) return fail(-1, "bad zigzag range");
	if ((uint64_t)w * h > 0xffffffffull) return fail(-1, "image too large for 32-bit offsets");
	return be_scan_zigzag(d_lin, w, h, first, count, s) ? fail(-4, "launch failed") : 0;
}
// ---- scan methods other than zigzag (scan_core.h) ----
#include "scan_core.h"
extern "C" uint64_t dspfft_scan_limit(int method, uint32_t w32, uint32_t h32)
{
	const uint64_t w = w32, h = h32;
	switch (method) {
	case SCANM_ROW: return h;
	case SCANM_COLUMN: return w;
	case SCANM_DIAGONAL: return w + h - 1;
	case SCANM_MIRROR: case SCANM_BOX: return std::max(w, h);
	case SCANM_IBOX: return std::min(w, h);
	case SCANM_RADIAL: case SCANM_IRADIAL: return rint_hypot(w - 1, h - 1) + 1;      // the farthest corner's bucket is the last one
	case SCANM_HORIZONTAL: case SCANM_VERTICAL: case SCANM_ZIGZAG: return w * h;
	default: return 0;
	}
}
extern "C" uint64_t dspfft_scan_max_interval(int method, uint32_t w32, uint32_t h32)
{
	const uint64_t w = w32, h = h32;
	switch (method) {
	case SCANM_ROW: return w;
	case SCANM_COLUMN: return h;
	case SCANM_DIAGONAL: return std::min(w, h);
	case SCANM_MIRROR: return std::min(w, h) * 2 - 1;
	case SCANM_BOX: case SCANM_IBOX: return w + h - 1;                                // limit_sum (scan_methods.c:23,496,502)
	case SCANM_HORIZONTAL: case SCANM_VERTICAL: case SCANM_ZIGZAG: return 1;
	default: return 0;                                                                // radial / iradial: no closed form (owner ids only)
	}
}
// entries per scan index in a coordinate list: max_interval, plus the one entry the reference's callers over-allocate (scan.c:346) and
// ibox needs (index 0 emits its corner twice: w + h coordinates against max_interval = w + h - 1)
extern "C" uint64_t dspfft_scan_coord_slots(int method, uint32_t w, uint32_t h)
{
	const uint64_t m = dspfft_scan_max_interval(method, w, h);
	return (method == SCANM_BOX || method == SCANM_IBOX) ? m + 1 : m;
}
static int scan_args_ok(const void *p, int method, uint32_t w, uint32_t h)
{
	if (!p || !w || !h || method < 0 || method >= SCANM_COUNT) return fail(-1, "bad scan arguments");
	if ((uint64_t)w * h > 0xfffffffeull) return fail(-1, "image too large for 32-bit offsets");
	return 0;
}
extern "C" int dspfft_scan_owner_index(uint32_t *d_index, int method, uint32_t w, uint32_t h, void *s)
{
	if (int rc = scan_args_ok(d_index, method, w, h)) return rc;
	if (method == SCANM_BOX) return fail(-2, "box has no single owner per pixel: use dspfft_scan_coords + dspfft_scan_stamp");
	if (method == SCANM_ZIGZAG) return be_scan_zigzag_frame_ids(d_index, w, h, 0, s) ? fail(-4, "launch failed") : 0;
	return be_scan_owner_index(d_index, method, w, h, 0, s) ? fail(-4, "launch failed") : 0;
}
extern "C" int dspfft_scan_frame_ids(uint32_t *d_ids, int method, uint32_t w, uint32_t h, uint64_t step, void *s)
{
	if (int rc = scan_args_ok(d_ids, method, w, h)) return rc;
	if (!step) return fail(-1, "step must be >= 1");
	if (method == SCANM_BOX) return fail(-2, "box has no single owner per pixel: use dspfft_scan_coords + dspfft_scan_stamp");
	if (method == SCANM_ZIGZAG) return be_scan_zigzag_frame_ids(d_ids, w, h, step, s) ? fail(-4, "launch failed") : 0;
	return be_scan_owner_index(d_ids, method, w, h, step, s) ? fail(-4, "launch failed") : 0;
}
extern "C" int dspfft_scan_coords(uint32_t *d_lin, int method, uint32_t w, uint32_t h, uint64_t first, uint64_t count, void *s)
{
	if (int rc = scan_args_ok(d_lin, method, w, h)) return rc;
	const uint64_t slots = dspfft_scan_coord_slots(method, w, h);
	if (!slots) return fail(-2, "radial / iradial have no closed-form coordinate lists: use dspfft_scan_frame_ids");
	if (first + count > dspfft_scan_limit(method, w, h)) return fail(-1, "scan index range out of bounds");
	if (method == SCANM_ZIGZAG) return be_scan_zigzag(d_lin, w, h, first, count, s) ? fail(-4, "launch failed") : 0;
	return be_scan_coords(d_lin, method, w, h, first, count, slots, s) ? fail(-4, "launch failed") : 0;
}
extern "C" int dspfft_scan_stamp(uint32_t *d_ids, const uint32_t *d_lin, uint64_t nslots, uint32_t frame_id, void *s)
{
	if (!d_ids || (!d_lin && nslots)) return fail(-1, "bad arguments");
	return be_scan_stamp(d_ids, d_lin, nslots, frame_id, s) ? fail(-4, "launch failed") : 0;
}
extern "C" int dspfft_scan_index_to_frame_ids(uint32_t *d_ids, uint64_t n, uint64_t step, void *s)
{
	if (!d_ids || !n || !step) return fail(-1, "bad arguments");
	return be_scan_index_to_frame_ids(d_ids, n, step, s) ? fail(-4, "launch failed") : 0;
}
extern "C" size_t dspfft_scan_magnitude_work_bytes(uint32_t w, uint32_t h) { return be_scan_magnitude_work_bytes(w, h); }
extern "C" int dspfft_scan_magnitude_index(uint32_t *d_index, const float *d_coeffs, uint32_t w, uint32_t h, int channels, double qfactor,
                                           void *d_work, size_t work_bytes, uint32_t *limit, void *s)
{
	if (!d_index || !d_coeffs || !d_work || !w || !h || channels < 1) return fail(-1, "bad arguments");
	if ((uint64_t)w * h > 0xfffffffeull) return fail(-1, "image too large for 32-bit offsets");
	if (work_bytes < be_scan_magnitude_work_bytes(w, h)) return fail(-1, "work buffer too small: dspfft_scan_magnitude_work_bytes");
	return be_scan_magnitude_index(d_index, d_coeffs, w, h, channels, qfactor, d_work, work_bytes, limit, s) ? fail(-4, "magnitude sort failed") : 0;
}

extern "C" int dspfft_scan_scatter(float *r, const float *c, const uint32_t *lin, uint64_t count, uint64_t npix, int ch, void *s)
{
	if (!r || !c || (!lin && count) || ch < 1) return fail(-1, "bad arguments");
	return be_scan_scatter(r, c, lin, count, npix, ch, s) ? fail(-4, "launch failed") : 0;
}
extern "C" int dspfft_accumulate(float *sum, const float *img, uint64_t len, void *s)
{
	if (!sum || !img) return fail(-1, "bad arguments");
	return be_accumulate(sum, img, len, s) ? fail(-4, "launch failed") : 0;
}
extern "C" int dspfft_broadcast_dc(float *sum, const float *c, uint64_t npix, int ch, void *s)
{
	if (!sum || !c || ch < 1) return fail(-1, "bad arguments");
	return be_broadcast_dc(sum, c, npix, ch, s) ? fail(-4, "launch failed") : 0;
}
extern "C" int dspfft_u8_to_f32(float *d, const uint8_t *src, uint64_t len, void *s)
{
	if (!d || !src) return fail(-1, "bad arguments");
	return be_u8_to_f32(d, src, len, s) ? fail(-4, "launch failed") : 0;
}
extern "C" int dspfft_f32_to_u8(uint8_t *d, const float *src, double mul, uint64_t len, void *s)
{
	if (!d || !src) return fail(-1, "bad arguments");
	return be_f32_to_u8(d, src, mul, len, s) ? fail(-4, "launch failed") : 0;
}
