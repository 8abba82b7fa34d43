// fftw_shim.hip -- FFTW3-named host-pointer adapter (include/fftw3.h) over the device engine.
//
// Contract reproduced from the reference's use of FFTW: the plan captures the raw
// host pointers (no execute_r2r "new array" calls exist in the tools); plan creation never touches
// the arrays, whatever the planner flag (scan/scan.c:354-359 zeroes `reconstruction` BEFORE planning
// with FFTW_MEASURE and relies on it staying zero); execute may be called any number of times
// (scan/scan.c:447).  A NULL plan is returned only for arguments FFTW would reject as well; the tools
// never check (spec/spec.c:63-64), so failures are also reported on stderr.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../../include/dspfft.h"
#include "../../include/fftw3.h"

namespace {

std::mutex g_mu;
std::unordered_set<void *> g_pinned;

void *pinned_alloc(size_t bytes)
{
	void *p = nullptr;
	// pinned host memory: the H2D/D2H copies of execute() run at PCIe rate without staging
	if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
		fprintf(stderr, "dspfft: hipHostMalloc(%zu) failed: no usable GPU?\n", bytes);
		return nullptr;
	}
	std::lock_guard<std::mutex> lk(g_mu);
	g_pinned.insert(p);
	return p;
}
void pinned_free(void *p)
{
	if (!p) return;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		auto it = g_pinned.find(p);
		if (it == g_pinned.end()) { free(p); return; }
		g_pinned.erase(it);
	}
	(void)hipHostFree(p);
}

size_t span(int rank, const int *n, const int *embed, int stride, int dist, int howmany)
{
	size_t idx = 0;
	for (int a = 0; a < rank; a++) idx = idx * (size_t)(embed ? embed[a] : n[a]) + (size_t)(n[a] - 1);
	return idx * (size_t)stride + (size_t)(howmany - 1) * (size_t)dist + 1;
}

// ---- sparse upload -------------------------------------------------------------------------------------------------------------------
// scan/scan.c:429-447 clears `reconstruction` and sets one step's coefficients (<= 3 % of an 8K frame) before every fftw_execute: the dense upload moves
// 398 MB over a 57 GB/s link for 12-30 MB of data.  Host threads read memory 3-4x faster than the link carries it (1.7 ms for those 398 MB on 12-16 threads,
// 7.0 ms by hipMemcpy), so execute() looks first: 4 KB blocks that hold anything but zero bits are packed into a pinned staging buffer and put in place on the
// device by one small kernel behind a hipMemsetAsync.  -0.0f has a bit set and travels; what arrives is byte for byte what the dense copy delivers.
constexpr size_t kUpBlock = 4096;                 // bytes per block = one 256-thread workgroup x 16 B
constexpr size_t kUpChunk = 256;                  // blocks a thread takes at a time (1 MB): ranges interleave so that a dense corner spreads over all threads
constexpr size_t kUpMinBytes = (size_t)32 << 20;  // below this the dense copy takes < 0.6 ms
constexpr int kUpFillPct = 25;                    // denser than this (per thread): plain hipMemcpyAsync
constexpr size_t kUpMaxStage = (size_t)1 << 30;   // pinned + device staging per plan, at most

__global__ __launch_bounds__(256) void scatter_blocks_kernel(const uint4 *__restrict__ stage, const uint32_t *__restrict__ index, uint4 *__restrict__ dst)
{
	const uint32_t b = index[blockIdx.x];
	if (b == 0xFFFFFFFFu) return;                 // unused tail of a thread's share
	dst[(size_t)b * (kUpBlock / 16) + threadIdx.x] = stage[(size_t)blockIdx.x * (kUpBlock / 16) + threadIdx.x];
}

int upload_threads()
{
	// DSPFFT_UPLOAD_THREADS is read at every execute (0 = always the dense copy); the default is found once
	if (const char *e = getenv("DSPFFT_UPLOAD_THREADS")) { const int t = atoi(e); return t > 64 ? 64 : t; }
	static const int n = [] {
		int t = (int)std::thread::hardware_concurrency();
		// a container's CPU quota, not the machine's core count, is what the threads get (cgroup v2: "<quota> <period>" or "max <period>")
		if (FILE *fp = fopen("/sys/fs/cgroup/cpu.max", "r")) {
			long long q = 0, per = 0;
			if (fscanf(fp, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0 && (int)(q / per) < t) t = (int)(q / per);
			fclose(fp);
		}
		return t > 12 ? 12 : t;          // 12 threads read a pinned array as fast as 16 or 24 (profiles/r06_scan_gpu_8k.txt)
	}();
	return n;
}

std::atomic<unsigned long long> g_sparse_uploads{0};
// DSPFFT_UPLOAD_DEBUG=1: one line per large execute on stderr (which way the input went up, and why)
bool upload_debug() { static const bool on = [] { const char *e = getenv("DSPFFT_UPLOAD_DEBUG"); return e && *e == '1'; }(); return on; }

struct SparseUp {
	int threads = 0;
	size_t nblk = 0, share = 0;                   // whole blocks of the array; staging blocks per thread
	char *h_stage = nullptr, *d_stage = nullptr;
	uint32_t *h_index = nullptr, *d_index = nullptr;
	unsigned skip = 0, dense_streak = 0;          // dense inputs: look again only after 1, 2, 4 ... 64 executes
	~SparseUp() { (void)hipHostFree(h_stage); (void)hipHostFree(h_index); (void)hipFree(d_stage); (void)hipFree(d_index); }
};

struct Shim {
	SparseUp *up = nullptr;
	dspfft_plan plan = nullptr;
	void *h_in = nullptr, *h_out = nullptr;
	size_t in_len = 0, out_len = 0;          // elements
	bool out_dense = false;                  // the transform writes every element of [0, out_len): `out` needs no upload
	size_t es = 4;                           // bytes per element: 4 (fftwf_) or 8 (fftw_)
	void *d_in = nullptr, *d_out = nullptr;  // d_out == d_in when in-place
	hipStream_t stream = nullptr;
	int device = 0;                          // the device current at plan time: the stream's
};

Shim *make_plan(int rank, const int *n, int howmany, void *in, const int *inembed, int istride, int idist,
                void *out, const int *onembed, int ostride, int odist, const int *kinds, bool f64, unsigned flags)
{
	if (!in || !out || !n || !kinds) { fprintf(stderr, "dspfft: plan_many_r2r: null argument\n"); return nullptr; }
	Shim *s = new Shim();
	s->es = f64 ? sizeof(double) : sizeof(float); s->h_in = in; s->h_out = out;
	// FFTW_ESTIMATE = plan fast; FFTW_MEASURE / PATIENT / EXHAUSTIVE (scan.c:359, motion.c:93-103) = this plan will run many times:
	// frame sizes without a listed specialised kernel get one compiled now (dspfft_set_thread_plan_effort: the calling thread's own override of the
	// process-wide effort, set and restored around this one plan -- concurrent planning from other threads is not affected)
	const int effort_before = dspfft_get_thread_plan_effort();
	dspfft_set_thread_plan_effort((flags & FFTW_ESTIMATE) ? 0 : (flags & (FFTW_PATIENT | FFTW_EXHAUSTIVE)) ? 2 : 1);
	// fftw_ (double) plans compute in double on the device, as the reference's default build does on the CPU
	const int rc = f64 ? dspfft_plan_many_r2r_f64(&s->plan, rank, n, howmany, inembed, istride, idist, onembed, ostride, odist, kinds)
	                   : dspfft_plan_many_r2r(&s->plan, rank, n, howmany, inembed, istride, idist, onembed, ostride, odist, kinds);
	dspfft_set_thread_plan_effort(effort_before);
	if (rc) {
		fprintf(stderr, "dspfft: plan_many_r2r failed: %s\n", dspfft_last_error());
		delete s; return nullptr;
	}
	s->in_len = span(rank, n, inembed, istride, idist, howmany);
	s->out_len = span(rank, n, onembed, ostride, odist, howmany);
	size_t written = (size_t)howmany;
	for (int a = 0; a < rank; a++) written *= (size_t)n[a];
	s->out_dense = written == s->out_len;     // as many outputs as the span holds: no embedding gaps (scan/scan.c:359: onembed == NULL)
	// in place: ONE device buffer that covers both layouts (they may differ: inembed != onembed on the same array)
	if (in == out) { const size_t both = s->in_len > s->out_len ? s->in_len : s->out_len; s->in_len = both; if (!s->out_dense) s->out_len = both; }
	bool ok = hipMalloc(&s->d_in, s->in_len * s->es) == hipSuccess;
	if (in == out) s->d_out = s->d_in;
	else ok = ok && hipMalloc(&s->d_out, s->out_len * s->es) == hipSuccess;
	ok = ok && hipStreamCreate(&s->stream) == hipSuccess && hipGetDevice(&s->device) == hipSuccess;
	if (!ok) {
		fprintf(stderr, "dspfft: device allocation failed (%zu + %zu elements)\n", s->in_len, s->out_len);
		dspfft_destroy_plan(s->plan); (void)hipFree(s->d_in); if (s->d_out != s->d_in) (void)hipFree(s->d_out);
		delete s; return nullptr;
	}
	return s;
}

// h_in -> d_in.  Returns false on a HIP error.
bool upload_in(Shim *s)
{
	const size_t bytes = s->in_len * s->es;
	const auto dense = [&] { return hipMemcpyAsync(s->d_in, s->h_in, bytes, hipMemcpyHostToDevice, s->stream) == hipSuccess; };
	int T = upload_threads();
	if (bytes < kUpMinBytes || T < 4 || ((uintptr_t)s->h_in & 15)) return dense();     // < 4 threads read no faster than the link
	if (s->up && s->up->threads) T = s->up->threads;                                    // the staging buffers are laid out for the first execute's team
	if (!s->up) {
		SparseUp *u = new SparseUp();
		u->threads = T; u->nblk = bytes / kUpBlock;
		const size_t mine = (u->nblk + T - 1) / T;                                      // blocks a thread looks at, at most (+ one chunk of imbalance)
		u->share = (mine + kUpChunk) * kUpFillPct / 100 + 1;
		if (u->share * T * kUpBlock > kUpMaxStage) u->share = kUpMaxStage / kUpBlock / T;   // (a larger array counts as dense at a lower fill)
		const size_t cap = u->share * T;
		if (hipHostMalloc((void **)&u->h_stage, cap * kUpBlock, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&u->h_index, cap * 4, hipHostMallocDefault) != hipSuccess ||
		    hipMalloc((void **)&u->d_stage, cap * kUpBlock) != hipSuccess || hipMalloc((void **)&u->d_index, cap * 4) != hipSuccess) {
			(void)hipGetLastError(); delete u; u = new SparseUp();                      // no room for the staging buffers: dense from now on
			u->skip = ~0u;
		}
		s->up = u;
	}
	SparseUp *u = s->up;
	if (u->skip) { if (upload_debug()) fprintf(stderr, "dspfft: upload dense: not looked at (%u more)\n", u->skip - 1); if (u->skip != ~0u) u->skip--; return dense(); }
	const auto back_off = [&] { u->skip = 1u << (u->dense_streak < 6 ? u->dense_streak : 6); u->dense_streak++; };
	{
		// 128 blocks across the array first (microseconds): an image or a spectrum has something in most of them and is not worth a team.
		// One block from each 128th of the array, at a scrambled place in it: evenly spaced samples walk a diagonal of the frame, and a zigzag
		// scan's band IS a diagonal (511 rows: 127 of 128 even samples fell into a band that covers a twenty-fourth of the frame)
		size_t set = 0;
		uint32_t lcg = 12345u;
		for (size_t k = 0; k < 128; k++) {
			lcg = lcg * 1664525u + 1013904223u;
			const size_t lo = u->nblk * k / 128, hi = u->nblk * (k + 1) / 128;
			const uint64_t *q = (const uint64_t *)((const char *)s->h_in + (lo + (hi > lo ? (lcg >> 8) % (hi - lo) : 0)) * kUpBlock);
			uint64_t any = 0;
			for (size_t i = 0; i < kUpBlock / 8; i++) any |= q[i];
			set += any != 0;
		}
		if (set > 64) { if (upload_debug()) fprintf(stderr, "dspfft: upload dense: %zu of 128 sampled blocks are set\n", set); back_off(); return dense(); }
	}
	// the device side clears while the host looks (a dense frame overwrites it all the same)
	const auto t_set = std::chrono::steady_clock::now();
	bool ok = hipMemsetAsync(s->d_in, 0, u->nblk * kUpBlock, s->stream) == hipSuccess;
	const double set_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_set).count();
	std::atomic<bool> full{false}, failed{false};
	std::vector<size_t> used(T, 0);
	const auto look = [&](int t) {
		(void)hipSetDevice(s->device);
		const size_t at = (size_t)t * u->share;
		char *stage = u->h_stage + at * kUpBlock;
		uint32_t *index = u->h_index + at;
		size_t n = 0, sent = 0;
		// what is packed goes up while the rest is still being read: a copy per half megabyte, enqueued by the thread that packed it
		const auto send = [&] {
			if (n > sent && hipMemcpyAsync(u->d_stage + (at + sent) * kUpBlock, stage + sent * kUpBlock, (n - sent) * kUpBlock, hipMemcpyHostToDevice, s->stream) != hipSuccess)
				failed.store(true, std::memory_order_relaxed);
			sent = n;
		};
		for (size_t c = (size_t)t * kUpChunk; c < u->nblk && !full.load(std::memory_order_relaxed); c += (size_t)T * kUpChunk) {
			const size_t end = c + kUpChunk < u->nblk ? c + kUpChunk : u->nblk;
			for (size_t b = c; b < end; b++) {
				const uint64_t *q = (const uint64_t *)((const char *)s->h_in + b * kUpBlock);
				uint64_t any = 0;
				for (size_t i = 0; i < kUpBlock / 8; i++) any |= q[i];
				if (!any) continue;
				if (n == u->share) { full.store(true, std::memory_order_relaxed); return; }
				memcpy(stage + n * kUpBlock, q, kUpBlock);
				index[n++] = (uint32_t)b;
			}
			if (n - sent >= 128) send();
		}
		send();
		used[t] = n;
		for (size_t i = n; i < u->share; i++) index[i] = 0xFFFFFFFFu;
	};
	const auto t_look = std::chrono::steady_clock::now();
	{
		// (a fresh team per execute.  Threads parked on a condition variable between executes were SLOWER here: 11.3-14.0 against 10.5-11.1 ms per 8K frame,
		// alternating runs on one box -- woken threads resume on the cores they slept on, new ones are placed on idle cores.)
		std::vector<std::thread> fresh;
		for (int t = 1; t < T; t++) fresh.emplace_back(look, t);
		look(0);
		for (auto &th : fresh) th.join();
	}
	const double look_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_look).count();
	if (failed.load()) return false;
	if (full.load()) {
		// (staged blocks already on their way are never put in place: the scatter kernel is not launched)
		if (upload_debug()) fprintf(stderr, "dspfft: upload dense: a thread found more than %zu of its blocks set\n", u->share);
		back_off();
		return dense();
	}
	if (upload_debug()) { size_t n = 0; for (size_t v : used) n += v; fprintf(stderr, "dspfft: upload sparse: %zu of %zu blocks on %d threads, read in %.3f ms (memset enqueued in %.3f ms)\n", n, u->nblk, T, look_ms, set_ms); }
	u->dense_streak = 0;
	g_sparse_uploads++;
	size_t top = 0;                                                                      // staging blocks up to the last share in use
	for (int t = 0; t < T; t++) if (used[t]) top = ((size_t)t + 1) * u->share;
	if (top && ok) {
		ok = hipMemcpyAsync(u->d_index, u->h_index, top * 4, hipMemcpyHostToDevice, s->stream) == hipSuccess;
		hipLaunchKernelGGL(scatter_blocks_kernel, dim3((unsigned)top), dim3(256), 0, s->stream, (const uint4 *)u->d_stage, (const uint32_t *)u->d_index, (uint4 *)s->d_in);
		ok = ok && hipGetLastError() == hipSuccess;
	}
	// what does not fill a block: as it is
	if (const size_t tail = bytes - u->nblk * kUpBlock)
		ok = ok && hipMemcpyAsync((char *)s->d_in + u->nblk * kUpBlock, (const char *)s->h_in + u->nblk * kUpBlock, tail, hipMemcpyHostToDevice, s->stream) == hipSuccess;
	return ok;
}

void run(Shim *s)
{
	if (!s) { fprintf(stderr, "dspfft: execute on a NULL plan\n"); return; }
	const auto t_run = std::chrono::steady_clock::now();
	bool ok = upload_in(s);
	const double up_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run).count();
	// out-of-place with embedding gaps: elements the transform does not write must survive in `out`, so the host's `out` goes up
	// first.  A dense output (scan's per-frame plan, scan/scan.c:359,447) is overwritten whole: two transfers per execute, not three.
	if (s->d_out != s->d_in && !s->out_dense) ok = ok && hipMemcpyAsync(s->d_out, s->h_out, s->out_len * s->es, hipMemcpyHostToDevice, s->stream) == hipSuccess;
	if (ok) {
		const int rc = s->es == 8 ? dspfft_execute_f64(s->plan, (const double *)s->d_in, (double *)s->d_out, s->stream)
		                          : dspfft_execute(s->plan, (const float *)s->d_in, (float *)s->d_out, s->stream);
		if (rc) {
			// the uploads (and whatever passes were enqueued) still read the caller's arrays: drain the stream before handing them back
			fprintf(stderr, "dspfft: execute failed: %s\n", dspfft_last_error());
			(void)hipStreamSynchronize(s->stream);
			return;
		}
	}
	const bool dbg = upload_debug() && s->in_len * s->es >= kUpMinBytes;
	const auto t_up = std::chrono::steady_clock::now();
	if (dbg) (void)hipStreamSynchronize(s->stream);
	const auto t_dn = std::chrono::steady_clock::now();
	ok = ok && hipMemcpyAsync(s->h_out, s->d_out, s->out_len * s->es, hipMemcpyDeviceToHost, s->stream) == hipSuccess;
	ok = ok && hipStreamSynchronize(s->stream) == hipSuccess;
	if (dbg) fprintf(stderr, "dspfft: execute: upload enqueued in %.3f ms, passes by %.3f ms, uploads + passes drained %.3f ms later, download %.3f ms\n", up_ms,
	                 std::chrono::duration<double, std::milli>(t_up - t_run).count(), std::chrono::duration<double, std::milli>(t_dn - t_up).count(),
	                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dn).count());
	if (!ok) fprintf(stderr, "dspfft: execute: HIP error %s\n", hipGetErrorString(hipGetLastError()));
}

void destroy(Shim *s)
{
	if (!s) return;
	dspfft_destroy_plan(s->plan);
	if (s->d_out != s->d_in) (void)hipFree(s->d_out);
	(void)hipFree(s->d_in);
	if (s->stream) (void)hipStreamDestroy(s->stream);
	delete s->up;
	delete s;
}

}  // namespace

extern "C" {

unsigned long long dspfft_fftw_sparse_uploads(void) { return g_sparse_uploads.load(); }

float *fftwf_alloc_real(size_t n) { return (float *)pinned_alloc(n * sizeof(float)); }
void fftwf_free(void *p) { pinned_free(p); }
fftwf_plan fftwf_plan_many_r2r(int rank, const int *n, int howmany, float *in, const int *inembed, int istride, int idist,
                               float *out, const int *onembed, int ostride, int odist, const fftwf_r2r_kind *kind, unsigned flags)
{
	int k[3] = {0, 0, 0};
	if (rank < 1 || rank > 3) { fprintf(stderr, "dspfft: rank %d unsupported\n", rank); return nullptr; }
	for (int a = 0; a < rank; a++) k[a] = (int)kind[a];
	return (fftwf_plan)make_plan(rank, n, howmany, in, inembed, istride, idist, out, onembed, ostride, odist, k, false, flags);
}
fftwf_plan fftwf_plan_r2r_2d(int n0, int n1, float *in, float *out, fftwf_r2r_kind k0, fftwf_r2r_kind k1, unsigned flags)
{
	int n[2] = {n0, n1}, k[2] = {(int)k0, (int)k1};
	return (fftwf_plan)make_plan(2, n, 1, in, nullptr, 1, 0, out, nullptr, 1, 0, k, false, flags);
}
void fftwf_execute(const fftwf_plan p) { run((Shim *)p); }
void fftwf_destroy_plan(fftwf_plan p) { destroy((Shim *)p); }
void fftwf_cleanup(void) {}
int fftwf_init_threads(void) { return 1; }
void fftwf_plan_with_nthreads(int) {}
void fftwf_cleanup_threads(void) {}
int fftwf_import_wisdom_from_filename(const char *) { return 0; }      /* FFTW: 0 = nothing imported */
int fftwf_export_wisdom_to_filename(const char *f) { FILE *fp = f ? fopen(f, "w") : nullptr; if (!fp) return 0; fputs("(dspfft_wisdom)\n", fp); fclose(fp); return 1; }

double *fftw_alloc_real(size_t n) { return (double *)pinned_alloc(n * sizeof(double)); }
void fftw_free(void *p) { pinned_free(p); }
fftw_plan fftw_plan_many_r2r(int rank, const int *n, int howmany, double *in, const int *inembed, int istride, int idist,
                             double *out, const int *onembed, int ostride, int odist, const fftw_r2r_kind *kind, unsigned flags)
{
	int k[3] = {0, 0, 0};
	if (rank < 1 || rank > 3) { fprintf(stderr, "dspfft: rank %d unsupported\n", rank); return nullptr; }
	for (int a = 0; a < rank; a++) k[a] = (int)kind[a];
	return (fftw_plan)make_plan(rank, n, howmany, in, inembed, istride, idist, out, onembed, ostride, odist, k, true, flags);
}
fftw_plan fftw_plan_r2r_2d(int n0, int n1, double *in, double *out, fftw_r2r_kind k0, fftw_r2r_kind k1, unsigned flags)
{
	int n[2] = {n0, n1}, k[2] = {(int)k0, (int)k1};
	return (fftw_plan)make_plan(2, n, 1, in, nullptr, 1, 0, out, nullptr, 1, 0, k, true, flags);
}
void fftw_execute(const fftw_plan p) { run((Shim *)p); }
void fftw_destroy_plan(fftw_plan p) { destroy((Shim *)p); }
void fftw_cleanup(void) {}
int fftw_init_threads(void) { return 1; }
void fftw_plan_with_nthreads(int) {}
void fftw_cleanup_threads(void) {}
int fftw_import_wisdom_from_filename(const char *) { return 0; }
int fftw_export_wisdom_to_filename(const char *f) { return fftwf_export_wisdom_to_filename(f); }

}  // extern "C"
