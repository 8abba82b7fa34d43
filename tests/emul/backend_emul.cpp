// backend_emul.cpp -- TEST-ONLY CPU emulation of the device backend.
//
// Runs the very same phase functions (dspfun_amd/csrc/dct_core.h) the HIP kernels run, one
// "workgroup" at a time, with `for (tid)` loops standing in for the threads and the loop
// boundaries standing in for __syncthreads().  It exists so the planner and the kernel logic can
// be unit-tested in a container without a GPU.  It is NOT part of the product: nothing under
// dspfun_amd/ builds, links, loads or falls back to it (libdspfft_emul.so lives under tests/).
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "backend.h"
#include "dct_spec.h"
#include "spec_list.h"
#include "elementwise_core.h"
#include "scan_core.h"
#include <algorithm>

namespace dspfft {

void *be_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void be_free(void *p) { free(p); }
int be_upload(void *dst, const void *src, size_t bytes) { memcpy(dst, src, bytes); return 0; }
size_t be_max_lds() { const char *e = getenv("DSPFFT_EMUL_LDS"); return e ? (size_t)atol(e) : 160 * 1024; }
const char *be_name() { return "cpu-emulation (tests only)"; }
void *be_event_create() { return malloc(1); }
void be_event_destroy(void *e) { free(e); }
int be_event_record(void *, void *) { return 0; }
int be_event_synchronize(void *) { return 0; }
int be_event_elapsed_ms(void *, void *, float *ms) { *ms = 0.f; return 0; }
void *be_stream_create() { return malloc(1); }
void be_stream_destroy(void *s) { free(s); }
int be_stream_synchronize(void *) { return 0; }
void *be_order_event_create() { return malloc(1); }
int be_stream_wait_event(void *, void *) { return 0; }

#define PHASE(stmt) for (int tid = 0; tid < nthr; tid++) { stmt; }

template <class R>
static int emul_row(const PassArgsT<R> &a, const LaunchGeom &g)
{
	std::vector<unsigned char> lds(g.lds_bytes + 32);
	cx<R> *buf = (cx<R> *)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15);
	long long bases[2 * 16];
	const int nthr = g.nthr, L = a.N / 2, B = a.C * a.LPW;
	for (int wg = 0; wg < g.nwg; wg++) {
		int cnt = 0;
		PHASE(cnt = row_bases(a, wg, bases, tid));
		if (a.kind == KIND_REDFT10) { PHASE(row_load10(a, buf, bases, cnt, tid, nthr)); }
		else { PHASE(row_load01(a, buf, bases, cnt, tid, nthr)); }
		for (int s = 0; s < a.fft.ns; s++) PHASE(fft_stage(buf, L, a.fft.st[s], B, a.divB, a.W, tid, nthr));
		if (a.kind == KIND_REDFT10) { PHASE(row_post10(a, buf, bases, cnt, tid, nthr)); }
		else { PHASE(row_store01(a, buf, bases, cnt, tid, nthr)); }
	}
	return 0;
}

template <class R>
static int emul_col(const PassArgsT<R> &a, const LaunchGeom &g)
{
	std::vector<unsigned char> lds(g.lds_bytes + 32);
	cx<R> *buf = (cx<R> *)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15);
	const int nthr = g.nthr;
	for (int wg = 0; wg < g.nwg; wg++) {
		long long bin, bout; int valid;
		col_base(a, wg, bin, bout, valid);
		if (a.kind == KIND_REDFT10) { PHASE(col_load2(a, buf, bin, valid, tid, nthr)); }
		else { PHASE(col_pre3(a, buf, bin, valid, tid, nthr)); }
		for (int s = 0; s < a.fft.ns; s++) PHASE(fft_stage(buf, a.N, a.fft.st[s], a.B, a.divB, a.W, tid, nthr));
		if (a.kind == KIND_REDFT10) { PHASE(col_post2(a, buf, bout, valid, tid, nthr)); }
		else { PHASE(col_unpack3(a, buf, bout, valid, tid, nthr)); }
	}
	return 0;
}

template <class R>
static int emul_dense(const DenseArgsT<R> &a, const LaunchGeom &g)
{
	if (a.stage) {      // lines too long for LDS: stage every line, then sum from the staging array (two launches on the device)
		const int nthr = g.nthr;
		for (int wg = 0; wg < g.nwg; wg++) { long long bin, bout; dense_base(a, wg, bin, bout); for (int tid = 0; tid < nthr; tid++) dense_load(a, a.stage + (size_t)wg * a.N, bin, tid, nthr); }
		for (int wg = 0; wg < g.nwg; wg++) { long long bin, bout; dense_base(a, wg, bin, bout); for (int tid = 0; tid < nthr; tid++) dense_compute(a, a.stage + (size_t)wg * a.N, bout, tid, nthr); }
		return 0;
	}
	std::vector<R> x(a.N);
	const int nthr = g.nthr;
	for (int wg = 0; wg < g.nwg; wg++) {
		long long bin, bout;
		dense_base(a, wg, bin, bout);
		PHASE(dense_load(a, x.data(), bin, tid, nthr));
		PHASE(dense_compute(a, x.data(), bout, tid, nthr));
	}
	return 0;
}

template <class R>
static int emul_blue(const BlueArgsT<R> &a, const LaunchGeom &g)
{
	std::vector<unsigned char> lds(g.lds_bytes + 32);
	cx<R> *A = (cx<R> *)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15);
	const int nthr = g.nthr;
	for (int wg = 0; wg < g.nwg; wg++) {
		long long bin, bout; int valid;
		col_base(a, wg, bin, bout, valid);
		if (a.kind == KIND_REDFT10) { PHASE(col_load2(a, A, bin, valid, tid, nthr)); }
		else { PHASE(col_pre3(a, A, bin, valid, tid, nthr)); }
		PHASE(blue_chirp_in(a, A, tid, nthr));
		for (int s = 0; s < a.fftM.ns; s++) PHASE(fft_stage(A, a.M, a.fftM.st[s], a.B, a.divB, a.WM, tid, nthr));
		PHASE(blue_mul(a, A, tid, nthr));
		for (int s = a.fftM.ns - 1; s >= 0; s--) PHASE(fft_stage_inv(A, a.M, a.fftM.st[s], a.B, a.divB, a.WM, tid, nthr));
		PHASE(blue_chirp_out(a, A, tid, nthr));
		if (a.kind == KIND_REDFT10) { PHASE(col_post2(a, A, bout, valid, tid, nthr)); }
		else { PHASE(col_unpack3(a, A, bout, valid, tid, nthr)); }
	}
	return 0;
}
int be_launch_blue(const BlueArgs &a, const LaunchGeom &g, void *) { return emul_blue(a, g); }
int be_launch_blue(const BlueArgsD &a, const LaunchGeom &g, void *) { return emul_blue(a, g); }

template <int N, class R>
static int emul_tiny_n(const TinyArgsT<R> &a)
{
	if (a.packed) {
		long long rest = 1;
		for (int d = 1; d < a.nd; d++) rest *= a.bn[d];
		std::vector<R> lds((size_t)TINY_CHUNK * tiny_pitch<N>());
		for (long long wg = 0; wg < rest * a.chunk_div.d; wg++) {
			long long bin, bout; int cnt;
			tiny_row_base(a, (uint32_t)wg, bin, bout, cnt);
			for (int tid = 0; tid < TINY_CHUNK; tid++) tiny_row_load<N>(a, lds.data(), bin, cnt, tid, TINY_CHUNK);
			for (int tid = 0; tid < TINY_CHUNK; tid++) {
				if (a.kind == KIND_REDFT10) tiny_row_compute<N, KIND_REDFT10>(a, lds.data(), cnt, tid); else tiny_row_compute<N, KIND_REDFT01>(a, lds.data(), cnt, tid);
			}
			for (int tid = 0; tid < TINY_CHUNK; tid++) tiny_row_store<N>(a, lds.data(), bout, cnt, tid, TINY_CHUNK);
		}
		return 0;
	}
	for (long long line = 0; line < a.nlines; line++) {
		if (a.kind == KIND_REDFT10) tiny_line<N, KIND_REDFT10>(a, line); else tiny_line<N, KIND_REDFT01>(a, line);
	}
	return 0;
}
template <class R>
static int emul_tiny(const TinyArgsT<R> &a)
{
	switch (a.N) {
#define DSP_TINY_CASE(n) case n: return emul_tiny_n<n>(a);
	DSP_TINY_CASE(1) DSP_TINY_CASE(2) DSP_TINY_CASE(3) DSP_TINY_CASE(4) DSP_TINY_CASE(5) DSP_TINY_CASE(6) DSP_TINY_CASE(7) DSP_TINY_CASE(8)
	DSP_TINY_CASE(9) DSP_TINY_CASE(10) DSP_TINY_CASE(11) DSP_TINY_CASE(12) DSP_TINY_CASE(13) DSP_TINY_CASE(14) DSP_TINY_CASE(15) DSP_TINY_CASE(16)
	DSP_TINY_CASE(17) DSP_TINY_CASE(18) DSP_TINY_CASE(19) DSP_TINY_CASE(20) DSP_TINY_CASE(21) DSP_TINY_CASE(22) DSP_TINY_CASE(23) DSP_TINY_CASE(24)
	DSP_TINY_CASE(25) DSP_TINY_CASE(26) DSP_TINY_CASE(27) DSP_TINY_CASE(28) DSP_TINY_CASE(29) DSP_TINY_CASE(30) DSP_TINY_CASE(31) DSP_TINY_CASE(32)
#undef DSP_TINY_CASE
	default: return -1;
	}
}
int be_launch_tiny(const TinyArgs &a, void *) { return emul_tiny(a); }
int be_launch_tiny(const TinyArgsD &a, void *) { return emul_tiny(a); }

int be_launch_row(const PassArgs &a, const LaunchGeom &g, void *) { return emul_row(a, g); }
int be_launch_col(const PassArgs &a, const LaunchGeom &g, void *) { return emul_col(a, g); }
int be_launch_dense(const DenseArgs &a, const LaunchGeom &g, void *) { return emul_dense(a, g); }
int be_launch_row(const PassArgsD &a, const LaunchGeom &g, void *) { return emul_row(a, g); }
int be_launch_col(const PassArgsD &a, const LaunchGeom &g, void *) { return emul_col(a, g); }
int be_launch_dense(const DenseArgsD &a, const LaunchGeom &g, void *) { return emul_dense(a, g); }

// one "workgroup" per line / tile, as in backend_hip.hip: prefetch (global loads into the per-thread State),
// then the barrier-separated phases
template <class S, int KIND>
int launch_row_spec(const typename S::PA &a, int nwork, void *)
{
	// channel lines (spec_kernels.h row_chan_kernel): one "workgroup" per (line, channel), in the order chan_work hands them out
	typedef typename chan_lines_of<typename S::Re, S::N, S::C>::type CH;
	if constexpr (!std::is_void<CH>::value) {
		if (chan_lines_enabled() || S::LDS > 160 * 1024) {
			std::vector<unsigned char> clds(CH::LDS + 32);
			typename CH::CX *planes = (typename CH::CX *)(((uintptr_t)clds.data() + 31) & ~(uintptr_t)31);
			for (int wg = 0; wg < nwork * CH::GS; wg++) {
				std::vector<typename CH::template State<KIND>> st(CH::T);
				int line, ch;
				chan_work<CH::GS>(wg, nwork, line, ch);
				long long bin, bout;
				row_base(a, line, bin, bout);
				bin += ch; bout += ch;
				const uint8_t *zf = a.zflags ? a.zflags + (line & 1) * a.zhalf : nullptr;
				for (int tid = 0; tid < CH::T; tid++) CH::template prefetch<KIND>(a, bin, tid, st[tid], nullptr, zf, ch);
				static_for<0, CH::NPH>([&](auto ph) { for (int tid = 0; tid < CH::T; tid++) CH::template phase<KIND, ph>(a, planes, bout, tid, st[tid]); });
			}
			return 0;
		}
	}
	std::vector<unsigned char> lds(S::LDS + 32);
	typename S::CX *planes = (typename S::CX *)(((uintptr_t)lds.data() + 31) & ~(uintptr_t)31);
	for (int wg = 0; wg < nwork; wg++) {
		std::vector<typename S::template State<KIND>> st(S::T);
		long long bin, bout;
		row_base(a, wg, bin, bout);
		const uint8_t *zf = a.zflags ? a.zflags + (wg & 1) * a.zhalf : nullptr;
		for (int tid = 0; tid < S::T; tid++) S::template prefetch<KIND>(a, bin, tid, st[tid], nullptr, zf);
		static_for<0, S::NPH>([&](auto ph) { for (int tid = 0; tid < S::T; tid++) S::template phase<KIND, ph>(a, planes, bout, tid, st[tid]); });
	}
	return 0;
}
template <class S, int KIND>
int launch_col_spec(const typename S::PA &a, int nwork, void *)
{
	std::vector<unsigned char> lds(S::LDS + 64);
	typename S::V *buf = (typename S::V *)(((uintptr_t)lds.data() + 31) & ~(uintptr_t)31);
	for (int wg = 0; wg < nwork; wg++) {
		std::vector<typename S::template State<KIND>> st(S::T);
		long long bin, bout;
		int tile;
		bool hit = false;
		S::base(a, wg, bin, bout, tile);
		if (a.zflags && a.mask && a.zranges && (a.mask_id < a.zranges[2 * tile] || a.mask_id > a.zranges[2 * tile + 1])) { a.zflags[tile] = 0; continue; }
		for (int tid = 0; tid < S::T; tid++) S::template prefetch<KIND>(a, bin, tid, st[tid], hit);
		if (a.zflags && a.mask) { a.zflags[tile] = hit; if (!hit) continue; }
		static_for<0, S::NPH>([&](auto ph) { for (int tid = 0; tid < S::T; tid++) S::template phase<KIND, ph>(a, buf, bout, tid, st[tid]); });
	}
	return 0;
}
template <class S>
int launch_row_sum2(const typename S::PA &a, const typename S::PA &b, int nwork, void *)
{
	std::vector<unsigned char> lds(S::LDS + 32);
	typename S::CX *planes = (typename S::CX *)(((uintptr_t)lds.data() + 31) & ~(uintptr_t)31);
	for (int wg = 0; wg < nwork; wg++) {
		std::vector<typename S::template State<KIND_REDFT01>> st(S::T), st2(S::T);
		std::vector<typename S::Hold> hold(S::T);
		long long bin, bout, bin2, bout2;
		row_base(a, wg, bin, bout);
		row_base(b, wg, bin2, bout2);
		for (int tid = 0; tid < S::T; tid++) { S::template prefetch<KIND_REDFT01>(a, bin, tid, st[tid]); S::template prefetch<KIND_REDFT01>(b, bin2, tid, st2[tid]); }
		static_for<0, S::NPH - 1>([&](auto ph) { for (int tid = 0; tid < S::T; tid++) S::template phase<KIND_REDFT01, ph>(a, planes, bout, tid, st[tid]); });
		for (int tid = 0; tid < S::T; tid++) S::template final01_hold<1>(a, planes, bout, tid, hold[tid]);
		static_for<0, S::NPH - 1>([&](auto ph) { for (int tid = 0; tid < S::T; tid++) S::template phase<KIND_REDFT01, ph>(b, planes, bout2, tid, st2[tid]); });
		for (int tid = 0; tid < S::T; tid++) S::template final01_hold<2>(b, planes, bout2, tid, hold[tid]);
	}
	return 0;
}
// zoom's x stage on the duo row kernel (spec_kernels.h zoomx_lean_kernel), the same two paths: the line's pixels loaded whole and the channels
// unrolled (Z::WHOLE), or channel by channel
template <class S, int C, int NSRC, bool CLIP>
int emul_zoomx(const ZoomXArgs &a)
{
	typedef ZoomXLeanT<S, C, NSRC, CLIP> Z;
	std::vector<unsigned char> lds(S::LDS + 64);
	typename S::V *buf = (typename S::V *)(((uintptr_t)lds.data() + 31) & ~(uintptr_t)31);
	PassArgs w = {};
	w.W = a.W;
	for (int line = 0; line < a.lines; line++) {
		const long long bin = (long long)line * a.in_pitch, bout = (long long)line * a.out_pitch;
		std::vector<typename Z::State> st(S::T);
		std::vector<typename Z::Ex> ex(S::T);
		if constexpr (Z::WHOLE) {
			std::vector<typename Z::Pixels> px(S::T);
			for (int tid = 0; tid < S::T; tid++) Z::load_pixels(a, bin, tid, px[tid]);
			static_for<0, C>([&](auto c) {
				for (int tid = 0; tid < S::T; tid++) Z::template phase_a_held<c>(a, w, buf, tid, px[tid]);
				static_for<1, S::NS - 1>([&](auto I) { for (int tid = 0; tid < S::T; tid++) Z::template phase_b<I>(w, buf, tid); });
				for (int tid = 0; tid < S::T; tid++) Z::phase_c(buf, tid, ex[tid]);
				for (int tid = 0; tid < S::T; tid++) Z::template phase_c_emit_ch<c>(a, bout, tid, ex[tid], st[tid]);
			});
			continue;
		}
		for (int c = 0; c < C; c++) {
			for (int tid = 0; tid < S::T; tid++) Z::phase_a(a, w, buf, bin, c, tid);
			static_for<1, S::NS - 1>([&](auto I) { for (int tid = 0; tid < S::T; tid++) Z::template phase_b<I>(w, buf, tid); });
			for (int tid = 0; tid < S::T; tid++) Z::phase_c(buf, tid, ex[tid]);
			for (int tid = 0; tid < S::T; tid++) Z::phase_c_emit(a, bout, c, tid, ex[tid], st[tid]);
		}
	}
	return 0;
}
template <class S, int C>
int launch_zoomx(const ZoomXArgs &a, int nsrc, bool clip, void *)
{
	switch (nsrc * 2 + (clip ? 1 : 0)) {
	case 2: return emul_zoomx<S, C, 1, false>(a);
	case 3: return emul_zoomx<S, C, 1, true>(a);
	case 4: return emul_zoomx<S, C, 2, false>(a);
	case 5: return emul_zoomx<S, C, 2, true>(a);
	case 8: return emul_zoomx<S, C, 4, false>(a);
	case 9: return emul_zoomx<S, C, 4, true>(a);
	}
	return -1;
}
int be_zoomx_tables(float *tab, int M, int cw, int nsrc, double theta, double scale, void *)
{
	for (int i = 0; i < nsrc * (M / 4 + 1); i++) zoomx_table_item(tab, M, cw, nsrc, theta, scale, i);
	return 0;
}
// chirp-z rows (spec_kernels.h czt_rows_kernel / czt_spectrum_kernel)
template <class S>
int launch_czt_rows(const CztArgs &a, void *)
{
	std::vector<unsigned char> lds(S::LDS + 64);
	cf *plane = (cf *)(((uintptr_t)lds.data() + 31) & ~(uintptr_t)31);
	PassArgs w = {};
	w.W = a.W;
	for (int wg = 0; wg < a.lines; wg++) {
		long long bin, bout;
		S::base(a, wg, bin, bout);
		static_for<0, S::NPH>([&](auto ph) { for (int tid = 0; tid < S::T; tid++) S::template phase<ph>(a, w, plane, bin, bout, tid); });
	}
	return 0;
}
template <class S>
int launch_czt_spectrum(const CztArgs &a, cf *hspec, void *)
{
	std::vector<unsigned char> lds(S::LDS + 64);
	cf *plane = (cf *)(((uintptr_t)lds.data() + 31) & ~(uintptr_t)31);
	PassArgs w = {};
	w.W = a.W;
	for (int tid = 0; tid < S::T; tid++) S::f0(a, plane, 0, tid);
	static_for<1, S::NS - 1>([&](auto I) { for (int tid = 0; tid < S::T; tid++) S::template fwd<I>(w, plane, tid); });
	for (int tid = 0; tid < S::T; tid++) S::flast(plane, tid);
	for (int i = 0; i < S::P; i++) hspec[i] = plane[S::F::padded(i)];
	return 0;
}
int be_czt_tables(cf *atab, cf *etab, cf *htab, int nc, int nout, int P, double omega, double phi, double scale, void *)
{
	if (atab) for (int i = 0; i < nc; i++) atab[i] = czt_a_entry(i, omega, phi);
	if (etab) for (int i = 0; i < nout; i++) etab[i] = czt_e_entry(i, omega, scale);
	if (htab) for (int i = 0; i < P; i++) htab[i] = czt_h_entry(i, P, nc, nout, omega);
	return 0;
}
int be_transpose(float *out, long long out_pitch, const float *in, long long in_pitch, int rows, int cols, void *)
{
	for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) out[(long long)c * out_pitch + r] = in[(long long)r * in_pitch + c];
	return 0;
}
template <class S, int KIND>
int launch_row_spec_u8(const PassArgs &a, const U8IO &io, int nwork, void *)
{
	std::vector<unsigned char> lds(S::LDS + 16);
	cf *planes = (cf *)lds.data();
	for (int wg = 0; wg < nwork; wg++) {
		std::vector<typename S::template State<KIND>> st(S::T);
		long long bin, bout;
		row_base(a, wg, bin, bout);
		for (int tid = 0; tid < S::T; tid++) S::template prefetch<KIND>(a, bin, tid, st[tid], &io);
		static_for<0, S::NPH>([&](auto ph) { for (int tid = 0; tid < S::T; tid++) S::template phase<KIND, ph>(a, planes, bout, tid, st[tid], &io); });
	}
	return 0;
}
static inline bool is_plain_pass(const PassArgs &a) { return !a.mask && !a.zflags && !a.accumulate && a.win_hi <= 0 && !a.alt_out && !a.in_mul && !a.in_rev; }
// the pipelined pair kernel's order of work (spec_kernels.h row_pair_pipe_kernel): the pair's butterfly on the OUTPUT side -- T(r1) kept as values
// (RowSpecG::final_each), T(r2) added / subtracted and both lines stored -- on the workgroup size the HIP launcher uses.  EMUL_PAIR_PIPE=1 sends every
// plain interleaved pair pass through it (the HIP launcher takes it for lines that fill a CU's LDS only), =0 none.
template <class S, int KIND>
int launch_row_pair_pipe(const PassArgs &a_, int npairs)
{
	std::vector<unsigned char> lds(S::LDS + 16);
	cf *planes = (cf *)lds.data();
	const PassArgs &a = a_;                          // (plain, or the scan step's pair pass: launch_row_pair checks)
	const int pairs = a.nb0 >> 1;
	typedef typename S::template State<KIND> ST;
	typedef typename S::template OutHold<KIND> OH;
	for (int wg = 0; wg < npairs; wg++) {
		const int i1 = wg / pairs, n = wg - i1 * pairs;
		const int y[2] = {2 * n, a.nb0 - 1 - 2 * n};
		const long long b1 = y[0] * a.sb0_out + i1 * a.sb1_out, b2 = y[1] * a.sb0_out + i1 * a.sb1_out;
		std::vector<ST> st(S::T), st2(S::T);
		std::vector<OH> h1(S::T);
		// (both lines are read before anything is stored, as in the kernel: in place is safe)
		// the scan form (REDFT01 behind the masked half-tile pass): the loads go by this thread's precomputed tile flags -- even rows' flags for r1, odd rows' for r2
		const bool scan = KIND == KIND_REDFT01 && !is_plain_pass(a);
		for (int tid = 0; tid < S::T; tid++) {
			if constexpr (KIND == KIND_REDFT01) {
				if (scan) {
					S::prefetch01_bits(a, y[0] * a.sb0_in + i1 * a.sb1_in, tid, st[tid], S::flag_bits01(a.zflags, a.zshift, tid));
					S::prefetch01_bits(a, y[1] * a.sb0_in + i1 * a.sb1_in, tid, st2[tid], S::flag_bits01(a.zflags ? a.zflags + a.zhalf : nullptr, a.zshift, tid));
					continue;
				}
			}
			S::template prefetch<KIND>(a, y[0] * a.sb0_in + i1 * a.sb1_in, tid, st[tid], nullptr, nullptr);
			S::template prefetch<KIND>(a, y[1] * a.sb0_in + i1 * a.sb1_in, tid, st2[tid], nullptr, nullptr);
		}
		for (int second = 0; second < 2; second++) {
			std::vector<ST> &cur = second ? st2 : st;
			static_for<0, S::NS + 2>([&](auto ph) { for (int tid = 0; tid < S::T; tid++) S::template phase<KIND, ph, ST, true>(a, planes, 0, tid, cur[tid]); });
			// REDFT10 as the kernel runs it: every slot of every thread (final_each UNCOND: clamped items, repeated pixels), this thread's twiddles through tw_index
			constexpr bool UNC = KIND == KIND_REDFT10;
			for (int tid = 0; tid < S::T; tid++) {
				if constexpr (UNC) for (int ri = 0; ri < S::K_ROUNDS; ri++) cur[tid].tw[ri] = a.T[S::template tw_index<true>(tid + ri * S::T)];
				if (!second) S::template final_each<KIND, UNC>(a, planes, tid, cur[tid], [&](auto slot, long long, Pix<S::C, float> v) { h1[tid].v[slot] = v; });
				else S::template final_each<KIND, UNC>(a, planes, tid, cur[tid], [&](auto slot, long long off, Pix<S::C, float> v) {
					Pix<S::C, float> o1, o2;
					for (int c = 0; c < S::C; c++) { o1.v[c] = h1[tid].v[slot].v[c] + v.v[c]; o2.v[c] = h1[tid].v[slot].v[c] - v.v[c]; }
					if (a.accumulate) {               // sum += image (scan.c:451-459)
						const Pix<S::C, float> q1 = load_pix<S::C, float>(a.out + b1 + off), q2 = load_pix<S::C, float>(a.out + b2 + off);
						for (int c = 0; c < S::C; c++) { o1.v[c] += q1.v[c]; o2.v[c] += q2.v[c]; }
					}
					store_pix<S::C, float>(a.out + b1 + off, o1);
					store_pix<S::C, float>(a.out + b2 + off, o2);
				});
			}
		}
	}
	return 0;
}
template <class S, int KIND>
int launch_row_pair(const PassArgs &a, int npairs, void *)
{
	{
		const char *e = getenv("EMUL_PAIR_PIPE");
		const int mode = e ? atoi(e) : -1;
		constexpr bool fills_a_cu = S::C > 1 && S::LDS > 80 * 1024;
		// plain passes, and REDFT01 pair passes of the fused scan step (tile flags + accumulation, out of place), as the HIP launcher chooses
		const bool scan_pair = KIND == KIND_REDFT01 && !a.mask && a.win_hi <= 0 && !a.alt_out && !a.in_mul && !a.in_rev && a.in != a.out;
		if ((is_plain_pass(a) || scan_pair) && S::C > 1 && (mode == 1 || (mode != 0 && fills_a_cu))) {
			if constexpr (fills_a_cu) return launch_row_pair_pipe<typename S::template with_threads<768>, KIND>(a, npairs);
			else if constexpr (S::C > 1) return launch_row_pair_pipe<S, KIND>(a, npairs);
		}
	}
	std::vector<unsigned char> lds(S::LDS + 16);
	cf *planes = (cf *)lds.data();
	const int pairs = a.nb0 >> 1;
	for (int wg = 0; wg < npairs; wg++) {
		std::vector<typename S::template State<KIND>> st(S::T), st2(S::T);
		const int i1 = wg / pairs, n = wg - i1 * pairs;
		const int y1 = 2 * n, y2 = a.nb0 - 1 - 2 * n;
		const long long bin1 = y1 * a.sb0_in + i1 * a.sb1_in, bin2 = y2 * a.sb0_in + i1 * a.sb1_in;
		const long long bout1 = y1 * a.sb0_out + i1 * a.sb1_out, bout2 = y2 * a.sb0_out + i1 * a.sb1_out;
		constexpr int NPRE = (int)(sizeof(st[0].pre) / sizeof(float));
		for (int tid = 0; tid < S::T; tid++) {
			for (int i = 0; i < NPRE; i++) st[tid].pre[i] = st2[tid].pre[i] = 0.f;
			S::template prefetch<KIND>(a, bin1, tid, st[tid], nullptr, a.zflags);
			S::template prefetch<KIND>(a, bin2, tid, st2[tid], nullptr, a.zflags ? a.zflags + a.zhalf : nullptr);
			for (int i = 0; i < NPRE; i++) { const float p = st[tid].pre[i], q = st2[tid].pre[i]; st[tid].pre[i] = p + q; st2[tid].pre[i] = p - q; }
		}
		typedef typename S::template State<KIND> ST;
		static_for<0, S::NPH>([&](auto ph) { for (int tid = 0; tid < S::T; tid++) S::template phase<KIND, ph, ST, true>(a, planes, bout1, tid, st[tid]); });
		for (int tid = 0; tid < S::T; tid++) for (int i = 0; i < NPRE; i++) st[tid].pre[i] = st2[tid].pre[i];
		static_for<0, S::NPH>([&](auto ph) { for (int tid = 0; tid < S::T; tid++) S::template phase<KIND, ph, ST, true>(a, planes, bout2, tid, st[tid]); });
	}
	return 0;
}
template <class S, int KIND>
int launch_col_half(const PassArgs &a, int nwork, void *)
{
	std::vector<unsigned char> lds(S::LDS + 32);
	typename S::V *buf = (typename S::V *)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15);
	for (int wg = 0; wg < nwork; wg++) {
		std::vector<typename S::template State<KIND>> st(S::T);
		long long bin, bout; int h, tile;
		bool hit = false;
		S::base(a, wg, bin, bout, h, tile);
		if (a.zflags && a.mask && a.zranges && (a.mask_id < a.zranges[2 * tile] || a.mask_id > a.zranges[2 * tile + 1])) { a.zflags[tile] = 0; continue; }
		for (int tid = 0; tid < S::T; tid++) S::template prefetch<KIND>(a, bin, h, tid, st[tid], hit);
		if (a.zflags && a.mask) { a.zflags[tile] = hit; if (!hit) continue; }
		static_for<0, S::NPH>([&](auto ph) { for (int tid = 0; tid < S::T; tid++) S::template phase<KIND, ph>(a, buf, bout, h, tid, st[tid]); });
	}
	return 0;
}
struct FilterOp {
	MotionFilter p;
	SigVec<float, 2> operator()(long long e, SigVec<float, 2> v, unsigned long long &coded) const
	{
		if (!p.enabled) return v;
		float4 f; f.x = v.s[0].x; f.y = v.s[0].y; f.z = v.s[1].x; f.w = v.s[1].y;
		f = motion_filter4(p, (uint32_t)e, f, coded);
		v.s[0].x = f.x; v.s[0].y = f.y; v.s[1].x = f.z; v.s[1].y = f.w;
		return v;
	}
};
template <class S>
int launch_col_roundtrip(const PassArgs &af, const PassArgs &ai, const MotionFilter &filt, unsigned long long *coded, int nwork, void *)
{
	std::vector<unsigned char> lds(S::LDS + 32);
	typename S::V *buf = (typename S::V *)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15);
	FilterOp f; f.p = filt;
	unsigned long long mine = 0;
	for (int wg = 0; wg < nwork; wg++) {
		std::vector<typename S::StateRT> st(S::T);
		long long bin, bout;
		S::base(af, wg, bin, bout);
		for (int tid = 0; tid < S::T; tid++) S::template prefetch<KIND_REDFT10>(af, bin, tid, st[tid]);
		static_for<0, S::NS + 2>([&](auto ph) { for (int tid = 0; tid < S::T; tid++) S::template phase<KIND_REDFT10, ph>(af, buf, bout, tid, st[tid]); });
		for (int tid = 0; tid < S::T; tid++) S::mid_read(af, ai, buf, bout, tid, st[tid], f, mine);
		for (int tid = 0; tid < S::T; tid++) S::mid_write(buf, tid, st[tid]);
		static_for<1, S::NPH>([&](auto ph) { for (int tid = 0; tid < S::T; tid++) S::template phase<KIND_REDFT01, ph>(ai, buf, bout, tid, st[tid]); });
	}
	if (coded) *coded += mine;
	return 0;
}
#include "spec_registry.inc"

// TEST-ONLY entry: the 8-bit quantiser of the fused stores (elementwise_core.h quantise_u8_of: single precision away from rounding boundaries,
// the double path next to them) over n values, beside the double path itself
extern "C" void emul_quantise_u8_of(const float *v, double mul, uint8_t *fast, uint8_t *exact, long long n)
{
	const float mulf = (float)mul;
	for (long long i = 0; i < n; i++) { fast[i] = (uint8_t)quantise_u8_of(v[i], mul, mulf); exact[i] = quantise_u8((double)v[i] * mul); }
}

// TEST-ONLY entry: motion's filter (motion_filter.h motion_filter_at, the function the HIP kernels call per element) over one block, with the
// arguments of dspfft_motion_filter (include/dspfft.h); returns the count of coded coefficients
extern "C" unsigned long long emul_motion_filter(float *c, const int active[3], const int minbuf_hw[2], const int bb[3], const int be[3], float damp, float boost,
                                                 float thr_lo, float thr_hi, int preserve_dc, float grey_add, float quantizer)
{
	MotionFilter p;
	p.ad = active[0]; p.ah = active[1]; p.aw = active[2]; p.mh = minbuf_hw[0]; p.mw = minbuf_hw[1];
	p.b0d = bb[0]; p.b0h = bb[1]; p.b0w = bb[2]; p.b1d = be[0]; p.b1h = be[1]; p.b1w = be[2];
	p.damp = damp; p.boost = boost; p.thr_lo = thr_lo; p.thr_hi = thr_hi; p.preserve_dc = preserve_dc; p.grey_add = grey_add; p.quantizer = quantizer;
	p.enabled = 1; motion_filter_set_divs(p, p.ad > 0 ? p.ad : 1);
	unsigned long long coded = 0;
	for (int z = 0; z < p.ad; z++) for (int y = 0; y < p.ah; y++) for (int x = 0; x < p.aw; x++) {
		float &v = c[((size_t)z * p.mh + y) * p.mw + x];
		v = motion_filter_at(p, z, y, x, v, coded);
	}
	return coded;
}

// TEST-ONLY entries: the per-sample functions of motion_ops.hip's load / store kernels (motion_filter.h motion_load_pel / motion_store_pel) over a
// {n[0], n[1], n[2]} corner of planes of minbuf_hw[0] x minbuf_hw[1], with the arguments of dspfft_motion_load_u8 / _f32 and dspfft_motion_store_u8 / _f32
extern "C" void emul_motion_load(float *c, const void *pix, int float_pixels, const int n[3], const int minbuf_hw[2], int mode, double ic, double norm)
{
	for (int z = 0; z < n[0]; z++) for (int y = 0; y < n[1]; y++) for (int x = 0; x < n[2]; x++) {
		const size_t o = ((size_t)z * minbuf_hw[0] + y) * minbuf_hw[1] + x;
		const double pel = float_pixels ? (double)(((const float *)pix)[o] * 255.0f) : (double)((const uint8_t *)pix)[o];
		c[o] = (float)motion_load_pel(pel, mode, ic, norm);
	}
}
extern "C" void emul_motion_store(void *pix, int float_pixels, const float *c, const int n[3], const int minbuf_hw[2], int mode, double scalefactor, double norm, double cc)
{
	for (int z = 0; z < n[0]; z++) for (int y = 0; y < n[1]; y++) for (int x = 0; x < n[2]; x++) {
		const size_t o = ((size_t)z * minbuf_hw[0] + y) * minbuf_hw[1] + x;
		const double pel = motion_store_pel((double)c[o], mode, scalefactor, norm, cc);
		if (float_pixels) ((float *)pix)[o] = (float)(pel / 255);
		else ((uint8_t *)pix)[o] = pel > 255 ? 255 : pel < 0 ? 0 : (uint8_t)lround(pel);
	}
}

int be_motion_filter(float *buf, const MotionFilter &filt, uint64_t span, unsigned long long *coded, void *)
{
	unsigned long long mine = 0;
	for (uint64_t i = 0; i < span; i++) buf[i] = motion_filter_elem(filt, (uint32_t)i, buf[i], mine);
	if (coded) *coded += mine;
	return 0;
}

int be_scan_zigzag(uint32_t *lin, uint32_t w, uint32_t h, uint64_t first, uint64_t count, void *)
{
	for (uint64_t i = 0; i < count; i++) lin[i] = zigzag_lin(w, h, first + i);
	return 0;
}
int be_scan_zigzag_frame_ids(uint32_t *ids, uint32_t w, uint32_t h, uint64_t step, void *)
{
	for (uint64_t i = 0; i < (uint64_t)w * h; i++) { const uint32_t p = zigzag_lin(w, h, i); ids[p] = !step ? (uint32_t)i : p ? (uint32_t)(i / step) : 0xffffffffu; }
	return 0;
}
int be_scan_owner_index(uint32_t *idx, int method, uint32_t w, uint32_t h, uint64_t step, void *)
{
	for (uint64_t p = 0; p < (uint64_t)w * h; p++) {
		const uint64_t i = scan_owner_index(method, w, h, p / w, p % w);
		idx[p] = step ? (p ? (uint32_t)(i / step) : SCAN_NONE) : (uint32_t)i;
	}
	return 0;
}
int be_scan_coords(uint32_t *lin, int method, uint32_t w, uint32_t h, uint64_t first, uint64_t count, uint64_t slots, void *)
{
	for (uint64_t k = 0; k < count; k++)
		for (uint64_t j = 0; j < slots; j++)
			lin[k * slots + j] = j < scan_interval(method, w, h, first + k) ? scan_coord_lin(method, w, h, first + k, j) : SCAN_NONE;
	return 0;
}
int be_scan_stamp(uint32_t *ids, const uint32_t *lin, uint64_t n, uint32_t frame, void *)
{
	for (uint64_t t = 0; t < n; t++) if (lin[t] != SCAN_NONE && lin[t] != 0) ids[lin[t]] = frame;
	return 0;
}
// the emulation instantiates the block kernels for the shapes its tests use only (the product's list of 36 shapes costs g++ minutes);
// plans for other shapes fall back to one TINY pass per axis here
#undef DSPFFT_BLOCK_SHAPES
#define DSPFFT_BLOCK_SHAPES(X) X(8, 8, 8) X(4, 4, 4) X(16, 16, 16) X(8, 8, 1) X(16, 16, 1) X(16, 8, 8) X(8, 8, 4) X(4, 16, 8) X(8, 16, 4)
bool be_jit_available() { return false; }
int be_jit_build(const char *, int, int, const char *, void **, char *, size_t) { return -1; }
int be_jit_launch(void *, const void *, int, int, void *) { return -1; }
int be_jit_launch_n(void *, void **, int, int, void *) { return -1; }

bool be_block_supported(int nx, int ny, int nz)
{
#define DSP_BLOCK_HAS(X_, Y_, Z_) if (nx == X_ && ny == Y_ && nz == Z_) return true;
	DSPFFT_BLOCK_SHAPES(DSP_BLOCK_HAS)
#undef DSP_BLOCK_HAS
	return false;
}
template <int NX, int NY, int NZ, int KIND>
static int emul_block(const BlockArgs &a, int nwg, size_t lds_bytes)
{
	std::vector<unsigned char> raw(lds_bytes + 32);
	float *lds = (float *)(((uintptr_t)raw.data() + 15) & ~(uintptr_t)15);
	for (int wg = 0; wg < nwg; wg++) {
		long long bin, bout; int cnt;
		block_base(a, (uint32_t)wg, bin, bout, cnt);
		for (int tid = 0; tid < BLOCK_THREADS; tid++) block_load_x<NX, NY, NZ, KIND, false>(a, block_axis_args(a.s, 0, NY == 1 && NZ == 1), a.in, nullptr, lds, bin, cnt, tid);
		for (int tid = 0; tid < BLOCK_THREADS; tid++) block_lines_y<NX, NY, NZ, KIND>(a, block_axis_args(a.s, 1, NZ == 1), lds, cnt, tid);
		for (int tid = 0; tid < BLOCK_THREADS; tid++) block_lines_z<NX, NY, NZ, KIND>(a, block_axis_args(a.s, 2, true), lds, cnt, tid);
		for (int tid = 0; tid < BLOCK_THREADS; tid++) block_store_rows<NX, NY, NZ>(a, a.out, lds, bout, cnt, tid);
	}
	return 0;
}
int be_launch_block(const BlockArgs &a, int nwg, size_t lds, void *)
{
#define DSP_BLOCK_CASE(X_, Y_, Z_) \
	if (a.nx == X_ && a.ny == Y_ && a.nz == Z_) \
		return a.kind == KIND_REDFT10 ? emul_block<X_, Y_, Z_, KIND_REDFT10>(a, nwg, lds) : emul_block<X_, Y_, Z_, KIND_REDFT01>(a, nwg, lds);
	DSPFFT_BLOCK_SHAPES(DSP_BLOCK_CASE)
#undef DSP_BLOCK_CASE
	return -1;
}
template <int NX, int NY, int NZ, bool IN8, bool OUT8>
static int emul_block_rt(const BlockRtArgs &a, int nwg, size_t lds_bytes)
{
	std::vector<unsigned char> raw(lds_bytes + 32);
	float *lds = (float *)(((uintptr_t)raw.data() + 15) & ~(uintptr_t)15);
	unsigned long long mine = 0;
	for (int wg = 0; wg < nwg; wg++) {
		long long bin, bout; int cnt;
		block_base(a, (uint32_t)wg, bin, bout, cnt);
		for (int tid = 0; tid < BLOCK_THREADS; tid++) block_load_x<NX, NY, NZ, KIND_REDFT10, IN8>(a, block_axis_args(a.f, 0, NY == 1 && NZ == 1), a.in, a.in8, lds, bin, cnt, tid);
		if constexpr (NZ > 1) {
			for (int tid = 0; tid < BLOCK_THREADS; tid++) block_lines_y<NX, NY, NZ, KIND_REDFT10>(a, block_axis_args(a.f, 1, false), lds, cnt, tid);
			for (int tid = 0; tid < BLOCK_THREADS; tid++) block_lines_mid<NX, NY, NZ, true>(a, block_axis_args(a.f, 2, true), block_axis_args(a.i, 2, false), a.filt, lds, cnt, tid, mine);
			for (int tid = 0; tid < BLOCK_THREADS; tid++) block_lines_y<NX, NY, NZ, KIND_REDFT01>(a, block_axis_args(a.i, 1, false), lds, cnt, tid);
		} else {
			for (int tid = 0; tid < BLOCK_THREADS; tid++) block_lines_mid<NX, NY, NZ, false>(a, block_axis_args(a.f, 1, true), block_axis_args(a.i, 1, false), a.filt, lds, cnt, tid, mine);
		}
		for (int tid = 0; tid < BLOCK_THREADS; tid++) block_store_x<NX, NY, NZ, KIND_REDFT01, OUT8>(a, block_axis_args(a.i, 0, true), a.out, a.out8, a.mul8, lds, bout, cnt, tid);
	}
	if (a.coded) *a.coded += mine;
	return 0;
}
int be_launch_block_roundtrip(const BlockRtArgs &a, int nwg, size_t lds, void *)
{
#define DSP_BLOCK_CASE(X_, Y_, Z_) \
	if (a.nx == X_ && a.ny == Y_ && a.nz == Z_) { \
		if (a.in8) return a.out8 ? emul_block_rt<X_, Y_, Z_, true, true>(a, nwg, lds) : emul_block_rt<X_, Y_, Z_, true, false>(a, nwg, lds); \
		return a.out8 ? emul_block_rt<X_, Y_, Z_, false, true>(a, nwg, lds) : emul_block_rt<X_, Y_, Z_, false, false>(a, nwg, lds); \
	}
	DSPFFT_BLOCK_SHAPES(DSP_BLOCK_CASE)
#undef DSP_BLOCK_CASE
	return -1;
}
int be_scan_tile_ranges(uint32_t *ranges, const uint32_t *ids, TileRangeGeom g0, int halves, void *)
{
	for (int b = 0; b < g0.ntiles * halves; b++) {
		TileRangeGeom g = g0;
		const int half = b / g.ntiles, tile = b - half * g.ntiles;
		if (halves == 2) { g.row_start = half; g.row_step = 2; }
		uint32_t lo = 0xFFFFFFFFu, hi = 0;
		for (long long it = 0; it < (long long)g.nrows * g.K; it++) tile_range_item(g, ids, tile, it, lo, hi);
		ranges[2 * b] = lo; ranges[2 * b + 1] = hi;
	}
	return 0;
}
int be_scan_tile_eids(void *eids, const uint32_t *ids, TileEidGeom g, void *)
{
	for (long long it = 0; it < (long long)g.ntiles * g.N * g.K; it++) tile_eid_item(g, ids, eids, it);
	return 0;
}
int be_download(void *dst, const void *src, size_t bytes, void *) { memcpy(dst, src, bytes); return 0; }
int be_scan_index_to_frame_ids(uint32_t *ids, uint64_t n, uint64_t step, void *)
{
	for (uint64_t p = 0; p < n; p++) ids[p] = p ? (uint32_t)(ids[p] / step) : SCAN_NONE;
	return 0;
}
size_t be_scan_magnitude_work_bytes(uint32_t, uint32_t) { return 16; }
int be_scan_magnitude_index(uint32_t *idx, const float *coeffs, uint32_t w, uint32_t h, int ch, double q, void *, size_t, uint32_t *limit, void *)
{
	const uint64_t n = (uint64_t)w * h;
	std::vector<float> key(n);
	std::vector<uint32_t> order(n);
	for (uint64_t p = 0; p < n; p++) { key[p] = scan_magnitude_key(coeffs + p * ch, ch, p / w, p % w, q); order[p] = (uint32_t)p; }
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] > key[b]; });
	uint32_t j = 0;
	for (uint64_t k = 0; k < n; k++) {
		if (k > 0 && (k == 1 || key[order[k - 1]] != key[order[k - 2]])) j++;
		idx[order[k]] = j;
	}
	if (limit) *limit = j + 1;
	return 0;
}
int be_scan_scatter(float *recon, const float *coeffs, const uint32_t *lin, uint64_t count, uint64_t npixels, int ch, void *)
{
	memset(recon, 0, sizeof(float) * npixels * ch);
	for (uint64_t i = 0; i < count; i++) if (lin[i]) for (int z = 0; z < ch; z++) recon[(uint64_t)lin[i] * ch + z] = coeffs[(uint64_t)lin[i] * ch + z];
	return 0;
}
int be_accumulate(float *sum, const float *img, uint64_t len, void *) { for (uint64_t i = 0; i < len; i++) sum[i] += img[i]; return 0; }
int be_broadcast_dc(float *sum, const float *c, uint64_t npix, int ch, void *) { for (uint64_t p = 0; p < npix; p++) for (int z = 0; z < ch; z++) sum[p * ch + z] = c[z]; return 0; }
int be_zero(void *p, size_t bytes, void *) { memset(p, 0, bytes); return 0; }
int be_region_u8_to_f32(float *dst, const uint8_t *src, const int n[3], const long long sd[3], const long long ss[3], void *)
{
	for (long long z = 0; z < n[0]; z++) for (long long y = 0; y < n[1]; y++) for (long long x = 0; x < n[2]; x++)
		dst[z * sd[0] + y * sd[1] + x * sd[2]] = (float)src[z * ss[0] + y * ss[1] + x * ss[2]];
	return 0;
}
int be_region_f32_to_u8(uint8_t *dst, const float *src, double mul, const int n[3], const long long sd[3], const long long ss[3], void *)
{
	for (long long z = 0; z < n[0]; z++) for (long long y = 0; y < n[1]; y++) for (long long x = 0; x < n[2]; x++)
		dst[z * sd[0] + y * sd[1] + x * sd[2]] = quantise_u8((double)src[z * ss[0] + y * ss[1] + x * ss[2]] * mul);
	return 0;
}
int be_u8_to_f32(float *d, const uint8_t *s, uint64_t len, void *) { for (uint64_t i = 0; i < len; i++) d[i] = (float)s[i]; return 0; }
int be_f32_to_u8(uint8_t *d, const float *s, double mul, uint64_t len, void *) { for (uint64_t i = 0; i < len; i++) d[i] = quantise_u8((double)s[i] * mul); return 0; }

}  // namespace dspfft
