// spec_kernels.h -- the compile-time-specialised kernels (dct_spec.h) and their host launchers, as templates.
// backend_hip.hip only REFERENCES the instantiations (extern template); spec_inst_*.hip define them, one group of
// specs per translation unit, so the groups compile in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "backend.h"
#include "dct_spec.h"
#include "spec_list.h"
#include "spec_fused.h"
#include "dct_duo.h"
#include "dct_czt.h"

namespace dspfft {

// tools/kstamp.hip defines DSP_STAMP to record the clock behind the barriers of one kernel at a time (per-phase durations); nothing in the product
#ifndef DSP_STAMP
#define DSP_STAMP(i) ((void)0)
#endif

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

template <class K>
static int allow_lds(K kernel, size_t bytes)
{
	if (bytes > 48 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
	return 0;
}
// The attribute is per device: a process that drives several GPUs must set it on each (one flag per device ordinal and launcher;
// a race between two host threads sets it twice, which is harmless).
struct DevOnce { enum { MAXDEV = 32 }; unsigned char done[MAXDEV] = {}; int rc[MAXDEV] = {}; int cus[MAXDEV] = {}; };
static inline int current_device() { int d = 0; return hipGetDevice(&d) == hipSuccess && d >= 0 ? d : 0; }
template <class... K>
static int allow_lds_dev(DevOnce &o, size_t bytes, K... kernels)
{
	const int d = current_device();
	if (d >= DevOnce::MAXDEV) { int rc = 0; ((rc = rc ? rc : allow_lds(kernels, bytes)), ...); return rc; }
	if (!o.done[d]) { int rc = 0; ((rc = rc ? rc : allow_lds(kernels, bytes)), ...); o.rc[d] = rc; o.done[d] = 1; }
	return o.rc[d];
}
static inline int device_cus()
{
	static DevOnce o;
	const int d = current_device();
	if (d >= DevOnce::MAXDEV) return 256;
	if (!o.cus[d]) { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n < 1) n = 256; o.cus[d] = n; }
	return o.cus[d];
}

// ---- compile-time-specialised kernels (dct_spec.h) ----
// One workgroup per line / tile.  (A persistent variant that prefetched the next item into
// registers was measured slower on MI355X: the extra ~60 VGPRs cost a resident workgroup per CU,
// and co-resident workgroups already overlap each other's memory and LDS phases.)
// PLAIN: the instantiation a plain dspfft_execute runs -- no owner-id mask, no tile flags, no accumulation.  The fused scan step's
// fields are pinned to "off" on a local copy of the arguments, so every branch on them folds away after inlining (the fused code is a
// third of the kernel's text; two such kernels sharing the CUs ran 1.5 % faster without it: tools/sbench.hip vs -DUSELIB)
// (plain_args / is_plain: spec_fused.h, shared with the plan-time kernels)

template <class S, int KIND, bool PLAIN>
__global__ void __launch_bounds__(S::T, S::WPE) row_spec_kernel(const typename S::PA a_)
{
	const typename S::PA a = PLAIN ? plain_args(a_) : a_;
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	typename S::CX *planes = reinterpret_cast<typename S::CX *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	row_base(a, blockIdx.x, bin, bout);
	// behind a masked column pass that skipped its empty tiles: this line's tile flags (PassGeom::zflags; one image: line = row)
	const uint8_t *zf = a.zflags ? a.zflags + (blockIdx.x & 1) * a.zhalf : nullptr;
	S::template prefetch<KIND>(a, bin, tid, st, nullptr, zf);
	S::fetch_stage_twiddles(a, tid, st);
	S::template phase<KIND, 0>(a, planes, bout, tid, st);
	__syncthreads();
	static_for<1, S::NPH>([&](auto ph) {
		{
			S::template phase<KIND, ph, decltype(st), false, true>(a, planes, bout, tid, st);
			if constexpr (ph + 1 < S::NPH) __syncthreads();
		}
	});
}

// ---- channel lines (dct_spec.h RowChanSpecT, chan_work): one workgroup per (line, channel) of an interleaved line ----
template <class S, int KIND, bool PLAIN>
__global__ void __launch_bounds__(S::T, S::WPE) row_chan_kernel(const typename S::PA a_, int lines)
{
	const typename S::PA a = PLAIN ? plain_args(a_) : a_;
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	typename S::CX *planes = reinterpret_cast<typename S::CX *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	int line, ch;
	chan_work<S::GS>(blockIdx.x, lines, line, ch);
	long long bin, bout;
	row_base(a, line, bin, bout);
	bin += ch; bout += ch;
	const uint8_t *zf = a.zflags ? a.zflags + (line & 1) * a.zhalf : nullptr;     // see row_spec_kernel
	S::template prefetch<KIND>(a, bin, tid, st, nullptr, zf, ch);
	S::template phase<KIND, 0>(a, planes, bout, tid, st);
	__syncthreads();
	static_for<1, S::NPH>([&](auto ph) {
		{
			S::template phase<KIND, ph>(a, planes, bout, tid, st);
			if constexpr (ph + 1 < S::NPH) __syncthreads();
		}
	});
}

// ---- lines that fill a CU's LDS on their own (7680 x 3 floats, 3840 x 3 doubles: 92 KB -> ONE workgroup per CU) ----
// With a single resident workgroup nothing overlaps its load, butterfly and store phases.  This variant is persistent (one workgroup
// per CU walks the lines) and software-pipelined at no register cost: the line's samples wait in State::pre only until phase 0 has put
// them into LDS, so the NEXT line is fetched into the same registers right after phase 0 and lands during the butterfly stages.
// vmcnt is in-order, so any vector-memory load between that prefetch and its use would wait for it: the stage twiddles W (the only
// loads of phases 1 .. NS + 1) are therefore kept in LDS (30 KB beside the 92 KB line; one workgroup per CU either way).
template <class S> constexpr size_t persist_lds() { return S::LDS + sizeof(typename S::CX) * (size_t)S::L; }
template <class S> constexpr bool persist_ok() { return S::C > 1 && S::LDS > 80 * 1024 && persist_lds<S>() <= 160 * 1024; }
template <class S, int KIND>
__global__ void __launch_bounds__(S::T, S::WPE) row_persist_kernel(const typename S::PA a_, int nwork)
{
	typedef typename S::CX CX;
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	CX *planes = reinterpret_cast<CX *>(lds);
	CX *wtab = reinterpret_cast<CX *>(lds + S::LDS);
	const int tid = threadIdx.x;
	for (int i = tid; i < S::L; i += S::T) wtab[i] = a_.W[i];
	typename S::PA a = plain_args(a_);
	a.W = wtab;
	typename S::template State<KIND> st;
	int work = blockIdx.x;
	long long bin, bout;
	row_base(a, work, bin, bout);
	S::template prefetch<KIND>(a, bin, tid, st, nullptr, nullptr);
	__syncthreads();                                         // the twiddle table is in place
	while (work < nwork) {
		// the thread index is re-made opaque every iteration: everything the phases derive from it (LDS addresses, twiddle indices)
		// is loop-invariant, and hoisted out of the loop it would be carried in registers across all of it (128 VGPRs + scratch)
		int t = tid; asm volatile("" : "+v"(t));
		S::template phase<KIND, 0>(a, planes, bout, t, st);
		__syncthreads();
		const int next = work + (int)gridDim.x;
		const long long bout_cur = bout;
		if (next < nwork) {                                  // uniform: the next line's loads go out now and land behind the stages
			row_base(a, next, bin, bout);
			S::template prefetch<KIND>(a, bin, t, st, nullptr, nullptr);
		}
		static_for<1, S::NPH>([&](auto ph) {
			int u = t; asm volatile("" : "+v"(u));
			S::template phase<KIND, ph>(a, planes, bout_cur, u, st);
			__syncthreads();
		});
		work = next;
	}
}

// ---- out = A(in_a) + B(in_b): two REDFT01 row transforms of the same output line in ONE workgroup (dspfft_execute_sum2) ----
// zoom's x stage by fast transforms is the sum of a cosine and a sine part (zoom_fft.hip), each a REDFT01 of its own zero-padded input:
// as two launches the second reads the frame back to add into it; here the first part's line waits in registers (RowSpecG::Hold) while
// the second runs through the same LDS, and the frame is written once.  Each part keeps its own scale, input window and output sign.
template <class S>
__global__ void __launch_bounds__(S::T, 1) row_sum2_kernel(const typename S::PA a, const typename S::PA b)
{
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	typename S::CX *planes = reinterpret_cast<typename S::CX *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND_REDFT01> st, st2;
	long long bin, bout, bin2, bout2;
	row_base(a, blockIdx.x, bin, bout);
	row_base(b, blockIdx.x, bin2, bout2);
	S::template prefetch<KIND_REDFT01>(a, bin, tid, st);
	S::template prefetch<KIND_REDFT01>(b, bin2, tid, st2);
	typename S::Hold hold;
	static_for<0, S::NPH - 1>([&](auto ph) {
		S::template phase<KIND_REDFT01, ph>(a, planes, bout, tid, st);
		__syncthreads();
	});
	S::template final01_hold<1>(a, planes, bout, tid, hold);
	__syncthreads();
	int t = tid; asm volatile("" : "+v"(t));             // keeps the second transform's index arithmetic from being hoisted above the first
	static_for<0, S::NPH - 1>([&](auto ph) {
		S::template phase<KIND_REDFT01, ph>(b, planes, bout2, t, st2);
		__syncthreads();
	});
	S::template final01_hold<2>(b, planes, bout2, t, hold);
}

// ---- zoom's x stage with the cosine and the sine part of a line as the two halves of a Pk2 (dct_duo.h ZoomXLeanT): the channels of a
// line one after another through a 16-byte-slot plane, three phases each -- first stage fed from global memory, middle stages in the
// plane, last stage to registers; a thread's pixels wait in registers until the last channel stores them whole ----
template <class S, int C, int NSRC, int WPE, bool CLIP>
__global__ void __launch_bounds__(S::T, WPE) zoomx_lean_kernel(const ZoomXArgs a)
{
	typedef ZoomXLeanT<S, C, NSRC, CLIP> Z;
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	typename S::V *buf = reinterpret_cast<typename S::V *>(lds);
	const int tid = threadIdx.x;
	const long long bin = (long long)blockIdx.x * a.in_pitch, bout = (long long)blockIdx.x * a.out_pitch;
	PassArgs w;
	w.W = a.W;                                               // the stages read nothing else
	typename Z::State st;
	// a real loop over the channels: unrolled, the compiler interleaves them and carries one channel's addresses and twiddles through the
	// next (80 -> 176 -> 220 VGPRs for 1 -> 2 -> 3 unrolled channels of the first cut)
	DSP_STAMP(0);
	if constexpr (Z::WHOLE) {
		typename Z::Pixels px;
		Z::load_pixels(a, bin, tid, px);
		static_for<0, C>([&](auto c) {
			int t = tid; asm volatile("" : "+v"(t));
			Z::template phase_a_held<c>(a, w, buf, t, px);
			__syncthreads();
			DSP_STAMP(1 + c * 4);
			static_for<1, S::NS - 1>([&](auto I) {
				Z::template phase_b<I>(w, buf, t);
				__syncthreads();
			});
			DSP_STAMP(2 + c * 4);
			typename Z::Ex e;
			Z::phase_c(buf, t, e);
			Z::template phase_c_emit_ch<c>(a, bout, t, e, st);
			DSP_STAMP(3 + c * 4);
			if constexpr (c + 1 < C) __syncthreads();
			DSP_STAMP(4 + c * 4);
		});
		return;
	}
#pragma nounroll
	for (int c = 0; c < C; c++) {
		int t = tid; asm volatile("" : "+v"(t));
		Z::phase_a(a, w, buf, bin, c, t);
		__syncthreads();
		DSP_STAMP(1 + c * 4);
		static_for<1, S::NS - 1>([&](auto I) {
			Z::template phase_b<I>(w, buf, t);
			__syncthreads();
		});
		DSP_STAMP(2 + c * 4);
		typename Z::Ex e;
		Z::phase_c(buf, t, e);
		Z::phase_c_emit(a, bout, c, t, e, st);
		DSP_STAMP(3 + c * 4);
		if (c + 1 < C) __syncthreads();                      // the plane is the next channel's
		DSP_STAMP(4 + c * 4);
	}
}

// ---- chirp-z rows (dct_czt.h): one workgroup per line, the line's circular convolution in one LDS plane ----
template <class S> constexpr int czt_waves_per_simd() { const int w = (int)((160 * 1024) / S::LDS) * S::T / 256; return w < 1 ? 1 : w > 4 ? 4 : w; }
template <class S>
__global__ void __launch_bounds__(S::T, czt_waves_per_simd<S>()) czt_rows_kernel(const CztArgs a)
{
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	cf *plane = reinterpret_cast<cf *>(lds);
	const int tid = threadIdx.x;
	long long bin, bout;
	S::base(a, blockIdx.x, bin, bout);
	PassArgs w;
	w.W = a.W;                                               // the stages read nothing else
	static_for<0, S::NPH>([&](auto ph) {
		S::template phase<ph>(a, w, plane, bin, bout, tid);
		if constexpr (ph + 1 < S::NPH) __syncthreads();
	});
}
// FFT_P of the table in a.atab (P entries), in the slot order the forward stages leave: what czt_rows_kernel multiplies by in its MID phase
template <class S>
__global__ void __launch_bounds__(S::T, 1) czt_spectrum_kernel(const CztArgs a, cf *hspec)
{
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	cf *plane = reinterpret_cast<cf *>(lds);
	const int tid = threadIdx.x;
	PassArgs w;
	w.W = a.W;
	S::f0(a, plane, 0, tid);
	__syncthreads();
	static_for<1, S::NS - 1>([&](auto I) { S::template fwd<I>(w, plane, tid); __syncthreads(); });
	S::flast(plane, tid);
	__syncthreads();
	for (int i = tid; i < S::P; i += S::T) hspec[i] = plane[S::F::padded(i)];
}
template <class S>
int launch_czt_rows(const CztArgs &a, void *stream)
{
	static DevOnce once;
	if (int rc = allow_lds_dev(once, S::LDS, czt_rows_kernel<S>, czt_spectrum_kernel<S>)) return rc;
	hipLaunchKernelGGL((czt_rows_kernel<S>), dim3(a.lines), dim3(S::T), S::LDS, (hipStream_t)stream, a);
	HIPCHK(hipGetLastError());
	return 0;
}
template <class S>
int launch_czt_spectrum(const CztArgs &a, cf *hspec, void *stream)
{
	static DevOnce once;
	if (int rc = allow_lds_dev(once, S::LDS, czt_rows_kernel<S>, czt_spectrum_kernel<S>)) return rc;
	hipLaunchKernelGGL((czt_spectrum_kernel<S>), dim3(1), dim3(S::T), S::LDS, (hipStream_t)stream, a, hspec);
	HIPCHK(hipGetLastError());
	return 0;
}

// the same with 8-bit input (REDFT10) or quantised 8-bit output (REDFT01): planar rows only
// Short lines (motion's 1920 / 960 sample rows: two waves a line, 8 KB of LDS) are asked to fit 64 registers so that eight waves share a SIMD: their phases are
// a few hundred cycles of work each between barriers and global accesses, and only other workgroups fill the gaps (REDFT10 of config 5's luma clip,
// kernel alone: 896 -> 813 us; 74 -> 64 VGPRs, no scratch).  Asked where the kernel FITS 64 registers: longer lines keep the allocator's choice (at 64 they spill), and
// `make -C dspfun_amd/csrc check-scratch` fails on scratch in these kernels.  (960-sample REDFT10 lines spilled 12 bytes a lane beside the radix-16 butterfly of their
// round-5 radices (2, 16, 15) and ran at six waves; with (4, 8, 15) they fit -- six or eight waves measure the same, 3.91-3.94 ms per clip.)
template <class S, int KIND> constexpr int u8_waves_per_simd() { return S::N >= 2048 ? S::WPE : 8; }
template <class S, int KIND>
__global__ void __launch_bounds__(S::T, (u8_waves_per_simd<S, KIND>())) row_spec_u8_kernel(const typename S::PA a_, const U8IO io_)
{
	// the 8-bit ends belong to motion's plain roundtrip: no owner-id mask, tile flags, accumulation, input window or modulation (launch_row_spec_u8 refuses
	// them), so their per-load selects fold away as in the plain row kernels (round 5: they were a third of this kernel's first phase)
	const typename S::PA a = plain_args(a_);
	// this instantiation's 8-bit end is known: the phases' float alternatives (and what merging the two paths cost: a dynamically indexed
	// copy of the line's samples in scratch behind a full wait for the loads) fold away
	U8IO io = io_;
	if constexpr (KIND == KIND_REDFT10) { __builtin_assume(io.in != nullptr); io.out = nullptr; } else { __builtin_assume(io.out != nullptr); io.in = nullptr; }
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	typename S::CX *planes = reinterpret_cast<typename S::CX *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	row_base(a, blockIdx.x, bin, bout);
	DSP_STAMP(0);
	S::template prefetch<KIND>(a, bin, tid, st, &io);
	S::fetch_stage_twiddles(a, tid, st);
	S::template phase<KIND, 0>(a, planes, bout, tid, st, &io);
	__syncthreads();
	DSP_STAMP(1);
	static_for<1, S::NPH>([&](auto ph) {
		{
			S::template phase<KIND, ph, decltype(st), false, true>(a, planes, bout, tid, st, &io);
			if constexpr (ph + 1 < S::NPH) __syncthreads();
			DSP_STAMP(1 + ph);
		}
	});
}

template <class S, int KIND, bool PLAIN>
__global__ void __launch_bounds__(S::T, S::WPE) col_spec_kernel(const typename S::PA a_)
{
	const typename S::PA a = PLAIN ? plain_args(a_) : a_;
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	typename S::V *buf = reinterpret_cast<typename S::V *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	int tile;
	bool hit = false;
	S::base(a, blockIdx.x, bin, bout, tile);
	// sparse scan frames, prepared owner ids: this frame lies outside the tile's id range -- nothing to read at all
	if (a.zflags && a.mask && a.zranges && (a.mask_id < a.zranges[2 * tile] || a.mask_id > a.zranges[2 * tile + 1])) {
		if (tid == 0) a.zflags[tile] = 0;
		return;
	}
	S::template prefetch<KIND>(a, bin, tid, st, hit);
	// a tile none of whose coefficients belongs to this frame transforms to zeros -- say so and stop
	if (a.zflags && a.mask) {
		const int nz = __syncthreads_or(hit);
		if (tid == 0) a.zflags[tile] = (uint8_t)(nz != 0);
		if (!nz) return;
	}
	S::template phase<KIND, 0>(a, buf, bout, tid, st);
	__syncthreads();
	// (a REDFT01 pass that closes in its last stage has two empty phases: no barrier around those)
	static_for<1, S::NPH>([&](auto ph) {
		if (!S::template skip_phase<KIND>(a, ph)) {
			S::template phase<KIND, ph>(a, buf, bout, tid, st);
			if (S::template barrier_after<KIND>(a, ph)) __syncthreads();
		}
	});
}

// ---- outer radix-2 split of the column axis (dct_spec.h, ColHalfSpec) ----
// ROW side: one workgroup per row PAIR (y1 = 2n, y2 = N-1-2n of the split axis, which is the row pass's batch dimension 0):
// loads both lines, transforms (r1 + r2) into line y1 and (r1 - r2) into line y2.  The second line's samples wait in
// registers while the first is transformed.
// waves per SIMD to ask of the register allocator: as many workgroups as the line's LDS allows, capped at 6 (80 VGPRs) -- left
// alone the allocator spends 115 VGPRs on the 3840-pixel pair kernel (plain: 56) and a third workgroup no longer fits a CU
#ifndef DSP_PAIR_WPE_MAX
#define DSP_PAIR_WPE_MAX 6
#endif
#ifndef DSP_PAIR_WPE_LONG
#define DSP_PAIR_WPE_LONG 4
#endif
template <class S> constexpr int pair_waves_per_simd()
{
	const int wgs = (int)((160 * 1024) / S::LDS), w = wgs * S::T / 256;
	// long planar lines (7680 x 1: two lines of 15-30 samples per thread waiting beside the butterfly registers) need 108-141 registers: at
	// 5-6 waves per SIMD (80-96) the kernel spilled 104-260 bytes per lane; 4 waves (128) hold the 512-thread kernel's 108-111
	const int cap = (S::C == 1 && S::N >= 7680) ? DSP_PAIR_WPE_LONG : DSP_PAIR_WPE_MAX;
	return w < 1 ? 1 : w > cap ? cap : w;
}
// PLAIN: see row_spec_kernel (the 8K roundtrip's instantiation; the fused scan step of BASELINE config 4 runs the general one)
template <class S, int KIND, bool PLAIN = false>
__global__ void __launch_bounds__(S::T, pair_waves_per_simd<S>()) row_pair_kernel(const typename S::PA a_)
{
	const typename S::PA a = PLAIN ? plain_args(a_) : a_;
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	typename S::CX *planes = reinterpret_cast<typename S::CX *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st, st2;
	const int pairs = a.nb0 >> 1;
	const int i1 = blockIdx.x / pairs, n = blockIdx.x - i1 * pairs;
	const int y1 = 2 * n, y2 = a.nb0 - 1 - 2 * n;
	const long long bin1 = y1 * a.sb0_in + i1 * a.sb1_in, bin2 = y2 * a.sb0_in + i1 * a.sb1_in;
	const long long bout1 = y1 * a.sb0_out + i1 * a.sb1_out, bout2 = y2 * a.sb0_out + i1 * a.sb1_out;
	S::template prefetch<KIND>(a, bin1, tid, st, nullptr, a.zflags);              // masked loads when this is the first pass of a fused scan step,
	S::template prefetch<KIND>(a, bin2, tid, st2, nullptr, a.zflags ? a.zflags + a.zhalf : nullptr);   // tile flags when it follows a masked half-tile pass
	typedef typename S::Re Re;
	constexpr int NPRE = (int)(sizeof(st.pre) / sizeof(Re));
	Re cur[NPRE], diff[NPRE];
	DSP_STAMP(0);
	static_for<0, NPRE>([&](auto i) { const Re p = st.pre[i], q = st2.pre[i]; cur[i] = p + q; diff[i] = p - q; });
	DSP_STAMP(1);
	// a real loop (not two copies of the phases) whose only loop-carried values are the waiting line's samples: the compiler
	// otherwise hoists the second transform's index arithmetic and twiddle loads above the first, or carries the last stage's
	// butterfly registers around the loop, and keeps them live across every barrier (133 / 115 VGPRs: a resident workgroup fewer)
#pragma nounroll
	for (int rep = 0; rep < 2; rep++) {
		const long long bout = rep ? bout2 : bout1;
		int t = tid; asm volatile("" : "+v"(t));
		typename S::template State<KIND> w;
		static_for<0, NPRE>([&](auto i) { w.pre[i] = cur[i]; });
		// (the twiddles are fetched again per line instead of being carried around the loop: 8 more live registers made the planar kernel spill)
		static_for<0, S::K_ROUNDS>([&](auto i) { const int k = t + i * S::T; if ((i + 1) * S::T <= S::L / 2 + 1 || k <= S::L / 2) w.tw[i] = a.T[k]; });
		static_for<0, S::NPH>([&](auto ph) {
			S::template phase<KIND, ph, decltype(w), true>(a, planes, bout, t, w);
			__syncthreads();
			DSP_STAMP(2 + rep * 8 + ph);
		});
		static_for<0, NPRE>([&](auto i) { cur[i] = diff[i]; });
	}
}

// ---- row pairs of lines that fill a CU's LDS on their own (7680 x 3 floats: 92 KB, ONE workgroup per CU), the second line fetched ahead (round 6) ----
// row_pair_kernel loads both lines, waits, and transforms (r1 + r2) and (r1 - r2).  With a single resident workgroup nothing overlaps that wait, and it
// is a third of a pair's life: 184 KB through a CU's in-order memory pipe behind the previous pair's 184 KB of stores (tools/kstamp: wave 0 waits 5.5K
// clocks, the slowest wave 11K more, of 50K).  The transform is linear, so the pair's butterfly can sit on the OUTPUT side: out1 = T(r1) + T(r2),
// out2 = T(r1) - T(r2).  Then the lines are needed one at a time: a persistent workgroup (one per CU walks the pairs) waits for r1 only, fetches r2
// right behind phase 0 of r1 into the registers that phase has just emptied -- it lands under r1's stages -- keeps T(r1)'s outputs in registers (as
// the difference line waits today) and stores both output lines from r2's closing phase.  The next pair's r1 is requested behind those stores.
// The stage twiddles W and T[k] live in LDS (45 KB beside the 92 KB plane): vmcnt is in-order, a vector-memory load between the prefetch and its use
// would wait for it (see row_persist_kernel).  Plain passes only (the fused scan step's masked / accumulating pair pass keeps row_pair_kernel).
// (Fetching BOTH lines one transform ahead and storing the pair behind the next line's phase 0 was built first: the held output line costs 24 more
// registers through every phase, and 184 KB of stores in one burst block the issuing waves for 15-18K clocks: 236 / 232 us, profiles/r06_8k_pair_pipe.txt.)
#ifndef DSP_PAIR_PIPE_T
#define DSP_PAIR_PIPE_T 768
#endif
// `v` holds nothing worth keeping from here on: every register of it is (re)defined by an empty asm statement, which costs no instruction and ends
// the old values' live ranges.  State that is written under per-thread conditions (the last, partly filled round of a phase) otherwise stays alive
// around a persistent kernel's whole loop as far as the register allocator can tell.
template <int I, int N> __device__ __forceinline__ void forget_regs(float *p)
{
	if constexpr (I < N) { float f; asm volatile("" : "=v"(f)); p[I] = f; forget_regs<I + 1, N>(p); }
}
template <class V> __device__ __forceinline__ void forget(V &v)
{
	static_assert(sizeof(V) % 4 == 0, "registers");
	forget_regs<0, (int)(sizeof(V) / 4)>(reinterpret_cast<float *>(&v));
}
template <class S> constexpr size_t pair_pipe_lds() { return S::LDS + sizeof(typename S::CX) * (size_t)(S::L + S::L / 2 + 1); }
template <class S> constexpr bool pair_pipe_ok() { return S::C > 1 && S::LDS > 80 * 1024 && pair_pipe_lds<S>() <= 160 * 1024; }
// SCAN: the pair pass of the fused scan step (scan/scan.c:446-459, REDFT01 behind the masked half-tile pass, work buffer -> running sum).  Its lines are read with
// that pass's tile flags (PassGeom::zflags; copied into LDS, kPairPipeFlagBytes behind the tables: a flag fetched from memory in front of every pixel load would
// have the load wait for everything the pipe holds) and are mostly zeros (a frame keeps 1/32 of the coefficients), so there is little to hide on the input side;
// what row_pair_kernel waits for is `sum += image`: the old values of an output line are loaded in its closing phase, right before they are added -- twice per
// pair, 1.19 GB per frame at 4.3 TB/s.  Here the old values of the first output line are requested behind phase 0 of r2 (30 registers that hold nothing else
// then: the next pair's r1 goes out behind the stores, as in REDFT10's plain form) and have landed when r2's closing phase adds them; the second line's go out
// at the head of that phase and land while the first line is stored.
constexpr int kPairPipeFlagBytes = 4096;
template <class PA> __host__ __device__ inline PA scan_pair_args(const PA &a_)
{
	PA a = a_;
	a.mask = nullptr; a.zranges = nullptr; a.win_lo = a.win_hi = 0; a.alt_out = 0; a.in_mul = nullptr; a.in_rev = 0;
	return a;
}
template <class PA> static inline bool is_scan_pair(const PA &a) { return !a.mask && a.win_hi <= 0 && !a.alt_out && !a.in_mul && !a.in_rev && (!a.zflags || 2 * a.zhalf <= kPairPipeFlagBytes); }
template <class S, int KIND, bool EARLY = false, bool SCAN = false>
__global__ void __launch_bounds__(S::T, pair_waves_per_simd<S>()) row_pair_pipe_kernel(const typename S::PA a_, int nwork)
{
	typedef typename S::CX CX;
	typedef typename S::Re Re;
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	CX *planes = reinterpret_cast<CX *>(lds);
	CX *wtab = reinterpret_cast<CX *>(lds + S::LDS), *ttab = wtab + S::L;
	const int tid = threadIdx.x;
	for (int i = tid; i < S::L; i += S::T) wtab[i] = a_.W[i];
	for (int i = tid; i <= S::L / 2; i += S::T) ttab[i] = a_.T[i];
	typename S::PA a = SCAN ? scan_pair_args(a_) : plain_args(a_);
	a.W = wtab; a.T = ttab;
	// the tile flags of the two halves (even rows: r1's, odd rows: r2's), or null
	uint8_t *ftab = reinterpret_cast<uint8_t *>(ttab + S::L / 2 + 1);
	if constexpr (SCAN) { if (a.zflags) for (int i = tid; i < 2 * a.zhalf; i += S::T) ftab[i] = a.zflags[i]; }
	const uint8_t *zf1 = (SCAN && a.zflags) ? ftab : nullptr, *zf2 = (SCAN && a.zflags) ? ftab + a.zhalf : nullptr;
	unsigned long long bits1 = ~0ull, bits2 = ~0ull;     // SCAN: which of this thread's samples of an even / odd line are to be read (RowSpecG::flag_bits01)
	const int pairs = a.nb0 >> 1;
	// element offset of line `second` (0: y1 = 2n, 1: y2 = nb0 - 1 - 2n) of pair `work`
	auto line_in = [&](int work, int second) { const int i1 = work / pairs, n = work - i1 * pairs; return (long long)(second ? a.nb0 - 1 - 2 * n : 2 * n) * a.sb0_in + (long long)i1 * a.sb1_in; };
	auto line_out = [&](int work, int second) { const int i1 = work / pairs, n = work - i1 * pairs; return (long long)(second ? a.nb0 - 1 - 2 * n : 2 * n) * a.sb0_out + (long long)i1 * a.sb1_out; };
	typename S::template State<KIND> st;
	typename S::template OutHold<KIND> h1;           // T(r1)'s outputs
	typename S::template OutHold<KIND> old1, old2;   // SCAN: the running sum's old values of the pair's two output lines (dead otherwise)
	__syncthreads();                                 // the tables are in place
	// REDFT10 with its next pair requested in FRONT of the pair's stores: every store unconditional (RowSpecG::final_each UNCOND), the closing phase's twiddles
	// read from the LDS table where they are used (item L/2's for a thread beyond the last item; kept in registers across the loop they cost a spill among the stores)
	constexpr bool UNC = KIND == KIND_REDFT10 && EARLY;
	if constexpr (SCAN && KIND == KIND_REDFT01) { if (a.zflags) { bits1 = S::flag_bits01(zf1, a.zshift, tid); bits2 = S::flag_bits01(zf2, a.zshift, tid); } }
	// the loads of a line: plain, or (SCAN) with this thread's precomputed flags
	typedef Re PixV __attribute__((ext_vector_type(S::C == 3 ? 3 : S::C == 2 ? 2 : S::C == 4 ? 4 : 1)));
	constexpr bool FLY = KIND == KIND_REDFT01 && !SCAN;
	PixV fly[FLY ? 4 * S::K_ROUNDS : 1];             // REDFT01's line in flight
	auto fetch = [&](long long bin, int t, auto SECOND_LINE) __attribute__((always_inline)) {
		if constexpr (SCAN && KIND == KIND_REDFT01) S::prefetch01_bits(a, bin, t, st, SECOND_LINE ? bits2 : bits1);
		else if constexpr (FLY) {
			// RowSpecG::prefetch's loads with the item index CLAMPED instead of tested: under `if (k <= L/2)` the ragged last round's loads end in a join at
			// which the compiler waits for them -- and, vmcnt being in order, for every load of the line issued before them: the prefetch was waited for
			// where it was issued.  A thread beyond the last item loads item L/2's pixels again and phase 0 ignores them.
			// The pixels land in `fly`, whole 12-byte vectors as the load instruction writes them, and are taken apart where the line is consumed (head): written
			// straight into State::pre's scalars the allocator could not keep ten of the twelve vectors where they landed and copied them -- behind a wait -- at once.
			static_for<0, S::K_ROUNDS>([&](auto i) {
				const int k0 = t + i * S::T, k = k0 <= S::L / 2 ? k0 : S::L / 2;
				st.tw[i] = a.T[k];
				const Re *p = a.in + bin;
				__builtin_memcpy(&fly[i * 4 + 0], p + (long long)k * S::GS, S::C * sizeof(Re));
				__builtin_memcpy(&fly[i * 4 + 1], p + (long long)(k ? S::N - k : 0) * S::GS, S::C * sizeof(Re));
				__builtin_memcpy(&fly[i * 4 + 2], p + (long long)(S::L - k) * S::GS, S::C * sizeof(Re));
				__builtin_memcpy(&fly[i * 4 + 3], p + (long long)(S::L + k) * S::GS, S::C * sizeof(Re));
			});
		} else S::template prefetch<KIND>(a, bin, t, st, nullptr, nullptr);
	};
	// one pair per iteration, its two lines spelled out (SECOND is a compile-time constant)
	// `head`: phase 0 of a line (consumes the line's loads); `rest`: everything behind it.  The loop below is ROTATED -- r1's head of the next pair closes the
	// iteration instead of opening it -- so that the wait in front of it lies on ONE path, behind this pair's stores, and the compiler writes it as
	// vmcnt(stores behind the loads): at the top of the loop it would merge with the path from the first prefetch (nothing behind the loads) into vmcnt(0)
	// = the whole store queue.
	auto head = [&](auto SECOND) __attribute__((always_inline)) {
		// (the thread index is re-made opaque for every line: see row_persist_kernel)
		int t = tid; asm volatile("" : "+v"(t));
		forget(st.x);
		DSP_STAMP(SECOND ? 10 : 0);
		if constexpr (FLY) static_for<0, 4 * S::K_ROUNDS>([&](auto n) { static_for<0, S::C>([&](auto c) { st.pre[n * S::C + c] = fly[n][(int)decltype(c)::value]; }); });
		S::template phase<KIND, 0, decltype(st), true>(a, planes, 0, t, st);
		DSP_STAMP(SECOND ? 11 : 1);
	};
	auto rest = [&](auto SECOND, long long in_next, bool has_next, long long pb1, long long pb2) __attribute__((always_inline)) {
		int t = tid; asm volatile("" : "+v"(t));
		if constexpr (!SECOND) {                     // r2: lands behind r1's stages
			forget(h1); forget(st.pre);
			fetch(in_next, t, std::integral_constant<bool, true>());
		} else if constexpr (EARLY) {                // the next pair's r1 behind r2's stages, IN FRONT of this pair's stores
			forget(st.pre);
			if (has_next) fetch(in_next, t, std::integral_constant<bool, false>());
		} else if constexpr (SCAN && KIND == KIND_REDFT01) {
			if (a.accumulate) {                      // sum += image (scan.c:451-459): the first line's old values, landing behind r2's stages
				forget(old1);
				static_for<0, S::PIX_ROUNDS>([&](auto i) {
					const int x = t + i * S::T;
					if ((i + 1) * S::T <= S::N || x < S::N) old1.v[i] = load_pix<S::C, Re>(a.out + pb1 + (long long)x * S::GS);
				});
			}
		}
		__syncthreads();
		DSP_STAMP(SECOND ? 12 : 2);
		static_for<1, S::NS + 2>([&](auto ph) {
			int u = t; asm volatile("" : "+v"(u));
			S::template phase<KIND, ph, decltype(st), true>(a, planes, 0, u, st);
			__syncthreads();
			DSP_STAMP((SECOND ? 12 : 2) + ph);
		});
		int u = t; asm volatile("" : "+v"(u));
		if constexpr (UNC) static_for<0, S::K_ROUNDS>([&](auto ri) { st.tw[ri] = ttab[S::template tw_index<true>(u + ri * S::T)]; });
		auto each = [&](auto &&f) __attribute__((always_inline)) { S::template final_each<KIND, UNC>(a, planes, u, st, f); };
		if constexpr (!SECOND) {
			each([&](auto slot, long long, Pix<S::C, Re> v) { h1.v[slot] = v; });
		} else {
			bool done = false;
			if constexpr (SCAN && KIND == KIND_REDFT01) {
				if (a.accumulate) {
					// the second line's old values go out now and land while the first line is added and stored (two sweeps over the plane: REDFT01's closing
					// phase is one LDS read and a product per sample; both lines' old values held through r2's stages would be 30 registers too many: 56 spilled)
					forget(old2);
					static_for<0, S::PIX_ROUNDS>([&](auto i) {
						const int x = u + i * S::T;
						if ((i + 1) * S::T <= S::N || x < S::N) old2.v[i] = load_pix<S::C, Re>(a.out + pb2 + (long long)x * S::GS);
					});
					S::template final_each<KIND>(a, planes, u, st, [&](auto slot, long long off, Pix<S::C, Re> v) {
						Pix<S::C, Re> o;
						static_for<0, S::C>([&](auto c) { o.v[c] = (h1.v[slot].v[c] + v.v[c]) + old1.v[slot].v[c]; });
						store_pix<S::C, Re>(a.out + pb1 + off, o);
					});
					S::template final_each<KIND>(a, planes, u, st, [&](auto slot, long long off, Pix<S::C, Re> v) {
						Pix<S::C, Re> o;
						static_for<0, S::C>([&](auto c) { o.v[c] = (h1.v[slot].v[c] - v.v[c]) + old2.v[slot].v[c]; });
						store_pix<S::C, Re>(a.out + pb2 + off, o);
					});
					done = true;
				}
			}
			if (!done)
				each([&](auto slot, long long off, Pix<S::C, Re> v) {
					Pix<S::C, Re> o1, o2;
					static_for<0, S::C>([&](auto c) { o1.v[c] = h1.v[slot].v[c] + v.v[c]; o2.v[c] = h1.v[slot].v[c] - v.v[c]; });
					store_pix<S::C, Re>(a.out + pb1 + off, o1);
					store_pix<S::C, Re>(a.out + pb2 + off, o2);
				});
			if constexpr (!EARLY) {
				forget(st.pre);
				if (has_next) fetch(in_next, u, std::integral_constant<bool, false>());     // the next pair's r1, behind this pair's stores
			}
		}
		__syncthreads();                             // the plane is the next line's
		DSP_STAMP(SECOND ? 19 : 9);
	};
	int work = blockIdx.x;
	const std::integral_constant<bool, false> FIRST;
	const std::integral_constant<bool, true> SECOND_;
	if constexpr (EARLY) {
		if (work < nwork) { fetch(line_in(work, 0), tid, FIRST); head(FIRST); }
		while (work < nwork) {
			const int next = work + (int)gridDim.x;
			rest(FIRST, line_in(work, 1), true, 0, 0);
			head(SECOND_);
			rest(SECOND_, next < nwork ? line_in(next, 0) : 0, next < nwork, line_out(work, 0), line_out(work, 1));
			work = next;
			if (work < nwork) head(FIRST);
		}
	} else {
		// (the next pair's r1 is requested BEHIND the stores: its wait is for the whole queue on any path, and the rotated loop costs the scan form 36 spilled registers)
		if (work < nwork) fetch(line_in(work, 0), tid, FIRST);
		while (work < nwork) {
			const int next = work + (int)gridDim.x;
			head(FIRST);
			rest(FIRST, line_in(work, 1), true, 0, 0);
			head(SECOND_);
			rest(SECOND_, next < nwork ? line_in(next, 0) : 0, next < nwork, line_out(work, 0), line_out(work, 1));
			work = next;
		}
	}
}

// COL side: one workgroup per half tile (N/2 rows x K floats)
template <class S, int KIND, bool PLAIN = false>
__global__ void __launch_bounds__(S::T, S::WPE) col_half_kernel(const typename S::PA a_)
{
	const typename S::PA a = PLAIN ? plain_args(a_) : a_;
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	typename S::V *buf = reinterpret_cast<typename S::V *>(lds);
	const int tid = threadIdx.x;
	typename S::template State<KIND> st;
	long long bin, bout;
	int h, tile;
	bool hit = false;
	S::base(a, blockIdx.x, bin, bout, h, tile);
	if (a.zflags && a.mask && a.zranges && (a.mask_id < a.zranges[2 * tile] || a.mask_id > a.zranges[2 * tile + 1])) {   // see col_spec_kernel
		if (tid == 0) a.zflags[tile] = 0;
		return;
	}
	DSP_STAMP(0);
	S::template prefetch<KIND>(a, bin, h, tid, st, hit);
	if (a.zflags && a.mask) {
		const int nz = __syncthreads_or(hit);
		if (tid == 0) a.zflags[tile] = (uint8_t)(nz != 0);
		if (!nz) return;
	}
	static_for<0, S::NPH>([&](auto ph) {
		if (!S::template skip_phase<KIND>(a, ph)) {           // see col_spec_kernel
			S::template phase<KIND, ph>(a, buf, bout, h, tid, st);
			if (S::template barrier_after<KIND>(a, ph)) __syncthreads();
			DSP_STAMP(1 + ph);
		}
	});
}

// the fused column roundtrip (forward REDFT10 -> motion filter -> inverse REDFT01 along the tile's axis in one launch): spec_fused.h
template <class S>
__global__ void __launch_bounds__(S::T, rt_waves_per_simd<S>()) col_roundtrip_kernel(const typename S::PA af, const typename S::PA ai, const FilterOp filt, unsigned long long *coded)
{
	extern __shared__ __attribute__((aligned(32))) unsigned char lds[];
	col_roundtrip_body<S>(lds, af, ai, filt, coded);
}


template <class S, int KIND>
int launch_row_spec(const typename S::PA &a, int nwork, void *stream)
{
	typedef typename chan_lines_of<typename S::Re, S::N, S::C>::type CH;
	constexpr bool whole_fits = S::LDS <= 160 * 1024;       // a 7680 x 3 double line does not: it exists as channel lines only
	if constexpr (!std::is_void<CH>::value) {
		// DSPFFT_ROW_CHAN=0 keeps the interleaved line in one workgroup (A/B runs)
		if (chan_lines_enabled() || !whole_fits) {
			static DevOnce conce;
			if (int c_rc = allow_lds_dev(conce, CH::LDS, row_chan_kernel<CH, KIND, false>, row_chan_kernel<CH, KIND, true>)) return c_rc;
			if (is_plain(a)) hipLaunchKernelGGL((row_chan_kernel<CH, KIND, true>), dim3(nwork * CH::GS), dim3(CH::T), CH::LDS, (hipStream_t)stream, a, nwork);
			else hipLaunchKernelGGL((row_chan_kernel<CH, KIND, false>), dim3(nwork * CH::GS), dim3(CH::T), CH::LDS, (hipStream_t)stream, a, nwork);
			HIPCHK(hipGetLastError());
			return 0;
		}
	}
	if constexpr (!whole_fits) {
		return -1;          // (the interleaved kernels of such a line are never instantiated)
	} else {
		static DevOnce once;
		if (int lds_rc = allow_lds_dev(once, S::LDS, row_spec_kernel<S, KIND, false>, row_spec_kernel<S, KIND, true>)) return lds_rc;
		if constexpr (persist_ok<S>()) {
			// DSPFFT_ROW_PERSIST=0 keeps one workgroup per line (A/B runs)
			static const int on = []() { const char *e = getenv("DSPFFT_ROW_PERSIST"); return e ? atoi(e) : 1; }();
			const int cus = device_cus();
			if (on && is_plain(a) && nwork > cus) {
				static DevOnce ponce;
				if (int p_rc = allow_lds_dev(ponce, persist_lds<S>(), row_persist_kernel<S, KIND>)) return p_rc;
				hipLaunchKernelGGL((row_persist_kernel<S, KIND>), dim3(cus), dim3(S::T), persist_lds<S>(), (hipStream_t)stream, a, nwork);
				HIPCHK(hipGetLastError());
				return 0;
			}
		}
		if (is_plain(a)) hipLaunchKernelGGL((row_spec_kernel<S, KIND, true>), dim3(nwork), dim3(S::T), S::LDS, (hipStream_t)stream, a);
		else hipLaunchKernelGGL((row_spec_kernel<S, KIND, false>), dim3(nwork), dim3(S::T), S::LDS, (hipStream_t)stream, a);
		HIPCHK(hipGetLastError());
		return 0;
	}
}
template <class S, int KIND>
int launch_col_spec(const typename S::PA &a, int nwork, void *stream)
{
	static DevOnce once;
	if (int lds_rc = allow_lds_dev(once, S::LDS, col_spec_kernel<S, KIND, false>, col_spec_kernel<S, KIND, true>)) return lds_rc;
	if (is_plain(a)) hipLaunchKernelGGL((col_spec_kernel<S, KIND, true>), dim3(nwork), dim3(S::T), S::LDS, (hipStream_t)stream, a);
	else hipLaunchKernelGGL((col_spec_kernel<S, KIND, false>), dim3(nwork), dim3(S::T), S::LDS, (hipStream_t)stream, a);
	HIPCHK(hipGetLastError());
	return 0;
}
template <class S, int KIND>
int launch_row_pair(const typename S::PA &a, int npairs, void *stream)
{
	static DevOnce once;
	if (int lds_rc = allow_lds_dev(once, S::LDS, row_pair_kernel<S, KIND, false>, row_pair_kernel<S, KIND, true>)) return lds_rc;
	if constexpr (pair_pipe_ok<S>()) {
		// DSPFFT_PAIR_PIPE=0 keeps one workgroup per pair (A/B runs)
		static const int on = []() { const char *e = getenv("DSPFFT_PAIR_PIPE"); return e ? atoi(e) : 1; }();
		const int cus = device_cus();
		if constexpr (KIND == KIND_REDFT01) {
			// the fused scan step's accumulating pair pass (BASELINE config 4): DSPFFT_PAIR_PIPE_SCAN=0 keeps row_pair_kernel
			static const int scan_on = []() { const char *e = getenv("DSPFFT_PAIR_PIPE_SCAN"); return e ? atoi(e) : 1; }();
			if (on && scan_on && !is_plain(a) && is_scan_pair(a) && a.in != a.out && npairs > cus) {
				typedef typename S::template with_threads<DSP_PAIR_PIPE_T> SP;
				static DevOnce sonce;
				const size_t bytes = pair_pipe_lds<SP>() + kPairPipeFlagBytes;
				if (int p_rc = allow_lds_dev(sonce, bytes, row_pair_pipe_kernel<SP, KIND, false, true>)) return p_rc;
				hipLaunchKernelGGL((row_pair_pipe_kernel<SP, KIND, false, true>), dim3(cus), dim3(SP::T), bytes, (hipStream_t)stream, a, npairs);
				HIPCHK(hipGetLastError());
				return 0;
			}
		}
		if (on && is_plain(a) && npairs > cus) {
			// on DSP_PAIR_PIPE_T = 768 threads: the held output line, the line in flight and the closing phase's arithmetic need more than the 128
			// registers that 1024 threads leave each, and 768 fill the radix-16 / radix-15 stages' rounds better (720 and 768 butterflies); measured
			// 195.7 / 196.8 us against 221 / 220 on 1024 threads and 204.5 / 212 for row_pair_kernel (profiles/r06_8k_pair_pipe.txt).
			// Both kinds request the next pair's first line in FRONT of this pair's stores, which are unconditional so that the compiler counts them and the
			// wait at the next phase 0 is for the loads alone: REDFT01's twenty stores per thread are; REDFT10's sit under per-lane conditions (k > 0,
			// k != L/2, the ragged last round) and are made unconditional by RowSpecG::final_each's UNCOND form -- ROW10 198.0 us with the request in front of
			// conditional stores, 195.7 behind them, 185.1 in front of unconditional ones.
			typedef typename S::template with_threads<DSP_PAIR_PIPE_T> SP;
			constexpr bool EARLY = true;
			static DevOnce ponce;
			if (int p_rc = allow_lds_dev(ponce, pair_pipe_lds<SP>(), row_pair_pipe_kernel<SP, KIND, EARLY>)) return p_rc;
			hipLaunchKernelGGL((row_pair_pipe_kernel<SP, KIND, EARLY>), dim3(cus), dim3(SP::T), pair_pipe_lds<SP>(), (hipStream_t)stream, a, npairs);
			HIPCHK(hipGetLastError());
			return 0;
		}
	}
	if (is_plain(a)) hipLaunchKernelGGL((row_pair_kernel<S, KIND, true>), dim3(npairs), dim3(S::T), S::LDS, (hipStream_t)stream, a);
	else hipLaunchKernelGGL((row_pair_kernel<S, KIND, false>), dim3(npairs), dim3(S::T), S::LDS, (hipStream_t)stream, a);
	HIPCHK(hipGetLastError());
	return 0;
}
template <class S>
int launch_row_sum2(const typename S::PA &a, const typename S::PA &b, int nwork, void *stream)
{
	static DevOnce once;
	if (int lds_rc = allow_lds_dev(once, S::LDS, row_sum2_kernel<S>)) return lds_rc;
	hipLaunchKernelGGL((row_sum2_kernel<S>), dim3(nwork), dim3(S::T), S::LDS, (hipStream_t)stream, a, b);
	HIPCHK(hipGetLastError());
	return 0;
}
template <class S, int KIND>
int launch_col_half(const typename S::PA &a, int nwork, void *stream)
{
	static DevOnce once;
	if (int lds_rc = allow_lds_dev(once, S::LDS, col_half_kernel<S, KIND, false>, col_half_kernel<S, KIND, true>)) return lds_rc;
	if (is_plain(a)) hipLaunchKernelGGL((col_half_kernel<S, KIND, true>), dim3(nwork), dim3(S::T), S::LDS, (hipStream_t)stream, a);
	else hipLaunchKernelGGL((col_half_kernel<S, KIND, false>), dim3(nwork), dim3(S::T), S::LDS, (hipStream_t)stream, a);
	HIPCHK(hipGetLastError());
	return 0;
}
// nsrc = 1 (cw <= M/4), 2 (cw <= M/2) or 4 source pixels per slot pair; clip: vw < M
template <class S, int C>
int launch_zoomx(const ZoomXArgs &a, int nsrc, bool clip, void *stream)
{
	// waves per SIMD to ask of the register allocator: what the plane's LDS lets onto a CU, at most two (256 VGPRs: the held samples of two channels + a
	// radix-16 butterfly over Pk2 need 214-244)
	constexpr int WPE = (int)((160 * 1024) / S::LDS) * S::T / 256 >= 2 ? 2 : 1;
	static DevOnce once;
	if (int rc = allow_lds_dev(once, S::LDS, zoomx_lean_kernel<S, C, 1, WPE, false>, zoomx_lean_kernel<S, C, 1, WPE, true>, zoomx_lean_kernel<S, C, 2, WPE, false>,
	                           zoomx_lean_kernel<S, C, 2, WPE, true>, zoomx_lean_kernel<S, C, 4, WPE, false>, zoomx_lean_kernel<S, C, 4, WPE, true>)) return rc;
	const dim3 g(a.lines), b(S::T);
	hipStream_t st = (hipStream_t)stream;
	switch (nsrc * 2 + (clip ? 1 : 0)) {
	case 2: hipLaunchKernelGGL((zoomx_lean_kernel<S, C, 1, WPE, false>), g, b, S::LDS, st, a); break;
	case 3: hipLaunchKernelGGL((zoomx_lean_kernel<S, C, 1, WPE, true>), g, b, S::LDS, st, a); break;
	case 4: hipLaunchKernelGGL((zoomx_lean_kernel<S, C, 2, WPE, false>), g, b, S::LDS, st, a); break;
	case 5: hipLaunchKernelGGL((zoomx_lean_kernel<S, C, 2, WPE, true>), g, b, S::LDS, st, a); break;
	case 8: hipLaunchKernelGGL((zoomx_lean_kernel<S, C, 4, WPE, false>), g, b, S::LDS, st, a); break;
	case 9: hipLaunchKernelGGL((zoomx_lean_kernel<S, C, 4, WPE, true>), g, b, S::LDS, st, a); break;
	default: return -1;
	}
	HIPCHK(hipGetLastError());
	return 0;
}
template <class S, int KIND>
int launch_row_spec_u8(const typename S::PA &a, const U8IO &io, int nwork, void *stream)
{
	if (!is_plain(a)) return -5;                 // (the kernel is the plain instantiation only)
	static DevOnce once;
	if (int lds_rc = allow_lds_dev(once, S::LDS, row_spec_u8_kernel<S, KIND>)) return lds_rc;
	hipLaunchKernelGGL((row_spec_u8_kernel<S, KIND>), dim3(nwork), dim3(S::T), S::LDS, (hipStream_t)stream, a, io);
	HIPCHK(hipGetLastError());
	return 0;
}
template <class S>
int launch_col_roundtrip(const typename S::PA &af, const typename S::PA &ai, const MotionFilter &filt, unsigned long long *coded, int nwork, void *stream)
{
	if (!is_plain(af) || !is_plain(ai)) return -5;      // (col_roundtrip_body runs the plain instantiation of both passes)
	static DevOnce once;
	if (int lds_rc = allow_lds_dev(once, S::LDS, col_roundtrip_kernel<S>)) return lds_rc;
	FilterOp f; f.p = filt;
	hipLaunchKernelGGL((col_roundtrip_kernel<S>), dim3(nwork), dim3(S::T), S::LDS, (hipStream_t)stream, af, ai, f, coded);
	HIPCHK(hipGetLastError());
	return 0;
}

}  // namespace dspfft
