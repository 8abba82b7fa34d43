// dct_core.h -- the body of the generic (runtime-geometry) DCT-II / DCT-III axis passes, written as
// barrier-separated PHASES.  Each phase is a __host__ __device__ function of (tid, nthreads), so
//   * dct_kernels.hip runs   phase; __syncthreads(); phase; ...   inside one workgroup, and
//   * tests/emul (g++, CPU, test-only) runs  for(tid) phase; for(tid) phase; ...
// over exactly the same code.  The product never executes the CPU emulation.
//
// What is computed (FFTW 3.3 definitions, reached in the reference through fftw(plan_many_r2r) +
// fftw(execute): spec/spec.c:63-64, spec/ispec.c:165-166, zoom/zoom.c:263-264, scan/scan.c:292-293,359,447,
// motion/motion.c:535-552,641,753, applybasis/draw.c:74-75):
//   REDFT10: Y[k] = 2 sum_j X[j] cos(pi (j+1/2) k / N)        REDFT01: Y[k] = X[0] + 2 sum_{j>=1} X[j] cos(pi j (k+1/2) / N)
//
// Two pass shapes:
//   ROW  -- the transformed axis is (nearly) contiguous: a line is N*C consecutive floats holding C
//           interleaved signals (C = 3 for the image tools' HWC buffers, 1 for motion's planar rows).
//           N even; each real signal is packed into an N/2-point complex FFT (even/odd "Makhoul" order);
//           all C signals of the line are transformed together.
//   COL  -- the axis is strided and an inner contiguous dimension exists: a workgroup owns a tile of K
//           adjacent floats x all N rows; adjacent float columns are paired into one complex signal
//           (two real transforms per complex FFT of length N; any N).
// In both, the complex FFT runs in LDS, in place, decimation in frequency, mixed radix.
#pragma once
#include "radix.h"

// keeps the compiler from hoisting every load of a phase above the first use (register pressure)
#if defined(__HIP_DEVICE_COMPILE__)
#define DSP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// pins a float4 in registers HERE: without it the compiler sinks the computation of values that are
// consumed several barriers later and keeps their (twice as many) inputs alive instead
#define DSP_PIN4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
#define DSP_PIN1(v) asm volatile("" : "+v"(v))
#else
#define DSP_SCHED_FENCE() ((void)0)
#define DSP_PIN4(v) ((void)0)
#define DSP_PIN1(v) ((void)0)
#endif

namespace dspfft {

struct FastDiv {
	uint32_t d, mul;   // mul = ceil(2^32 / d) for d >= 2 (exact for n*d < 2^32)
	DSP_HD uint32_t div(uint32_t n) const {
		if (d == 1) return n;
#if defined(__HIP_DEVICE_COMPILE__)
		return __umulhi(n, mul);
#else
		return (uint32_t)(((uint64_t)n * mul) >> 32);
#endif
	}
	// exact for every n < 2^31 (div() overshoots by one when n * (mul * d - 2^32) reaches 2^32)
	DSP_HD uint32_t div_exact(uint32_t n) const
	{
		uint32_t q = div(n);
		if (q * d > n) q--;
		return q;
	}
};

struct StageDesc {
	int R;        // radix
	int Lc;       // length of the sub-transforms this stage splits
	int M1;       // Lc / R
	int twstep;   // L / Lc : w_Lc^e = W[e * twstep]
	FastDiv divM1;
};

struct FftDesc {
	int L;        // complex length
	int ns;       // number of stages
	StageDesc st[8];
};

enum { KIND_REDFT10 = 0, KIND_REDFT01 = 1 };

// everything about a pass that does not depend on the sample type
struct PassGeom {
	int N;            // real transform length
	int kind;         // KIND_*
	// ROW
	int C;            // interleaved signals per line
	int LPW;          // lines per workgroup (generic ROW kernel; the specialised kernels take one)
	long long nlines; // ROW: lines of the launch (nb0 * nb1 * the third level's extent)
	FastDiv divC, divNC, divKC;   // ROW: divide by C, by N*C (samples per line), by (N/4+1)*C (output pairs per line)
	// COL
	int K;            // tile width in samples (even)
	int B;            // complex columns per tile = K/2
	int ninner;       // extent of the inner contiguous dimension
	int ntiles;       // ceil(ninner / K)
	long long es_in, es_out;   // stride (elements) between consecutive samples of the axis (COL)
	// batch dimensions (two levels) -- line/tile base = i0*sb0 + i1*sb1
	int nb0, nb1;
	long long sb0_in, sb1_in, sb0_out, sb1_out;
	long long sb2_in, sb2_out;   // ROW passes: a third batch level (lines beyond nb0 * nb1 step by these; 0, 0 = none).  Round 4: the slab transform's
	                             // x pass writes the blocks of ALL destination ranks in one launch (dist.py SlabDCT3D)
	const uint32_t *pos;  // pos[k] = LDS slot holding FFT output k after the DIF stages
	// fused scan step (scan/scan.c:429-459): the FIRST pass zeroes every input element whose owner
	// id differs (mask[offset / mask_div] != mask_id); the LAST pass adds into `out` instead of storing
	const uint32_t *mask;
	uint32_t mask_id;
	FastDiv mask_div;     // elements per owner id (32-bit offsets: masked runs are limited to 2^32 / d elements)
	int mask_mode;        // column tiles (dct_spec.h, masked_two_step): 0 = owner ids from `mask`, item by item; 1 / 2 = from `eids`, one / two bytes per id
	const void *eids;     // dspfft_plan_scan_prepare: the owner id of every element, in column-tile order (eid_index); needs zpage
	int accumulate;
	// sparse scan frames (specialised kernels, dct_spec.h): a masked COL first pass records per tile whether any coefficient was
	// selected (zflags[tile] = 0 / 1) and skips the tiles with none; the ROW pass that follows reads zeros for those tiles, not `in`
	uint8_t *zflags;      // null: off.  COL side writes (when mask is set), ROW side reads
	int zshift;           // ROW side: log2 of the column pass's tile width in samples
	int zhalf;            // ROW side: the flags of odd lines start at zflags + zhalf (split column passes: half 1), else 0
	const void *zpage;    // ROW side: 64 zero bytes to load from in place of a skipped tile (also what rows outside win_* load)
	int alt_out;          // specialised COL / ROW REDFT01 last pass: output sample y of the axis is multiplied by (-1)^y (dspfft_plan_set_output_alternate: the sine
	                      // part of zoom's shifted cosine series is (-1)^b REDFT01 of the reversed input); 0: off
	int win_lo, win_hi;   // specialised COL / ROW REDFT01 first pass: input samples of the axis outside [win_lo, win_hi) are zero BY CONTRACT and are not read
	                      // (dspfft_plan_set_input_window: zoom's y stage transforms a spectrum zero-padded to 4x its length); 0, 0: off
	const void *in_mul;   // specialised ROW / COL REDFT01 first pass (dspfft_plan_set_input_modulation): input sample x of the axis is read from position
	int in_rev;           // p = in_rev > 0 ? in_rev - x : x of its line and multiplied by in_mul[p] (a table of the plan's sample type); null / 0: off
	int lean_off;         // specialised COL REDFT01 passes: 1 keeps the natural-order write + closing phase (DSPFFT_LEAN01=0, A/B runs); 0: the last stage stores
	const uint32_t *zranges;   // COL side, optional: (min, max) owner id of every tile (dspfft_plan_scan_prepare): a tile whose range
	                           // excludes mask_id is skipped without reading its owner ids
	FftDesc fft;
	FastDiv divB;         // divide by the number of signals in the LDS buffer (ROW: C*LPW, COL: B)
};

template <class R>
struct PassArgsT : PassGeom {
	const R *in;
	R *out;
	// tables (device memory)
	const cx<R> *T;       // T[j] = exp(-i pi j / (2N)), j in [0, N]
	const cx<R> *W;       // W[t] = exp(-2 pi i t / L),  t in [0, L)
	const cx<R> *H;       // half-tile column passes (dct_spec.h ColHalfSpec) only: H[n] = exp(-2 pi i n / N), n in [0, N/2)
	R scale;              // every output is multiplied by scale ...
	R out_scale0;         // ... and output index 0 of this axis additionally by out_scale0
	R in_scale0;          // input index 0 of this axis is multiplied by in_scale0 before transforming
};
typedef PassArgsT<float> PassArgs;      // the tuned single-precision path (and dct_spec.h)
typedef PassArgsT<double> PassArgsD;    // the fftw_ (double) API: generic kernels only

// 16-byte and two-sample vector types of each sample type
template <class R> struct vec_of;
template <> struct vec_of<float> { typedef float4 v16; typedef float2 v2; };
template <> struct vec_of<double> { typedef double2 v16; typedef double2 v2; };

template <class R>
DSP_HD R masked(const PassArgsT<R> &a, long long off, R v)
{
	if (!a.mask) return v;
	return a.mask[a.mask_div.div((uint32_t)off)] != a.mask_id ? R(0) : v;
}
// a.in[off] under the mask, without fetching elements that are masked out.  MASKED is chosen once per phase (a.mask
// is uniform): a test around every load would make each load wait for its data before the next one is issued
template <bool MASKED, class R>
DSP_HD R load_masked(const PassArgsT<R> &a, long long off)
{
	if constexpr (MASKED) { if (a.mask[a.mask_div.div((uint32_t)off)] != a.mask_id) return R(0); }
	return a.in[off];
}

// ------------------------------------------------------------------------------------------------
// FFT stage (in place in LDS).  Signals are interleaved: element n of signal s is buf[n*B + s].
template <int R, class C>
DSP_HD void fft_stage_r(C *buf, int L, const StageDesc &S, int B, FastDiv divB, const C *W, int tid, int nthr)
{
	const int nitems = (L / R) * B;
	const int stride = S.M1 * B;
	for (int it = tid; it < nitems; it += nthr) {
		const int q = (int)divB.div((uint32_t)it), s = it - q * B;
		const int blk = (int)S.divM1.div((uint32_t)q), m = q - blk * S.M1;
		const int base = (blk * S.Lc + m) * B + s;
		C x[R];
		static_for<0, R>([&](auto r) { x[r] = buf[base + r * stride]; });
		Dft<R>::run(x);
		if (S.M1 > 1) {
			// w^r for r = 2..R-1 from the one table value w = W[m * twstep] by a product tree (depth log2 R):
			// one table load per butterfly instead of R-1
			C w[R > 1 ? R : 2];
			w[1] = W[m * S.twstep];
			static_for<2, R>([&](auto r) { w[r] = cmul(w[r / 2], w[r - r / 2]); });
			static_for<1, R>([&](auto r) { x[r] = cmul(x[r], w[r]); });
		}
		static_for<0, R>([&](auto r) { buf[base + r * stride] = x[r]; });
	}
}

template <class C>
DSP_HD void fft_stage(C *buf, int L, const StageDesc &S, int B, FastDiv divB, const C *W, int tid, int nthr)
{
	switch (S.R) {
	case 2: fft_stage_r<2>(buf, L, S, B, divB, W, tid, nthr); break;
	case 3: fft_stage_r<3>(buf, L, S, B, divB, W, tid, nthr); break;
	case 4: fft_stage_r<4>(buf, L, S, B, divB, W, tid, nthr); break;
	case 5: fft_stage_r<5>(buf, L, S, B, divB, W, tid, nthr); break;
	case 6: fft_stage_r<6>(buf, L, S, B, divB, W, tid, nthr); break;
	case 7: fft_stage_r<7>(buf, L, S, B, divB, W, tid, nthr); break;
	case 8: fft_stage_r<8>(buf, L, S, B, divB, W, tid, nthr); break;
	case 9: fft_stage_r<9>(buf, L, S, B, divB, W, tid, nthr); break;
	case 10: fft_stage_r<10>(buf, L, S, B, divB, W, tid, nthr); break;
	case 11: fft_stage_r<11>(buf, L, S, B, divB, W, tid, nthr); break;
	case 12: fft_stage_r<12>(buf, L, S, B, divB, W, tid, nthr); break;
	case 13: fft_stage_r<13>(buf, L, S, B, divB, W, tid, nthr); break;
	case 15: fft_stage_r<15>(buf, L, S, B, divB, W, tid, nthr); break;
	case 16: fft_stage_r<16>(buf, L, S, B, divB, W, tid, nthr); break;
	default: break;
	}
}

// Exact inverse of fft_stage (up to the factor R): undo the twiddles, then the conjugate butterfly.  Running the
// stages of a transform LAST TO FIRST with this function takes the digit-reversed spectrum back to natural-order
// samples (times L), in place, with no reordering pass -- used by the Bluestein convolution below.
template <int R, class C>
DSP_HD void fft_stage_inv_r(C *buf, int L, const StageDesc &S, int B, FastDiv divB, const C *W, int tid, int nthr)
{
	const int nitems = (L / R) * B;
	const int stride = S.M1 * B;
	for (int it = tid; it < nitems; it += nthr) {
		const int q = (int)divB.div((uint32_t)it), s = it - q * B;
		const int blk = (int)S.divM1.div((uint32_t)q), m = q - blk * S.M1;
		const int base = (blk * S.Lc + m) * B + s;
		C x[R];
		static_for<0, R>([&](auto r) { x[r] = buf[base + r * stride]; });
		if (S.M1 > 1) {
			C w[R > 1 ? R : 2];
			w[1] = W[m * S.twstep];
			static_for<2, R>([&](auto r) { w[r] = cmul(w[r / 2], w[r - r / 2]); });
			static_for<1, R>([&](auto r) { x[r] = cmulc(x[r], w[r]); });
		}
		static_for<0, R>([&](auto r) { x[r] = cconj(x[r]); });
		Dft<R>::run(x);
		static_for<0, R>([&](auto r) { buf[base + r * stride] = cconj(x[r]); });
	}
}

template <class C>
DSP_HD void fft_stage_inv(C *buf, int L, const StageDesc &S, int B, FastDiv divB, const C *W, int tid, int nthr)
{
	switch (S.R) {
	case 2: fft_stage_inv_r<2>(buf, L, S, B, divB, W, tid, nthr); break;
	case 3: fft_stage_inv_r<3>(buf, L, S, B, divB, W, tid, nthr); break;
	case 4: fft_stage_inv_r<4>(buf, L, S, B, divB, W, tid, nthr); break;
	case 5: fft_stage_inv_r<5>(buf, L, S, B, divB, W, tid, nthr); break;
	case 6: fft_stage_inv_r<6>(buf, L, S, B, divB, W, tid, nthr); break;
	case 7: fft_stage_inv_r<7>(buf, L, S, B, divB, W, tid, nthr); break;
	case 8: fft_stage_inv_r<8>(buf, L, S, B, divB, W, tid, nthr); break;
	case 9: fft_stage_inv_r<9>(buf, L, S, B, divB, W, tid, nthr); break;
	case 10: fft_stage_inv_r<10>(buf, L, S, B, divB, W, tid, nthr); break;
	case 12: fft_stage_inv_r<12>(buf, L, S, B, divB, W, tid, nthr); break;
	case 15: fft_stage_inv_r<15>(buf, L, S, B, divB, W, tid, nthr); break;
	case 16: fft_stage_inv_r<16>(buf, L, S, B, divB, W, tid, nthr); break;
	default: break;       // 11 and 13 never occur: the convolution length is 7-smooth
	}
}

// ------------------------------------------------------------------------------------------------
// ROW pass.  A workgroup owns LPW consecutive lines (1 for long lines; several short ones so that it still has a few
// thousand samples).  LDS: buf[L*B] complex, L = N/2, B = C*LPW signals, channel s of line l of packed sample m at
// buf[m*B + l*C + s]; bases[2l], bases[2l+1] = input / output element offset of line l (filled by row_bases).  The
// lines go from global memory straight into the packed FFT input and from the FFT output straight back (no staging
// copy), so all loads of a line happen before its first store: in place is safe.
// index of sample v[n] of the even/odd-reordered signal inside the original signal
DSP_HD int makhoul_src(int n, int N) { return 2 * n < N ? 2 * n : 2 * (N - 1 - n) + 1; }
DSP_HD int makhoul_dst(int y, int N) { return (y & 1) ? N - 1 - (y >> 1) : (y >> 1); }

DSP_HD void row_base(const PassGeom &a, int line, long long &bin, long long &bout)
{
	int i1 = line / a.nb0;
	const int i0 = line - i1 * a.nb0;
	bin = i0 * a.sb0_in; bout = i0 * a.sb0_out;
	if (a.sb2_in | a.sb2_out) {                   // (uniform: only plans with a third level pay the second division)
		const int i2 = i1 / a.nb1;
		i1 -= i2 * a.nb1;
		bin += i2 * a.sb2_in; bout += i2 * a.sb2_out;
	}
	bin += i1 * a.sb1_in; bout += i1 * a.sb1_out;
}
// lines of workgroup `wg`: returns how many (the last workgroup may hold fewer than LPW) and fills bases[]
DSP_HD int row_bases(const PassGeom &a, int wg, long long *bases, int tid)
{
	const long long nlines = a.nlines;
	long long cnt = nlines - (long long)wg * a.LPW;
	if (cnt > a.LPW) cnt = a.LPW;
	if (tid < (int)cnt) row_base(a, wg * a.LPW + tid, bases[2 * tid], bases[2 * tid + 1]);
	return (int)cnt;
}

template <class R>
DSP_HD void row_put(const PassArgsT<R> &a, long long off, R v)
{
	if (a.accumulate) a.out[off] += v; else a.out[off] = v;
}

// REDFT10: pixels in memory order -> even/odd reordered, packed two reals per complex slot
template <bool MASKED, class R>
DSP_HD void row_load10_m(const PassArgsT<R> &a, cx<R> *buf, const long long *bases, int cnt, int tid, int nthr)
{
	const int N = a.N, C = a.C, B = a.C * a.LPW;
	R *bf = reinterpret_cast<R *>(buf);
	for (int it = tid; it < cnt * N * C; it += nthr) {
		const int l = (int)a.divNC.div((uint32_t)it), e = it - l * N * C;
		const int x = (int)a.divC.div((uint32_t)e), s = e - x * C;
		R v = load_masked<MASKED>(a, bases[2 * l] + e);
		if (x == 0) v *= a.in_scale0;
		const int n = makhoul_dst(x, N);
		bf[2 * ((n >> 1) * B + l * C + s) + (n & 1)] = v;
	}
}

template <class R>
DSP_HD void row_load10(const PassArgsT<R> &a, cx<R> *buf, const long long *bases, int cnt, int tid, int nthr)
{
	if (a.mask) row_load10_m<true>(a, buf, bases, cnt, tid, nthr); else row_load10_m<false>(a, buf, bases, cnt, tid, nthr);
}

// REDFT10: FFT output -> 4 real outputs per (k, L-k) pair, stored
template <class R>
DSP_HD void row_post10(const PassArgsT<R> &a, const cx<R> *buf, const long long *bases, int cnt, int tid, int nthr)
{
	typedef cx<R> C_;
	const int L = a.N / 2, C = a.C, N = a.N, B = a.C * a.LPW;
	const int nk = L / 2 + 1;
	for (int it = tid; it < cnt * nk * C; it += nthr) {
		const int l = (int)a.divKC.div((uint32_t)it), e = it - l * nk * C;
		const int k = (int)a.divC.div((uint32_t)e), s = e - k * C;
		const int km = k ? L - k : 0;
		const C_ zk = buf[a.pos[k] * B + l * C + s];
		const C_ zm = cconj(buf[a.pos[km] * B + l * C + s]);
		const C_ E = cscale(cadd(zk, zm), R(0.5));
		const C_ Dh = cscale(csub(zk, zm), R(0.5));
		const C_ D = cmul_mi(Dh);                 // (zk - conj zm) / (2i)
		// one table value per item: T[L-k] = e^{-i pi/4} conj(T[k]) and T[4k] = T[k]^4
		const C_ tk = a.T[k];
		const C_ tlk = cmul(cconj(tk), cmk<R>(R(0.70710678118654752440), R(-0.70710678118654752440)));
		const C_ t2 = cmk<R>(tk.x * tk.x - tk.y * tk.y, R(2) * tk.x * tk.y);
		const C_ t4 = cmk<R>(t2.x * t2.x - t2.y * t2.y, R(2) * t2.x * t2.y);
		const C_ P = cmul(t4, D);                 // exp(-2 pi i k / N) * D
		const C_ Vk = cadd(E, P);
		const C_ Vm = cconj(csub(E, P));          // V[L-k]
		const C_ wk = cmul(tk, Vk);
		const C_ wm = cmul(tlk, Vm);
		const long long o = bases[2 * l + 1] + s;
		const R sc = a.scale;
		row_put(a, o + (long long)k * C, R(2) * wk.x * sc * (k == 0 ? a.out_scale0 : R(1)));
		if (k > 0) row_put(a, o + (long long)(N - k) * C, R(-2) * wk.y * sc);
		if (L - k != k) row_put(a, o + (long long)(L - k) * C, R(2) * wm.x * sc);
		if (k > 0 && L + k != N - k) row_put(a, o + (long long)(L + k) * C, R(-2) * wm.y * sc);
	}
}

// REDFT01: natural-order input -> conj of the half-length spectrum in buf
template <bool MASKED, class R>
DSP_HD void row_load01_m(const PassArgsT<R> &a, cx<R> *buf, const long long *bases, int cnt, int tid, int nthr)
{
	typedef cx<R> C_;
	const int L = a.N / 2, C = a.C, N = a.N, B = a.C * a.LPW;
	const int nk = L / 2 + 1;
	for (int it = tid; it < cnt * nk * C; it += nthr) {
		const int l = (int)a.divKC.div((uint32_t)it), e = it - l * nk * C;
		const int k = (int)a.divC.div((uint32_t)e), s = e - k * C;
		const long long o = bases[2 * l] + s;
		auto ld = [&](int px) { return load_masked<MASKED>(a, o + (long long)px * C); };
		const R xk = ld(k) * (k == 0 ? a.in_scale0 : R(1));
		const R xnk = k ? ld(N - k) : R(0);
		const R xlk = ld(L - k);
		const R xlpk = ld(L + k);                              // k <= L/2 so L+k <= N-1
		const C_ tk = a.T[k];
		const C_ tlk = cmul(cconj(tk), cmk<R>(R(0.70710678118654752440), R(-0.70710678118654752440)));   // T[L-k]
		const C_ t2 = cmk<R>(tk.x * tk.x - tk.y * tk.y, R(2) * tk.x * tk.y);
		const C_ t4 = cmk<R>(t2.x * t2.x - t2.y * t2.y, R(2) * t2.x * t2.y);                              // T[4k]
		const C_ Vk = cmulc(cmk<R>(xk, -xnk), tk);             // conj(T[k]) * (X[k] - i X[N-k])
		const C_ Vm = cmulc(cmk<R>(xlk, -xlpk), tlk);          // V[L-k]
		const C_ S = cadd(Vk, cconj(Vm));
		const C_ D = csub(Vk, cconj(Vm));
		const C_ Q = cmul_pi(cmulc(D, t4));                    // i * conj(t1[k]) * D
		buf[k * B + l * C + s] = cconj(cadd(S, Q));
		if (k > 0) buf[(L - k) * B + l * C + s] = csub(S, Q);
	}
}

template <class R>
DSP_HD void row_load01(const PassArgsT<R> &a, cx<R> *buf, const long long *bases, int cnt, int tid, int nthr)
{
	if (a.mask) row_load01_m<true>(a, buf, bases, cnt, tid, nthr); else row_load01_m<false>(a, buf, bases, cnt, tid, nthr);
}

// REDFT01: FFT output -> time samples in memory order
template <class R>
DSP_HD void row_store01(const PassArgsT<R> &a, const cx<R> *buf, const long long *bases, int cnt, int tid, int nthr)
{
	const int N = a.N, C = a.C, B = a.C * a.LPW;
	const R *bf = reinterpret_cast<const R *>(buf);
	for (int it = tid; it < cnt * N * C; it += nthr) {
		const int l = (int)a.divNC.div((uint32_t)it), e = it - l * N * C;
		const int x = (int)a.divC.div((uint32_t)e), s = e - x * C;
		const int n = makhoul_dst(x, N);
		const R f = bf[2 * (a.pos[n >> 1] * B + l * C + s) + (n & 1)];
		const R sc = (x == 0) ? a.scale * a.out_scale0 : a.scale;
		row_put(a, bases[2 * l + 1] + e, ((n & 1) ? -f : f) * sc);
	}
}

// ------------------------------------------------------------------------------------------------
// COL pass.  LDS: buf[N*B] complex.  Tile t of batch (i0,i1) covers floats [t*K, t*K+K) of the inner dim.
DSP_HD void col_base(const PassGeom &a, int wg, long long &bin, long long &bout, int &valid)
{
	const int bt = wg / a.ntiles, t = wg - bt * a.ntiles;
	const int i1 = bt / a.nb0, i0 = bt - i1 * a.nb0;
	bin = i0 * a.sb0_in + i1 * a.sb1_in + (long long)t * a.K;
	bout = i0 * a.sb0_out + i1 * a.sb1_out + (long long)t * a.K;
	valid = a.ninner - t * a.K;   // number of valid float columns in this tile (may exceed K)
	if (valid > a.K) valid = a.K;
}

template <class R>
DSP_HD cx<R> ld2(const R *p, bool vec, int nvalid)
{
	if (nvalid >= 2) {
		if (vec) { const typename vec_of<R>::v2 v = *reinterpret_cast<const typename vec_of<R>::v2 *>(p); return cmk<R>(v.x, v.y); }
		return cmk<R>(p[0], p[1]);
	}
	return cmk<R>(nvalid >= 1 ? p[0] : R(0), R(0));
}
template <class R>
DSP_HD cx<R> ld2m(const PassArgsT<R> &a, long long off, bool vec, int nvalid)
{
	cx<R> v = ld2(a.in + off, vec, nvalid);
	if (a.mask) { v.x = masked(a, off, v.x); if (nvalid >= 2) v.y = masked(a, off + 1, v.y); }
	return v;
}
template <class R>
DSP_HD void st2(R *p, bool vec, int nvalid, R a, typename same_t<R>::type b)
{
	if (nvalid >= 2) {
		if (vec) { typename vec_of<R>::v2 v; v.x = a; v.y = b; *reinterpret_cast<typename vec_of<R>::v2 *>(p) = v; }
		else { p[0] = a; p[1] = b; }
	} else if (nvalid >= 1) p[0] = a;
}
template <class R>
DSP_HD void st2a(const PassArgsT<R> &a, R *p, bool vec, int nvalid, R x, typename same_t<R>::type y)
{
	if (a.accumulate) { if (nvalid >= 1) p[0] += x; if (nvalid >= 2) p[1] += y; }
	else st2(p, vec, nvalid, x, y);
}
// two-sample vector access: both strides even and the base pointer aligned to two samples
template <class R>
DSP_HD bool vec2_ok(long long base, long long es, const R *p) { return (((base | es) & 1) == 0) && ((((uintptr_t)p) & (2 * sizeof(R) - 1)) == 0); }

// REDFT10: load tile rows, even/odd reorder along the axis
template <class R>
DSP_HD void col_load2(const PassArgsT<R> &a, cx<R> *buf, long long bin, int valid, int tid, int nthr)
{
	const int N = a.N, B = a.B;
	const bool vec = vec2_ok(bin, a.es_in, a.in);
	for (int it = tid; it < N * B; it += nthr) {
		const int y = (int)a.divB.div((uint32_t)it), j = it - y * B;
		cx<R> v = ld2m(a, bin + (long long)y * a.es_in + 2 * j, vec, valid - 2 * j);
		if (y == 0) { v.x *= a.in_scale0; v.y *= a.in_scale0; }
		buf[makhoul_dst(y, N) * B + j] = v;
	}
}

// REDFT10: separate the two real transforms, quarter-sample twiddle, store rows k and N-k
template <class R>
DSP_HD void col_post2(const PassArgsT<R> &a, const cx<R> *buf, long long bout, int valid, int tid, int nthr)
{
	typedef cx<R> C_;
	const int N = a.N, B = a.B;
	const int nk = N / 2 + 1;
	const bool vec = vec2_ok(bout, a.es_out, a.out);
	for (int it = tid; it < nk * B; it += nthr) {
		const int k = (int)a.divB.div((uint32_t)it), j = it - k * B;
		const int km = k ? N - k : 0;
		const C_ zk = buf[a.pos[k] * B + j];
		const C_ zm = cconj(buf[a.pos[km] * B + j]);
		const C_ A2 = cadd(zk, zm);                   // 2 A[k]
		const C_ B2 = cmul_mi(csub(zk, zm));          // 2 B[k]
		const C_ t = a.T[k];
		const C_ wa = cmul(t, A2), wb = cmul(t, B2);
		const R sc = a.scale;
		R *o = a.out + bout + 2 * j;
		const R s0 = (k == 0) ? sc * a.out_scale0 : sc;
		st2a(a, o + (long long)k * a.es_out, vec, valid - 2 * j, wa.x * s0, wb.x * s0);
		if (k > 0 && km != k) st2a(a, o + (long long)km * a.es_out, vec, valid - 2 * j, -wa.y * sc, -wb.y * sc);
	}
}

// REDFT01: rows k and N-k -> conj spectrum of (a + i b)
template <class R>
DSP_HD void col_pre3(const PassArgsT<R> &a, cx<R> *buf, long long bin, int valid, int tid, int nthr)
{
	typedef cx<R> C_;
	const int N = a.N, B = a.B;
	const int nk = N / 2 + 1;
	const bool vec = vec2_ok(bin, a.es_in, a.in);
	for (int it = tid; it < nk * B; it += nthr) {
		const int k = (int)a.divB.div((uint32_t)it), j = it - k * B;
		const int km = k ? N - k : 0;
		const long long p = bin + 2 * j;
		C_ xk = ld2m(a, p + (long long)k * a.es_in, vec, valid - 2 * j);
		C_ xm = k ? ld2m(a, p + (long long)km * a.es_in, vec, valid - 2 * j) : cmk<R>(R(0), R(0));
		if (k == 0) { xk.x *= a.in_scale0; xk.y *= a.in_scale0; }
		const C_ t = a.T[k];
		const C_ Va = cmulc(cmk<R>(xk.x, -xm.x), t);     // conj(T[k]) (Xa[k] - i Xa[N-k])
		const C_ Vb = cmulc(cmk<R>(xk.y, -xm.y), t);
		buf[k * B + j] = cmk<R>(Va.x - Vb.y, -Va.y - Vb.x);            // conj(Va) - i conj(Vb)
		if (k > 0) buf[km * B + j] = cmk<R>(Va.x + Vb.y, Va.y - Vb.x); // Va - i Vb
	}
}

// REDFT01: FFT output -> both real signals, un-reordered, store
template <class R>
DSP_HD void col_unpack3(const PassArgsT<R> &a, const cx<R> *buf, long long bout, int valid, int tid, int nthr)
{
	const int N = a.N, B = a.B;
	const bool vec = vec2_ok(bout, a.es_out, a.out);
	for (int it = tid; it < N * B; it += nthr) {
		const int n = (int)a.divB.div((uint32_t)it), j = it - n * B;
		const cx<R> F = buf[a.pos[n] * B + j];
		const int y = makhoul_src(n, N);
		const R sc = (y == 0) ? a.scale * a.out_scale0 : a.scale;
		st2a(a, a.out + bout + (long long)y * a.es_out + 2 * j, vec, valid - 2 * j, F.x * sc, -F.y * sc);
	}
}

// ------------------------------------------------------------------------------------------------
// BLUESTEIN variant of the COL pass: lengths with a prime factor > 13.  The tile's N-point DFT (the only part of
// the COL pass that needs N to factor) becomes a length-M circular convolution, M >= 2N-1 and 7-smooth:
//   V[k] = conj(c[k]) * sum_n (v[n] conj(c[n])) c[k-n],   c[n] = exp(+i pi n^2 / N)
// computed in place as  conj(c) . IFFT_M( FFT_M(v conj(c), zero-padded) . Bhat ):  the forward stages leave the
// spectrum digit-reversed, Bhat = FFT_M(c wrapped to M) / M is stored in that same order (host, exact phase
// reduction n^2 mod 2N), and the inverse stages (fft_stage_inv, last to first) return natural-order samples.
// Loading, the two-for-one split, the quarter-sample twiddles, masks, scales and stores are the COL pass's own
// functions; `pos` is the identity because the DFT result lies in natural order.  LDS: M*B complex.
template <class R>
struct BlueArgsT : PassArgsT<R> {
	int M;                    // convolution length
	FftDesc fftM;
	const cx<R> *WM;          // exp(-2 pi i t / M)
	const cx<R> *chirp;       // c[n], n in [0, N)
	const cx<R> *Bhat;        // Bhat[p] = FFT_M(b)[k] / M at the LDS slot p that holds spectrum index k
};
typedef BlueArgsT<float> BlueArgs;
typedef BlueArgsT<double> BlueArgsD;

// A[n] *= conj(c[n]) for n < N; zero rows [N, M)
template <class R>
DSP_HD void blue_chirp_in(const BlueArgsT<R> &a, cx<R> *A, int tid, int nthr)
{
	const int B = a.B;
	for (int it = tid; it < a.M * B; it += nthr) {
		const int n = (int)a.divB.div((uint32_t)it);
		A[it] = n < a.N ? cmulc(A[it], a.chirp[n]) : cmk<R>(R(0), R(0));
	}
}
// pointwise product with the filter's spectrum (both digit-reversed)
template <class R>
DSP_HD void blue_mul(const BlueArgsT<R> &a, cx<R> *A, int tid, int nthr)
{
	const int B = a.B;
	for (int it = tid; it < a.M * B; it += nthr) {
		const int p = (int)a.divB.div((uint32_t)it);
		A[it] = cmul(A[it], a.Bhat[p]);
	}
}
// A[k] *= conj(c[k]) for k < N
template <class R>
DSP_HD void blue_chirp_out(const BlueArgsT<R> &a, cx<R> *A, int tid, int nthr)
{
	const int B = a.B;
	for (int it = tid; it < a.N * B; it += nthr) {
		const int k = (int)a.divB.div((uint32_t)it);
		A[it] = cmulc(A[it], a.chirp[k]);
	}
}

// ------------------------------------------------------------------------------------------------
// TINY pass: lengths up to 32 (motion's small blocks, e.g. -b 8x8x8: motion/README.md "3-dimensional analog to
// JPEG-style compression").  A whole line lives in one thread's registers and is transformed by the definition with
// compile-time cosines; no LDS, no barriers.  Lines are enumerated over up to six batch dimensions ordered by stride,
// so neighbouring threads touch neighbouring memory.
struct TinyGeom {
	int N, kind;
	long long es_in, es_out;
	int nd;                                   // batch dimensions in use
	int bn[6];
	long long bis[6], bos[6];
	FastDiv bdiv[6];
	long long nlines;
	int packed;                               // lines lie back to back (dimension 0 has stride N): chunked through LDS
	FastDiv chunk_div;                        // chunks per run of dimension 0
	const uint32_t *mask;
	uint32_t mask_id;
	FastDiv mask_div;
	int accumulate;
};
template <class R>
struct TinyArgsT : TinyGeom {
	const R *in;
	R *out;
	R scale, out_scale0, in_scale0;
};
typedef TinyArgsT<float> TinyArgs;
typedef TinyArgsT<double> TinyArgsD;

// cos(pi t / (2N)) for integer t, by the octant-reduced compile-time evaluation of radix.h
template <int N>
struct TinyCosTab {
	double c[4 * N];
	constexpr TinyCosTab() : c{} { for (int t = 0; t < 4 * N; t++) c[t] = ct::cossin(t, 4L * N).c; }
};


// y = unscaled transform of x.  Even lengths split once per factor of two into a half-length transform of the same kind and a
// half-length odd part done by its definition (REDFT10: sums / differences of mirrored inputs feed the even / odd outputs; REDFT01:
// the transpose): 8 points cost 24 multiply-adds and 12 additions instead of 64 multiply-adds, 16 points 88 + 28 instead of 256 --
// what the one-pass block kernels (block_core.h) are bound by.  Plain loops, fully unrolled: after unrolling every index is a
// constant, so x[] and y[] live in registers and the cosines fold into literals (a static_for form of this cost minutes of compile
// time per length).
template <int N, int KIND, class R>
DSP_HD void tiny_core(const R *x, R *y)
{
	constexpr TinyCosTab<N> tab = TinyCosTab<N>();
	if constexpr (N >= 4 && N % 2 == 0) {
		constexpr int H = N / 2;
		if constexpr (KIND == KIND_REDFT10) {
			R s[H], d[H], e[H];
#pragma unroll
			for (int j = 0; j < H; j++) { s[j] = x[j] + x[N - 1 - j]; d[j] = x[j] - x[N - 1 - j]; }
			tiny_core<H, KIND, R>(s, e);
#pragma unroll
			for (int m = 0; m < H; m++) {
				y[2 * m] = e[m];
				R acc = R(0);
#pragma unroll
				for (int j = 0; j < H; j++) acc += d[j] * (R)(2.0 * tab.c[((2 * j + 1) * (2 * m + 1)) % (4 * N)]);
				y[2 * m + 1] = acc;
			}
		} else {
			R xe[H], e[H];
#pragma unroll
			for (int m = 0; m < H; m++) xe[m] = x[2 * m];
			tiny_core<H, KIND, R>(xe, e);
#pragma unroll
			for (int k = 0; k < H; k++) {
				R o = R(0);
#pragma unroll
				for (int m = 0; m < H; m++) o += x[2 * m + 1] * (R)(2.0 * tab.c[((2 * m + 1) * (2 * k + 1)) % (4 * N)]);
				y[k] = e[k] + o; y[N - 1 - k] = e[k] - o;
			}
		}
	} else {
#pragma unroll
		for (int k = 0; k < N; k++) {
			R acc = (KIND == KIND_REDFT10) ? R(0) : x[0];
#pragma unroll
			for (int j = (KIND == KIND_REDFT10 ? 0 : 1); j < N; j++) {
				const int t = (KIND == KIND_REDFT10) ? ((2 * j + 1) * k) % (4 * N) : (j * (2 * k + 1)) % (4 * N);
				acc += x[j] * (R)(2.0 * tab.c[t]);
			}
			y[k] = acc;
		}
	}
}
template <int N, int KIND, class R>
DSP_HD void tiny_dct(const TinyArgsT<R> &a, R *x, R *y)
{
	x[0] *= a.in_scale0;
	tiny_core<N, KIND, R>(x, y);
#pragma unroll
	for (int k = 0; k < N; k++) y[k] *= a.scale * (k == 0 ? a.out_scale0 : R(1));
}

// base offsets of batch element `idx` over dimensions [d0, nd)
template <class R>
DSP_HD void tiny_base(const TinyArgsT<R> &a, uint32_t idx, int d0, long long &bin, long long &bout)
{
	bin = 0; bout = 0;
	for (int d = d0; d < a.nd; d++) {
		const uint32_t q = a.bdiv[d].div_exact(idx), i = idx - q * (uint32_t)a.bn[d];
		bin += (long long)i * a.bis[d]; bout += (long long)i * a.bos[d];
		idx = q;
	}
}

// strided lines: thread = line; neighbouring threads are neighbouring lines of the smallest-stride batch dimension
template <int N, int KIND, class R>
DSP_HD void tiny_line(const TinyArgsT<R> &a, long long line)
{
	long long bin, bout;
	tiny_base(a, (uint32_t)line, 0, bin, bout);      // nlines < 2^31 (planner)
	R x[N], y[N];
#pragma unroll
	for (int j = 0; j < N; j++) {
		const long long off = bin + (long long)j * a.es_in;
		const bool drop = a.mask && a.mask[a.mask_div.div((uint32_t)off)] != a.mask_id;
		x[j] = drop ? R(0) : a.in[off];
	}
	tiny_dct<N, KIND>(a, x, y);
#pragma unroll
	for (int k = 0; k < N; k++) {
		const long long off = bout + (long long)k * a.es_out;
		if (a.accumulate) a.out[off] += y[k]; else a.out[off] = y[k];
	}
}

// contiguous lines packed back to back (batch dimension 0 has stride N, e.g. the x axis of motion's blocks): a
// workgroup moves a chunk of TINY_CHUNK lines = TINY_CHUNK*N consecutive samples through LDS so that global accesses
// are element-contiguous across the wave; line pitch N|1 (odd) keeps the per-thread LDS accesses conflict-free
enum { TINY_CHUNK = 256 };
template <int N> constexpr int tiny_pitch() { return N | 1; }

template <int N, class R>
DSP_HD void tiny_row_load(const TinyArgsT<R> &a, R *lds, long long bin, int cnt, int tid, int nthr)
{
	for (int e = tid; e < cnt * N; e += nthr) {
		const long long off = bin + e;
		const bool drop = a.mask && a.mask[a.mask_div.div((uint32_t)off)] != a.mask_id;
		lds[(e / N) * tiny_pitch<N>() + e % N] = drop ? R(0) : a.in[off];
	}
}
template <int N, int KIND, class R>
DSP_HD void tiny_row_compute(const TinyArgsT<R> &a, R *lds, int cnt, int tid)
{
	if (tid >= cnt) return;
	R x[N], y[N];
	R *p = lds + tid * tiny_pitch<N>();
#pragma unroll
	for (int j = 0; j < N; j++) x[j] = p[j];
	tiny_dct<N, KIND>(a, x, y);
#pragma unroll
	for (int k = 0; k < N; k++) p[k] = y[k];
}
template <int N, class R>
DSP_HD void tiny_row_store(const TinyArgsT<R> &a, const R *lds, long long bout, int cnt, int tid, int nthr)
{
	for (int e = tid; e < cnt * N; e += nthr) {
		const R v = lds[(e / N) * tiny_pitch<N>() + e % N];
		if (a.accumulate) a.out[bout + e] += v; else a.out[bout + e] = v;
	}
}
// workgroup -> (chunk of dimension 0, element of the remaining dimensions)
template <class R>
DSP_HD void tiny_row_base(const TinyArgsT<R> &a, uint32_t wg, long long &bin, long long &bout, int &cnt)
{
	const uint32_t r = a.chunk_div.div_exact(wg), c = wg - r * a.chunk_div.d;
	tiny_base(a, r, 1, bin, bout);
	bin += (long long)c * TINY_CHUNK * a.N; bout += (long long)c * TINY_CHUNK * a.N;
	cnt = a.bn[0] - (int)c * TINY_CHUNK;
	if (cnt > TINY_CHUNK) cnt = TINY_CHUNK;
}

// ------------------------------------------------------------------------------------------------
// DENSE pass: any N, any stride; O(N^2) by the definition with an exactly reduced phase table.
// LDS: x[N] floats.  cosTab[t] = cos(pi t / (2N)), t in [0, 4N).
struct DenseGeom {
	int N, kind;
	long long es_in, es_out;
	int nb0, nb1, nb2;                       // three batch levels
	long long sb0_in, sb1_in, sb2_in, sb0_out, sb1_out, sb2_out;
	const uint32_t *mask;
	uint32_t mask_id;
	FastDiv mask_div;
	int accumulate;
};
template <class R>
struct DenseArgsT : DenseGeom {
	const R *in;
	R *out;
	const R *cosTab;
	R scale, out_scale0, in_scale0;
	R *stage;          // lines too long for LDS: the (masked, scaled) samples of every line are copied here first -- [line][N], owned by
	                   // the plan -- and the sums read them from memory (a second launch, so in-place transforms stay correct)
};
typedef DenseArgsT<float> DenseArgs;
typedef DenseArgsT<double> DenseArgsD;

DSP_HD void dense_base(const DenseGeom &a, long long line, long long &bin, long long &bout)
{
	const long long i2 = line / ((long long)a.nb0 * a.nb1), r = line - i2 * a.nb0 * a.nb1;
	const long long i1 = r / a.nb0, i0 = r - i1 * a.nb0;
	bin = i0 * a.sb0_in + i1 * a.sb1_in + i2 * a.sb2_in;
	bout = i0 * a.sb0_out + i1 * a.sb1_out + i2 * a.sb2_out;
}
template <class R>
DSP_HD void dense_load(const DenseArgsT<R> &a, R *x, long long bin, int tid, int nthr)
{
	for (int j = tid; j < a.N; j += nthr) {
		const long long off = bin + (long long)j * a.es_in;
		const R v = (a.mask && a.mask[a.mask_div.div((uint32_t)off)] != a.mask_id) ? R(0) : a.in[off];
		x[j] = v * (j == 0 ? a.in_scale0 : R(1));
	}
}
// sum_{j in [j0, N)} x[j] cos(pi (t0 + (j-j0) step) / (2N)) with four independent accumulators (shorter dependency
// chains, and a quarter of the sequential rounding growth)
template <class R>
DSP_HD R dense_dot(const R *x, const R *cosTab, int j0, int N, int t0, int step)
{
	const int fourN = 4 * N;
	int t[4];
	t[0] = t0;
	for (int u = 1; u < 4; u++) { t[u] = t[u - 1] + step; if (t[u] >= fourN) t[u] -= fourN; }
	int step4 = step;
	for (int u = 0; u < 2; u++) { step4 += step4; if (step4 >= fourN) step4 -= fourN; }
	R acc[4] = {R(0), R(0), R(0), R(0)};
	int j = j0;
	for (; j + 4 <= N; j += 4)
		for (int u = 0; u < 4; u++) {
			acc[u] += x[j + u] * cosTab[t[u]];
			t[u] += step4; if (t[u] >= fourN) t[u] -= fourN;
		}
	for (int u = 0; j < N; j++, u++) acc[u] += x[j] * cosTab[t[u]];
	return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

template <class R>
DSP_HD void dense_compute(const DenseArgsT<R> &a, const R *x, long long bout, int tid, int nthr)
{
	const int N = a.N, fourN = 4 * N;
	for (int k = tid; k < N; k += nthr) {
		R acc;
		if (a.kind == KIND_REDFT10) {
			// phase (2j+1) k mod 4N: k at j = 0, then + 2k per sample
			acc = R(2) * dense_dot(x, a.cosTab, 0, N, k % fourN, (2 * k) % fourN);
		} else {
			// phase j (2k+1) mod 4N, j >= 1
			const int step = (2 * k + 1) % fourN;
			acc = x[0] + R(2) * dense_dot(x, a.cosTab, 1, N, step, step);
		}
		const R r = acc * a.scale * (k == 0 ? a.out_scale0 : R(1));
		if (a.accumulate) a.out[bout + (long long)k * a.es_out] += r; else a.out[bout + (long long)k * a.es_out] = r;
	}
}

}  // namespace dspfft
