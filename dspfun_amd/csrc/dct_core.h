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
//           N even; each real signal is packed into an N/2-point complex FFT (even/odd "Makhoul" order).
//   COL  -- the axis is strided and an inner contiguous dimension exists: a workgroup owns a tile of K
//           adjacent floats x all N rows; adjacent float columns are paired into one complex signal
//           (two real transforms per complex FFT of length N; any N).
// In both, the complex FFT runs in LDS, in place, decimation in frequency, mixed radix.
#pragma once
#include "radix.h"

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

struct PassArgs {
	const float *in;
	float *out;
	int N;            // real transform length
	int kind;         // KIND_*
	// ROW
	int C;            // interleaved signals per line
	int Bg;           // channels transformed together (C, or 1 when LDS is tight)
	// COL
	int K;            // tile width in floats (even)
	int B;            // complex columns per tile = K/2
	int ninner;       // extent of the inner contiguous dimension
	int ntiles;       // ceil(ninner / K)
	long long es_in, es_out;   // stride (elements) between consecutive samples of the axis (COL)
	// batch dimensions (two levels) -- line/tile base = i0*sb0 + i1*sb1
	int nb0, nb1;
	long long sb0_in, sb1_in, sb0_out, sb1_out;
	// tables (device memory)
	const cf *T;          // T[j] = exp(-i pi j / (2N)), j in [0, N]
	const cf *W;          // W[t] = exp(-2 pi i t / L),  t in [0, L)
	const uint32_t *pos;  // pos[k] = LDS slot holding FFT output k after the DIF stages
	float scale;          // every output is multiplied by scale ...
	float out_scale0;     // ... and output index 0 of this axis additionally by out_scale0
	float in_scale0;      // input index 0 of this axis is multiplied by in_scale0 before transforming
	// fused scan step (scan/scan.c:429-459): the FIRST pass zeroes every input element whose owner
	// id differs (mask[offset / mask_div] != mask_id); the LAST pass adds into `out` instead of storing
	const uint32_t *mask;
	uint32_t mask_id;
	FastDiv mask_div;     // elements per owner id (32-bit offsets: masked runs are limited to 2^32 / d elements)
	int accumulate;
	FftDesc fft;
	FastDiv divB;         // divide by (ROW: Bg, COL: B)
};

DSP_HD float masked(const PassArgs &a, long long off, float v)
{
	if (!a.mask) return v;
	return a.mask[a.mask_div.div((uint32_t)off)] != a.mask_id ? 0.f : v;
}

// ------------------------------------------------------------------------------------------------
// FFT stage (in place in LDS).  Signals are interleaved: element n of signal s is buf[n*B + s].
template <int R>
DSP_HD void fft_stage_r(cf *buf, int L, const StageDesc &S, int B, FastDiv divB, const cf *W, int tid, int nthr)
{
	const int nitems = (L / R) * B;
	const int stride = S.M1 * B;
	for (int it = tid; it < nitems; it += nthr) {
		const int q = (int)divB.div((uint32_t)it), s = it - q * B;
		const int blk = (int)S.divM1.div((uint32_t)q), m = q - blk * S.M1;
		const int base = (blk * S.Lc + m) * B + s;
		cf x[R];
		static_for<0, R>([&](auto r) { x[r] = buf[base + r * stride]; });
		Dft<R>::run(x);
		if (S.M1 > 1) {
			const int tw = m * S.twstep;
			static_for<1, R>([&](auto r) { x[r] = cmul(x[r], W[tw * r]); });
		}
		static_for<0, R>([&](auto r) { buf[base + r * stride] = x[r]; });
	}
}

DSP_HD void fft_stage(cf *buf, int L, const StageDesc &S, int B, FastDiv divB, const cf *W, int tid, int nthr)
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

// ------------------------------------------------------------------------------------------------
// ROW pass.  LDS: raw[N*C] floats (the line as it lies in memory) + buf[L*Bg] complex, L = N/2.
// index of sample v[n] of the even/odd-reordered signal inside the original signal
DSP_HD int makhoul_src(int n, int N) { return 2 * n < N ? 2 * n : 2 * (N - 1 - n) + 1; }

DSP_HD void row_base(const PassArgs &a, int line, long long &bin, long long &bout)
{
	const int i1 = line / a.nb0, i0 = line - i1 * a.nb0;
	bin = i0 * a.sb0_in + i1 * a.sb1_in;
	bout = i0 * a.sb0_out + i1 * a.sb1_out;
}

DSP_HD void row_load(const PassArgs &a, float *raw, long long bin, int tid, int nthr)
{
	const int n = a.N * a.C;
	const float *src = a.in + bin;
	if (a.mask) {
		for (int i = tid; i < n; i += nthr) raw[i] = masked(a, bin + i, src[i]);
	} else if (((bin & 3) == 0) && ((n & 3) == 0) && ((((uintptr_t)a.in) & 15) == 0)) {
		const float4 *s4 = reinterpret_cast<const float4 *>(src);
		float4 *r4 = reinterpret_cast<float4 *>(raw);
		for (int i = tid; i < n / 4; i += nthr) r4[i] = s4[i];
	} else {
		for (int i = tid; i < n; i += nthr) raw[i] = src[i];
	}
}

DSP_HD void row_store(const PassArgs &a, const float *raw, long long bout, int tid, int nthr)
{
	const int n = a.N * a.C;
	float *dst = a.out + bout;
	if (a.accumulate) {
		for (int i = tid; i < n; i += nthr) dst[i] += raw[i];
	} else if (((bout & 3) == 0) && ((n & 3) == 0) && ((((uintptr_t)a.out) & 15) == 0)) {
		const float4 *r4 = reinterpret_cast<const float4 *>(raw);
		float4 *d4 = reinterpret_cast<float4 *>(dst);
		for (int i = tid; i < n / 4; i += nthr) d4[i] = r4[i];
	} else {
		for (int i = tid; i < n; i += nthr) dst[i] = raw[i];
	}
}

// REDFT10: pack channels [c0, c0+Bg) of raw into buf
DSP_HD void row_pack2(const PassArgs &a, const float *raw, cf *buf, int c0, int tid, int nthr)
{
	const int L = a.N / 2, Bg = a.Bg, C = a.C, N = a.N;
	for (int it = tid; it < L * Bg; it += nthr) {
		const int m = (int)a.divB.div((uint32_t)it), s = it - m * Bg;
		const int i0 = makhoul_src(2 * m, N), i1 = makhoul_src(2 * m + 1, N);
		float re = raw[i0 * C + c0 + s], im = raw[i1 * C + c0 + s];
		if (i0 == 0) re *= a.in_scale0;
		buf[it] = cmk(re, im);
	}
}

// REDFT10: FFT output -> 4 real outputs per (k, L-k) pair, written back into raw
DSP_HD void row_post2(const PassArgs &a, float *raw, const cf *buf, int c0, int tid, int nthr)
{
	const int L = a.N / 2, Bg = a.Bg, C = a.C, N = a.N;
	const int nk = L / 2 + 1;
	for (int it = tid; it < nk * Bg; it += nthr) {
		const int k = (int)a.divB.div((uint32_t)it), s = it - k * Bg;
		const int km = k ? L - k : 0;
		const cf zk = buf[a.pos[k] * Bg + s];
		const cf zm = cconj(buf[a.pos[km] * Bg + s]);
		const cf E = cscale(cadd(zk, zm), 0.5f);
		const cf Dh = cscale(csub(zk, zm), 0.5f);
		const cf D = cmul_mi(Dh);                 // (zk - conj zm) / (2i)
		const cf P = cmul(a.T[4 * k], D);         // exp(-2 pi i k / N) * D
		const cf Vk = cadd(E, P);
		const cf Vm = cconj(csub(E, P));          // V[L-k]
		const cf wk = cmul(a.T[k], Vk);
		const cf wm = cmul(a.T[L - k], Vm);
		float *o = raw + c0 + s;
		const float sc = a.scale;
		o[k * C] = 2.f * wk.x * sc * (k == 0 ? a.out_scale0 : 1.f);
		if (k > 0) o[(N - k) * C] = -2.f * wk.y * sc;
		o[(L - k) * C] = 2.f * wm.x * sc;
		if (k > 0) o[(L + k) * C] = -2.f * wm.y * sc;
	}
}

// REDFT01: natural-order input in raw -> conj of the half-length spectrum in buf
DSP_HD void row_pre3(const PassArgs &a, const float *raw, cf *buf, int c0, int tid, int nthr)
{
	const int L = a.N / 2, Bg = a.Bg, C = a.C, N = a.N;
	const int nk = L / 2 + 1;
	for (int it = tid; it < nk * Bg; it += nthr) {
		const int k = (int)a.divB.div((uint32_t)it), s = it - k * Bg;
		const float *x = raw + c0 + s;
		const float xk = x[k * C] * (k == 0 ? a.in_scale0 : 1.f);
		const float xnk = k ? x[(N - k) * C] : 0.f;
		const float xlk = x[(L - k) * C];
		const float xlpk = x[(L + k) * C];                     // k <= L/2 so L+k <= N-1
		const cf Vk = cmulc(cmk(xk, -xnk), a.T[k]);            // conj(T[k]) * (X[k] - i X[N-k])
		const cf Vm = cmulc(cmk(xlk, -xlpk), a.T[L - k]);      // V[L-k]
		const cf S = cadd(Vk, cconj(Vm));
		const cf D = csub(Vk, cconj(Vm));
		const cf Q = cmul_pi(cmulc(D, a.T[4 * k]));            // i * conj(t1[k]) * D
		buf[k * Bg + s] = cconj(cadd(S, Q));
		if (k > 0) buf[(L - k) * Bg + s] = csub(S, Q);
	}
}

// REDFT01: FFT output -> time samples, un-reordered, into raw
DSP_HD void row_unpack3(const PassArgs &a, float *raw, const cf *buf, int c0, int tid, int nthr)
{
	const int L = a.N / 2, Bg = a.Bg, C = a.C, N = a.N;
	for (int it = tid; it < L * Bg; it += nthr) {
		const int m = (int)a.divB.div((uint32_t)it), s = it - m * Bg;
		const cf F = buf[a.pos[m] * Bg + s];
		const int i0 = makhoul_src(2 * m, N), i1 = makhoul_src(2 * m + 1, N);
		raw[i0 * C + c0 + s] = F.x * a.scale * (i0 == 0 ? a.out_scale0 : 1.f);
		raw[i1 * C + c0 + s] = -F.y * a.scale;
	}
}

// ------------------------------------------------------------------------------------------------
// COL pass.  LDS: buf[N*B] complex.  Tile t of batch (i0,i1) covers floats [t*K, t*K+K) of the inner dim.
DSP_HD void col_base(const PassArgs &a, int wg, long long &bin, long long &bout, int &valid)
{
	const int bt = wg / a.ntiles, t = wg - bt * a.ntiles;
	const int i1 = bt / a.nb0, i0 = bt - i1 * a.nb0;
	bin = i0 * a.sb0_in + i1 * a.sb1_in + (long long)t * a.K;
	bout = i0 * a.sb0_out + i1 * a.sb1_out + (long long)t * a.K;
	valid = a.ninner - t * a.K;   // number of valid float columns in this tile (may exceed K)
	if (valid > a.K) valid = a.K;
}

DSP_HD cf ld2m(const PassArgs &a, long long off, bool vec, int nvalid);
DSP_HD cf ld2(const float *p, bool vec, int nvalid)
{
	if (nvalid >= 2) {
		if (vec) { const float2 v = *reinterpret_cast<const float2 *>(p); return cmk(v.x, v.y); }
		return cmk(p[0], p[1]);
	}
	return cmk(nvalid >= 1 ? p[0] : 0.f, 0.f);
}
DSP_HD cf ld2m(const PassArgs &a, long long off, bool vec, int nvalid)
{
	cf v = ld2(a.in + off, vec, nvalid);
	if (a.mask) { v.x = masked(a, off, v.x); if (nvalid >= 2) v.y = masked(a, off + 1, v.y); }
	return v;
}
DSP_HD void st2(float *p, bool vec, int nvalid, float a, float b);
DSP_HD void st2a(const PassArgs &a, float *p, bool vec, int nvalid, float x, float y)
{
	if (a.accumulate) { if (nvalid >= 1) p[0] += x; if (nvalid >= 2) p[1] += y; }
	else st2(p, vec, nvalid, x, y);
}
DSP_HD void st2(float *p, bool vec, int nvalid, float a, float b)
{
	if (nvalid >= 2) {
		if (vec) { float2 v; v.x = a; v.y = b; *reinterpret_cast<float2 *>(p) = v; }
		else { p[0] = a; p[1] = b; }
	} else if (nvalid >= 1) p[0] = a;
}

DSP_HD int makhoul_dst(int y, int N) { return (y & 1) ? N - 1 - (y >> 1) : (y >> 1); }

// REDFT10: load tile rows, even/odd reorder along the axis
DSP_HD void col_load2(const PassArgs &a, cf *buf, long long bin, int valid, int tid, int nthr)
{
	const int N = a.N, B = a.B;
	const bool vec = (((bin | a.es_in) & 1) == 0) && ((((uintptr_t)a.in) & 7) == 0);
	for (int it = tid; it < N * B; it += nthr) {
		const int y = (int)a.divB.div((uint32_t)it), j = it - y * B;
		cf v = ld2m(a, bin + (long long)y * a.es_in + 2 * j, vec, valid - 2 * j);
		if (y == 0) { v.x *= a.in_scale0; v.y *= a.in_scale0; }
		buf[makhoul_dst(y, N) * B + j] = v;
	}
}

// REDFT10: separate the two real transforms, quarter-sample twiddle, store rows k and N-k
DSP_HD void col_post2(const PassArgs &a, const cf *buf, long long bout, int valid, int tid, int nthr)
{
	const int N = a.N, B = a.B;
	const int nk = N / 2 + 1;
	const bool vec = (((bout | a.es_out) & 1) == 0) && ((((uintptr_t)a.out) & 7) == 0);
	for (int it = tid; it < nk * B; it += nthr) {
		const int k = (int)a.divB.div((uint32_t)it), j = it - k * B;
		const int km = k ? N - k : 0;
		const cf zk = buf[a.pos[k] * B + j];
		const cf zm = cconj(buf[a.pos[km] * B + j]);
		const cf A2 = cadd(zk, zm);                   // 2 A[k]
		const cf B2 = cmul_mi(csub(zk, zm));          // 2 B[k]
		const cf t = a.T[k];
		const cf wa = cmul(t, A2), wb = cmul(t, B2);
		const float sc = a.scale;
		float *o = a.out + bout + 2 * j;
		const float s0 = (k == 0) ? sc * a.out_scale0 : sc;
		st2a(a, o + (long long)k * a.es_out, vec, valid - 2 * j, wa.x * s0, wb.x * s0);
		if (k > 0 && km != k) st2a(a, o + (long long)km * a.es_out, vec, valid - 2 * j, -wa.y * sc, -wb.y * sc);
	}
}

// REDFT01: rows k and N-k -> conj spectrum of (a + i b)
DSP_HD void col_pre3(const PassArgs &a, cf *buf, long long bin, int valid, int tid, int nthr)
{
	const int N = a.N, B = a.B;
	const int nk = N / 2 + 1;
	const bool vec = (((bin | a.es_in) & 1) == 0) && ((((uintptr_t)a.in) & 7) == 0);
	for (int it = tid; it < nk * B; it += nthr) {
		const int k = (int)a.divB.div((uint32_t)it), j = it - k * B;
		const int km = k ? N - k : 0;
		const long long p = bin + 2 * j;
		cf xk = ld2m(a, p + (long long)k * a.es_in, vec, valid - 2 * j);
		cf xm = k ? ld2m(a, p + (long long)km * a.es_in, vec, valid - 2 * j) : cmk(0.f, 0.f);
		if (k == 0) { xk.x *= a.in_scale0; xk.y *= a.in_scale0; }
		const cf t = a.T[k];
		const cf Va = cmulc(cmk(xk.x, -xm.x), t);     // conj(T[k]) (Xa[k] - i Xa[N-k])
		const cf Vb = cmulc(cmk(xk.y, -xm.y), t);
		buf[k * B + j] = cmk(Va.x - Vb.y, -Va.y - Vb.x);            // conj(Va) - i conj(Vb)
		if (k > 0) buf[km * B + j] = cmk(Va.x + Vb.y, Va.y - Vb.x); // Va - i Vb
	}
}

// REDFT01: FFT output -> both real signals, un-reordered, store
DSP_HD void col_unpack3(const PassArgs &a, const cf *buf, long long bout, int valid, int tid, int nthr)
{
	const int N = a.N, B = a.B;
	const bool vec = (((bout | a.es_out) & 1) == 0) && ((((uintptr_t)a.out) & 7) == 0);
	for (int it = tid; it < N * B; it += nthr) {
		const int n = (int)a.divB.div((uint32_t)it), j = it - n * B;
		const cf F = buf[a.pos[n] * B + j];
		const int y = makhoul_src(n, N);
		const float sc = (y == 0) ? a.scale * a.out_scale0 : a.scale;
		st2a(a, a.out + bout + (long long)y * a.es_out + 2 * j, vec, valid - 2 * j, F.x * sc, -F.y * sc);
	}
}

// ------------------------------------------------------------------------------------------------
// DENSE pass: any N, any stride; O(N^2) by the definition with an exactly reduced phase table.
// LDS: x[N] floats.  cosTab[t] = cos(pi t / (2N)), t in [0, 4N).
struct DenseArgs {
	const float *in;
	float *out;
	int N, kind;
	long long es_in, es_out;
	int nb0, nb1, nb2;                       // three batch levels
	long long sb0_in, sb1_in, sb2_in, sb0_out, sb1_out, sb2_out;
	const float *cosTab;
	float scale, out_scale0, in_scale0;
	const uint32_t *mask;
	uint32_t mask_id;
	FastDiv mask_div;
	int accumulate;
};

DSP_HD void dense_base(const DenseArgs &a, long long line, long long &bin, long long &bout)
{
	const long long i2 = line / ((long long)a.nb0 * a.nb1), r = line - i2 * a.nb0 * a.nb1;
	const long long i1 = r / a.nb0, i0 = r - i1 * a.nb0;
	bin = i0 * a.sb0_in + i1 * a.sb1_in + i2 * a.sb2_in;
	bout = i0 * a.sb0_out + i1 * a.sb1_out + i2 * a.sb2_out;
}
DSP_HD void dense_load(const DenseArgs &a, float *x, long long bin, int tid, int nthr)
{
	for (int j = tid; j < a.N; j += nthr) {
		const long long off = bin + (long long)j * a.es_in;
		const float v = (a.mask && a.mask[a.mask_div.div((uint32_t)off)] != a.mask_id) ? 0.f : a.in[off];
		x[j] = v * (j == 0 ? a.in_scale0 : 1.f);
	}
}
DSP_HD void dense_compute(const DenseArgs &a, const float *x, long long bout, int tid, int nthr)
{
	const int N = a.N, fourN = 4 * N;
	for (int k = tid; k < N; k += nthr) {
		float acc = 0.f;
		if (a.kind == KIND_REDFT10) {
			int t = k % fourN;                 // (2j+1) k mod 4N, j = 0
			const int step = (2 * k) % fourN;
			for (int j = 0; j < N; j++) {
				acc += x[j] * a.cosTab[t];
				t += step; if (t >= fourN) t -= fourN;
			}
			acc *= 2.f;
		} else {
			const int step = (2 * k + 1) % fourN;   // j (2k+1) mod 4N
			int t = step;
			for (int j = 1; j < N; j++) {
				acc += x[j] * a.cosTab[t];
				t += step; if (t >= fourN) t -= fourN;
			}
			acc = x[0] + 2.f * acc;
		}
		const float r = acc * a.scale * (k == 0 ? a.out_scale0 : 1.f);
		if (a.accumulate) a.out[bout + (long long)k * a.es_out] += r; else a.out[bout + (long long)k * a.es_out] = r;
	}
}

}  // namespace dspfft
