// dct_fold.h -- ROW passes over lines too long for more than one workgroup per CU (7680 x 3 floats: 92 KB of LDS as one N/2-point FFT),
// FOLDED into two half-length transforms that run one after the other through HALF the LDS (46 KB: two or three workgroups per CU).
//
// With H = N/2 and s[n] = x[n] + x[N-1-n], d[n] = x[n] - x[N-1-n] (n < H):
//   REDFT10_N(x)[2k] = REDFT10_H(s)[k]        REDFT10_N(x)[2k+1] = REDFT11_H(d)[k]            (REDFT11 = DCT-IV, FFTW's definition)
//   REDFT01_N(X)[n] = e[n] + o[n],  REDFT01_N(X)[N-1-n] = e[n] - o[n],   e = REDFT01_H(X[2k]),  o = REDFT11_H(X[2k+1])
// "A" is the half-length REDFT10 / REDFT01: exactly RowSpecG<float, H, ...>'s packing into an M = H/2-point complex FFT, its stages and its
// pre / post arithmetic.  "B" is the DCT-IV, through the same M-point FFT:
//   t[n] = (d[2n] + i d[H-1-2n]) Om[n],  Om[n] = exp(-i pi (4n+1) / (4H));   U = FFT_M(t);   u[k] = U[k] Rh[k],  Rh[k] = exp(-i pi k / H);
//   y[2k] = 2 Re u[k],   y[H-1-2k] = -2 Im u[k]
// A and B meet in registers, never in partial cache lines (measured, tools/r05_probe.hip: storing every other pixel and filling the gaps a few
// microseconds later runs at 3.2 TB/s, whole stores at 5.1): a thread owns ORBITS {2q, 2q+1, H-2-2q, H-1-2q} of the half-length index -- the index
// pairs B's pre and post steps couple -- so that
//   REDFT10  loads pixel pairs (2q, 2q+1), (N-2-2q, N-1-2q), (H-2-2q, H-1-2q), (H+2q, H+2q+1)  and stores pixels 4q .. 4q+3, N-4-4q .. N-1-4q
//   REDFT01  loads pixels 4q .. 4q+3, N-4-4q .. N-1-4q  and stores those four pixel pairs
// (48 or 24 contiguous bytes per lane, consecutive lanes adjacent).  A's own pre / post steps couple {k, H-k, M-k, M+k} instead; its values
// change owner through the plane itself: REDFT10's post step leaves its four real outputs in the two slots it read (slot k: y[k], y[H-k];
// slot M-k: y[M-k], y[M+k]; y[M] rides in slot 0), REDFT01's pre step finds its inputs laid out the same way -- one barrier each, no second
// buffer.  The half not in the plane waits in registers (N C / (2 T) floats per thread).
//
// No reference code corresponds to this file (the reference delegates every transform to FFTW, include/precision.h:115).
#pragma once
#include "dct_spec.h"

namespace dspfft {

// row pairs of a split column pass (spec_kernels.h row_pair_kernel: lines y1 = 2n, y2 = N-1-2n of the split axis -> r1 + r2, r1 - r2), ONE
// output line per workgroup: the partners -- workgroups b and b + 8, the same XCD's L2, dispatched back to back -- both read both lines.
// work item b of 2 * npairs -> (pair, which line)
DSP_HD void fold_pair_work(int b, int npairs, int &pair, int &h)
{
	const int full = npairs >> 3;
	if (b < full * 16) { pair = (b >> 4) * 8 + (b & 7); h = (b >> 3) & 1; }
	else { const int r = b - full * 16; pair = full * 8 + (r >> 1); h = r & 1; }
}

template <int N_, int C_, int T_, int... Rs>
struct RowFoldT {
	typedef RowSpecG<float, N_ / 2, C_, C_, T_, Rs...> F;      // the half-length transform's packing, stages and tables
	typedef float Re;
	typedef cf CX;
	typedef PassArgs PA;
	static constexpr int N = N_, C = C_, T = T_, H = N_ / 2, M = N_ / 4, NS = F::NS, PL = F::PL;
	static constexpr int NQ = M / 2, QR = (NQ + T_ - 1) / T_;           // orbits and rounds of them per thread
	static constexpr int NK = M / 2 + 1, KR = (NK + T_ - 1) / T_;       // A's (k, M-k) items
	static constexpr size_t LDS = F::LDS;
	static_assert(N_ % 8 == 0, "fold needs N divisible by 8");
	static_assert(F::L == M, "half-length packing");
	// one table array (engine.cpp fold_tables), complex floats: Th[k] = exp(-i pi k / (2H)), k <= H | Wq[t] = exp(-2 pi i t / M) | Om | Rh
	static constexpr int OFF_T = 0, OFF_W = H + 1, OFF_OM = OFF_W + M, OFF_RH = OFF_OM + M, TAB_LEN = OFF_RH + M;
	// waves per SIMD asked of the register allocator: two workgroups per CU (tools/r05_probe.hip: 150-162 us per 8K pass against 155-162 for three)
#ifndef DSP_FOLD_WGS
#define DSP_FOLD_WGS 2
#endif
	static constexpr int WPE = DSP_FOLD_WGS * T_ / 256 > 8 ? 8 : (DSP_FOLD_WGS * T_ / 256 < 1 ? 1 : DSP_FOLD_WGS * T_ / 256);

	struct Orb { float v[QR][4][C_]; };                         // [round][orbit member j: 2q, 2q+1, H-2-2q, H-1-2q][channel]
	struct Last { CX x[F::LAST_ROUNDS * F::RL]; };
	template <int I> static DSP_HD bool item(int tid, int &q) { q = tid + I * T; return (I + 1) * T <= NQ || q < NQ; }
	template <int I> static DSP_HD bool kitem(int tid, int &k) { k = tid + I * T; return (I + 1) * T <= NK || k < NK; }
	// float position of real sample r of the even/odd-reordered half-length signal inside a channel plane (RowSpecG phase 0's)
	static DSP_HD int fpos(int r) { return 2 * F::padded(r >> 1) + (r & 1); }

	// ---- global access: 4 C consecutive floats (four pixels), 16-byte aligned ----
	static DSP_HD void ld4(const float *p, float *v) { static_for<0, C>([&](auto i) { const float4 t = *reinterpret_cast<const float4 *>(p + 4 * i); v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }); }
	static DSP_HD void st4(float *p, const float *v) { static_for<0, C>([&](auto i) { float4 t; t.x = v[4 * i]; t.y = v[4 * i + 1]; t.z = v[4 * i + 2]; t.w = v[4 * i + 3]; *reinterpret_cast<float4 *>(p + 4 * i) = t; }); }

	// ---- REDFT10: the line's pixel pairs -> s (sum with the mirror pixel) and d (difference), orbit order ----
	// PAIR: x = line + sg * line2 (the row-pair butterfly of a split column pass)
	template <bool PAIR>
	static DSP_HD void load10(const PA &a, long long bin, long long bin2, float sg, int tid, Orb &s, Orb &d)
	{
		static_for<0, QR>([&](auto i) {
			int q;
			if (!item<i>(tid, q)) return;
			// lo[j] = x[m_j], hi[j] = x[N-1-m_j]
			const int plo[4] = {2 * q, 2 * q + 1, H - 2 - 2 * q, H - 1 - 2 * q}, phi[4] = {N - 1 - 2 * q, N - 2 - 2 * q, H + 1 + 2 * q, H + 2 * q};
			Pix<C, float> lo[4], hi[4];
			static_for<0, 4>([&](auto j) { lo[j] = load_pix<C, float>(a.in + bin + (long long)plo[j] * C); hi[j] = load_pix<C, float>(a.in + bin + (long long)phi[j] * C); });
			if constexpr (PAIR) {
				static_for<0, 4>([&](auto j) {
					const Pix<C, float> l2 = load_pix<C, float>(a.in + bin2 + (long long)plo[j] * C), h2 = load_pix<C, float>(a.in + bin2 + (long long)phi[j] * C);
					static_for<0, C>([&](auto c) { lo[j].v[c] += sg * l2.v[c]; hi[j].v[c] += sg * h2.v[c]; });
				});
			}
			if (q == 0) static_for<0, C>([&](auto c) { lo[0].v[c] *= a.in_scale0; });
			static_for<0, 4>([&](auto j) { static_for<0, C>([&](auto c) { s.v[i][j][c] = lo[j].v[c] + hi[j].v[c]; d.v[i][j][c] = lo[j].v[c] - hi[j].v[c]; }); });
		});
	}
	// ---- REDFT01: pixels 4q .. 4q+3 and N-4-4q .. N-1-4q -> even coefficients xe, odd coefficients xo, orbit order ----
	// FLAGGED: the line follows a masked column pass that skipped its empty tiles (PassGeom::zflags): those read as zeros
	template <bool FLAGGED>
	static DSP_HD void load01_line(const PA &a, long long bin, const uint8_t *zf, int q, float *lo, float *hi)
	{
		if constexpr (FLAGGED) {
			static_for<0, 4>([&](auto t) {
				const int p0 = 4 * q + t, p1 = N - 4 - 4 * q + t;
				const Pix<C, float> u = load_pix_z<C, float>(a, zf, p0 * C, bin + (long long)p0 * C), w = load_pix_z<C, float>(a, zf, p1 * C, bin + (long long)p1 * C);
				static_for<0, C>([&](auto c) { lo[t * C + c] = u.v[c]; hi[t * C + c] = w.v[c]; });
			});
		} else {
			ld4(a.in + bin + (long long)(4 * q) * C, lo);
			ld4(a.in + bin + (long long)(N - 4 - 4 * q) * C, hi);
		}
	}
	template <bool PAIR, bool FLAGGED>
	static DSP_HD void load01(const PA &a, long long bin, long long bin2, float sg, const uint8_t *zf, const uint8_t *zf2, int tid, Orb &xe, Orb &xo)
	{
		static_for<0, QR>([&](auto i) {
			int q;
			if (!item<i>(tid, q)) return;
			float lo[4 * C], hi[4 * C];
			load01_line<FLAGGED>(a, bin, zf, q, lo, hi);
			if constexpr (PAIR) {
				float lo2[4 * C], hi2[4 * C];
				load01_line<FLAGGED>(a, bin2, zf2, q, lo2, hi2);
				static_for<0, 4 * C>([&](auto e) { lo[e] += sg * lo2[e]; hi[e] += sg * hi2[e]; });
			}
			if (q == 0) static_for<0, C>([&](auto c) { lo[c] *= a.in_scale0; });
			static_for<0, C>([&](auto c) {
				xe.v[i][0][c] = lo[0 * C + c]; xo.v[i][0][c] = lo[1 * C + c]; xe.v[i][1][c] = lo[2 * C + c]; xo.v[i][1][c] = lo[3 * C + c];
				xe.v[i][2][c] = hi[0 * C + c]; xo.v[i][2][c] = hi[1 * C + c]; xe.v[i][3][c] = hi[2 * C + c]; xo.v[i][3][c] = hi[3 * C + c];
			});
		});
	}

	// ---- A, REDFT10: s -> the plane, even/odd reordered (v[j] = s[2j], v[H-1-j] = s[2j+1]) and packed two reals to a slot ----
	static DSP_HD void a_scatter10(CX *planes, int tid, const Orb &s)
	{
		float *pf = reinterpret_cast<float *>(planes);
		static_for<0, QR>([&](auto i) {
			int q;
			if (!item<i>(tid, q)) return;
			const int f[4] = {fpos(q), fpos(H - 1 - q), fpos(M - 1 - q), fpos(M + q)};
			static_for<0, 4>([&](auto j) { static_for<0, C>([&](auto c) { pf[c * (2 * PL) + f[j]] = s.v[i][j][c]; }); });
		});
	}
	// A's closing step (RowSpecG::phase<KIND_REDFT10, NS + 2>) with the four real outputs of item k left IN PLACE: slot k = (y[k], y[H-k]),
	// slot M-k = (y[M-k], y[M+k]); k = 0: slot 0 = (y[0], y[M])
	static DSP_HD void a_post10(const PA &a, const CX *tab, CX *planes, int tid)
	{
		static_for<0, KR>([&](auto i) {
			int k;
			if (!kitem<i>(tid, k)) return;
			const int km = k ? M - k : 0;
			const CX tk = tab[OFF_T + k];
			const CX tlk = cmul(cconj(tk), cmk<Re>((Re)0.70710678118654752440, (Re)-0.70710678118654752440));
			const CX t1 = csqr(csqr(tk));
			const Re sc = a.scale;
			static_for<0, C>([&](auto c) {
				const CX zk = planes[c * PL + k];
				const CX zm = cconj(planes[c * PL + km]);
				const CX E = cadd(zk, zm);
				const CX D = cmul_mi(csub(zk, zm));
				const CX P = cmul(t1, D);
				const CX wk = cmul(tk, cadd(E, P));
				const CX wm = cmul(tlk, cconj(csub(E, P)));
				if (k == 0) planes[c * PL] = cmk<Re>(wk.x * sc * a.out_scale0, wm.x * sc);
				else {
					planes[c * PL + k] = cmk<Re>(wk.x * sc, -wk.y * sc);
					if (km != k) planes[c * PL + km] = cmk<Re>(wm.x * sc, -wm.y * sc);
				}
			});
		});
	}
	// ... and collected in orbit order: y[2q], y[2q+1] = slots 2q, 2q+1 (.x); y[H-1-2q] = slot 2q+1 (.y); y[H-2-2q] = slot 2q+2 (.y), y[M] = slot 0 (.y)
	static DSP_HD void a_gather10(const CX *planes, int tid, Orb &E)
	{
		static_for<0, QR>([&](auto i) {
			int q;
			if (!item<i>(tid, q)) return;
			const int s2 = (2 * q + 2 < M) ? 2 * q + 2 : 0;
			static_for<0, C>([&](auto c) {
				const CX u = planes[c * PL + 2 * q], w = planes[c * PL + 2 * q + 1];
				E.v[i][0][c] = u.x; E.v[i][1][c] = w.x; E.v[i][3][c] = w.y; E.v[i][2][c] = planes[c * PL + s2].y;
			});
		});
	}

	// ---- A, REDFT01: the even coefficients into the layout a_post10 leaves (slots padded as the stages expect them) ----
	static DSP_HD void a_scatter01(CX *planes, int tid, const Orb &xe)
	{
		static_for<0, QR>([&](auto i) {
			int q;
			if (!item<i>(tid, q)) return;
			const int p0 = F::padded(2 * q), p1 = F::padded(2 * q + 1), p2 = (2 * q + 2 < M) ? F::padded(2 * q + 2) : 0;
			static_for<0, C>([&](auto c) {
				planes[c * PL + p0].x = xe.v[i][0][c];
				planes[c * PL + p1] = cmk<Re>(xe.v[i][1][c], xe.v[i][3][c]);
				planes[c * PL + p2].y = xe.v[i][2][c];
			});
		});
	}
	// RowSpecG::phase<KIND_REDFT01, 0>'s arithmetic on the slots themselves (in place: item k reads and writes slots k and M-k)
	static DSP_HD void a_pre01(const CX *tab, CX *planes, int tid)
	{
		static_for<0, KR>([&](auto i) {
			int k;
			if (!kitem<i>(tid, k)) return;
			const CX tk = tab[OFF_T + k];
			const CX tlk = cmul(cconj(tk), cmk<Re>((Re)0.70710678118654752440, (Re)-0.70710678118654752440));
			const CX t1 = csqr(csqr(tk));
			const int pk = F::padded(k), pm = F::padded(k ? M - k : 0);
			static_for<0, C>([&](auto c) {
				const CX A = planes[c * PL + pk], B = planes[c * PL + pm];
				const Re xk = A.x, xnk = k ? A.y : (Re)0, xlk = k ? B.x : A.y, xlpk = k ? B.y : A.y;
				const CX Vk = cmulc(cmk<Re>(xk, -xnk), tk);
				const CX Vm = cmulc(cmk<Re>(xlk, -xlpk), tlk);
				const CX S = cadd(Vk, cconj(Vm)), D = csub(Vk, cconj(Vm));
				const CX Q = cmul_pi(cmulc(D, t1));
				planes[c * PL + pk] = cconj(cadd(S, Q));
				if (k > 0) planes[c * PL + pm] = csub(S, Q);
			});
		});
	}
	// A's output samples e[m] of the orbit, from the natural-order plane (RowSpecG::phase<KIND_REDFT01, NS + 2>'s `value`): reordered index
	// q, H-1-q, M-1-q, M+q; odd indices carry a minus sign
	static DSP_HD void a_gather01(const CX *planes, int tid, Orb &E)
	{
		const float *pf = reinterpret_cast<const float *>(planes);
		static_for<0, QR>([&](auto i) {
			int q;
			if (!item<i>(tid, q)) return;
			const int n[4] = {q, H - 1 - q, M - 1 - q, M + q};
			static_for<0, 4>([&](auto j) { static_for<0, C>([&](auto c) { const float f = pf[c * (2 * PL) + n[j]]; E.v[i][j][c] = (n[j] & 1) ? -f : f; }); });
		});
	}

	// ---- B: the DCT-IV of d (or of the odd coefficients) ----
	static DSP_HD void b_pre(const CX *tab, CX *planes, int tid, const Orb &d)
	{
		static_for<0, QR>([&](auto i) {
			int q;
			if (!item<i>(tid, q)) return;
			const CX w0 = tab[OFF_OM + q], w1 = tab[OFF_OM + M - 1 - q];
			const int p0 = F::padded(q), p1 = F::padded(M - 1 - q);
			static_for<0, C>([&](auto c) {
				planes[c * PL + p0] = cmul(cmk<Re>(d.v[i][0][c], d.v[i][3][c]), w0);          // n = q:      d[2q]     + i d[H-1-2q]
				planes[c * PL + p1] = cmul(cmk<Re>(d.v[i][2][c], d.v[i][1][c]), w1);          // n = M-1-q:  d[H-2-2q] + i d[2q+1]
			});
		});
	}
	// B's outputs of the orbit from the natural-order plane, times f: O[j] = y[m_j]
	static DSP_HD void b_post(const CX *tab, const CX *planes, int q, float f, float (*O)[C_])
	{
		const CX r0 = tab[OFF_RH + q], r1 = tab[OFF_RH + M - 1 - q];
		static_for<0, C>([&](auto c) {
			const CX u = cmul(planes[c * PL + q], r0), w = cmul(planes[c * PL + M - 1 - q], r1);
			O[0][c] = u.x * f; O[3][c] = -u.y * f; O[2][c] = w.x * f; O[1][c] = -w.y * f;
		});
	}
	// REDFT10's closing: even coefficients (held) and odd ones interleaved into pixels 4q .. 4q+3 and N-4-4q .. N-1-4q
	static DSP_HD void b_store10(const PA &a, const CX *tab, const CX *planes, long long bout, int tid, const Orb &E)
	{
		static_for<0, QR>([&](auto i) {
			int q;
			if (!item<i>(tid, q)) return;
			float O[4][C_];
			b_post(tab, planes, q, 2 * a.scale, O);
			float lo[4 * C], hi[4 * C];
			static_for<0, C>([&](auto c) {
				lo[0 * C + c] = E.v[i][0][c]; lo[1 * C + c] = O[0][c]; lo[2 * C + c] = E.v[i][1][c]; lo[3 * C + c] = O[1][c];
				hi[0 * C + c] = E.v[i][2][c]; hi[1 * C + c] = O[2][c]; hi[2 * C + c] = E.v[i][3][c]; hi[3 * C + c] = O[3][c];
			});
			st4(a.out + bout + (long long)(4 * q) * C, lo);
			st4(a.out + bout + (long long)(N - 4 - 4 * q) * C, hi);
		});
	}
	// REDFT01's closing: x[m] = e[m] + o[m], x[N-1-m] = e[m] - o[m] as the four pixel pairs load10 reads; `old`: the values to add to
	// (sum += image, scan/scan.c:451-459), fetched by fetch_old before the plane is read
	struct Old { Pix<C_, float> v[QR][8]; };
	static DSP_HD void fetch_old(const PA &a, long long bout, int tid, Old &old)
	{
		static_for<0, QR>([&](auto i) {
			int q;
			if (!item<i>(tid, q)) return;
			const int p[8] = {2 * q, 2 * q + 1, N - 2 - 2 * q, N - 1 - 2 * q, H - 2 - 2 * q, H - 1 - 2 * q, H + 2 * q, H + 1 + 2 * q};
			static_for<0, 8>([&](auto e) { old.v[i][e] = load_pix<C, float>(a.out + bout + (long long)p[e] * C); });
		});
	}
	template <bool ACC>
	static DSP_HD void b_store01(const PA &a, const CX *tab, const CX *planes, long long bout, int tid, const Orb &E, const Old &old)
	{
		static_for<0, QR>([&](auto i) {
			int q;
			if (!item<i>(tid, q)) return;
			float O[4][C_];
			b_post(tab, planes, q, 2.f, O);
			const Re sc = a.scale;
			// pixel order of fetch_old: x[m0], x[m1], x[N-1-m1], x[N-1-m0], x[m2], x[m3], x[N-1-m3], x[N-1-m2]
			Pix<C, float> o[8];
			static_for<0, C>([&](auto c) {
				o[0].v[c] = (E.v[i][0][c] + O[0][c]) * sc; o[1].v[c] = (E.v[i][1][c] + O[1][c]) * sc;
				o[2].v[c] = (E.v[i][1][c] - O[1][c]) * sc; o[3].v[c] = (E.v[i][0][c] - O[0][c]) * sc;
				o[4].v[c] = (E.v[i][2][c] + O[2][c]) * sc; o[5].v[c] = (E.v[i][3][c] + O[3][c]) * sc;
				o[6].v[c] = (E.v[i][3][c] - O[3][c]) * sc; o[7].v[c] = (E.v[i][2][c] - O[2][c]) * sc;
				if (q == 0) o[0].v[c] *= a.out_scale0;
			});
			const int p[8] = {2 * q, 2 * q + 1, N - 2 - 2 * q, N - 1 - 2 * q, H - 2 - 2 * q, H - 1 - 2 * q, H + 2 * q, H + 1 + 2 * q};
			static_for<0, 8>([&](auto e) {
				if constexpr (ACC) static_for<0, C>([&](auto c) { o[e].v[c] += old.v[i][e].v[c]; });
				store_pix<C, float>(a.out + bout + (long long)p[e] * C, o[e]);
			});
		});
	}
};

}  // namespace dspfft
