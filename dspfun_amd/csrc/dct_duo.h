// dct_duo.h -- ROW passes that carry TWO real signals of a line through ONE set of butterfly phases.
//
// The row kernels of dct_spec.h are bound by vector-ALU issue on long lines (7680 x 3: ~1350 VALU instructions per wave and
// transform, two thirds of them the butterflies' float arithmetic; the LDS and memory phases of the single resident workgroup
// run beside it, not instead of it).  Where a line needs two transforms of the same length -- zoom's x stage by fast transforms
// (cosine and sine part, zoom_fft.hip), the two lines of a row pair (outer radix-2 split) -- the two signals ride through the FFT as
// the halves of a Pk2 (packed FP32: every butterfly instruction is a v_pk_*_f32), i.e. ONE pass at half the instruction count instead
// of two passes through the same LDS.  A slot is then 16 bytes (re_a, re_b, im_a, im_b), so a 7680-sample signal pair needs 61 KB:
// the channels of an interleaved line go through the plane one after another and the line's outputs wait in registers, two
// workgroups per CU.
//
// The N/2-point complex FFT of the plane is the column kernels' (ColSpecT with one lane vector per row: NP = 1): stages,
// digit-reversed last stage, padding, all shared.
#pragma once
#include "dct_spec.h"

namespace dspfft {

template <int N_, int T_, int... Rs>
struct RowDuoT {
	typedef ColSpecT<float, N_ / 2, 4, T_, Rs...> B;
	typedef typename B::V V;         // LDS slot: (re_a, re_b, im_a, im_b)
	typedef typename B::LC LC;       // the slot as one complex number over Pk2
	static constexpr int N = N_, L = N_ / 2, T = T_, NS = B::NS;
	static constexpr size_t LDS = B::LDS;
	static constexpr int K_ROUNDS = (L / 2 + 1 + T - 1) / T;    // (k, L - k) slot pairs per thread
	static constexpr int X_ROUNDS = (N + T - 1) / T;            // samples of a signal per thread
	static constexpr int KT = L / 2 + 1;                        // entries of a per-k table
	static_assert(N_ % 4 == 0, "duo rows need N divisible by 4");
	struct Regs { LC x[B::LAST_ROUNDS * B::RL]; };
	static DSP_HD int padded(int p) { return B::padded(p); }
	// phases 1 .. NS + 1: the L-point forward FFT of the plane, in place (a barrier after each); w.W = exp(-2 pi i t / L)
	template <int PH> static DSP_HD void fft_phase(const PassArgs &w, V *buf, int tid, Regs &r)
	{
		if constexpr (PH < NS) B::template stage<PH - 1>(w, buf, tid);
		else if constexpr (PH == NS) B::last_read(buf, r, tid);
		else B::last_write(buf, r, tid);
	}
	// after the last phase the plane is in natural order: element n of the two length-N real sequences z[m] = v[2m] + i v[2m+1]
	static DSP_HD Pk2 sample(const V *buf, int n) { return reinterpret_cast<const Pk2 *>(buf)[n]; }
};

// =================================================================================================
// zoom's x stage (zoom/zoom.c:361-368 with the basis of :36-68 on a DCT-III grid, see zoom_fft.hip):
//     out[j][b][c] = sum_{u < cw} g_u in[j][u][c] cos(u (pi (b + 1/2) / M + theta)),   b < vw <= M,  g_0 = scale / 2, g_u = scale
// = yc[b] - ys[b] with yc / ys the cosine / sine series of in[u] cos(theta u) / in[u] sin(theta u).  In Makhoul's order
// (v[j] = y[2j], v[M-1-j] = y[2j+1]) both are real inverse DFTs of length M -- of the Hermitian resp. anti-Hermitian extension of
// V[u] = in[u] e^{i theta' u}-weighted spectra -- and each of those is an M/2-point complex FFT.  Every FFT input slot is a fixed
// complex multiple of at most four input pixels:
//     slot[k], slot[L - k]  <-  in[k], in[L - k], in[L + k], in[M - k]        (L = M / 2, k <= L / 2)
// with multipliers that depend on (k, theta, scale) only: a table the caller fills per frame (zoomx_table_entry, evaluated in double).
// cw <= L / 2 (scales >= 4: BASELINE config 3) needs the first source only, cw <= L the first two.
struct ZoomXArgs {
	const float *in;       // lines of cw pixels
	float *out;            // lines of vw pixels
	const float *tab;      // [NSRC][2][KT] slots of four floats: multiplier of source s for slot k (0) and slot L - k (1), as (re_a, re_b, im_a, im_b)
	const cf *W;           // exp(-2 pi i t / L), t < L
	long long in_pitch, out_pitch;    // floats between lines
	int cw, vw, lines;
};

// (lo, hi) = what a unit sample at pixel n contributes to slots k and L - k of the (cosine, sine) pair; everything in double.
// ca = g_n cos(theta n), sb = g_n sin(theta n) (0 for n >= cw).  Written for clarity, not speed: it runs once per frame and table entry.
struct ZoomXEntry { double lo[4], hi[4]; };    // (re_a, re_b, im_a, im_b)
// bin q of the length-M spectrum X enters Z[q mod L] as X (1 + i w^k) for q < L and X (1 - i w^k) for q >= L, w = e^{2 pi i / M}
// (v[2m] + i v[2m+1] = IDFT_L(Z)[m]); the plane holds conj(Z): the forward FFT of the conjugate is the conjugate of the inverse FFT
DSP_HD void zoomx_accumulate(double *slot, int M, int k, int bin, double xr_a, double xi_a, double xr_b, double xi_b)
{
	const int L = M / 2;
	const double pi = 3.14159265358979323846264338327950288;
	const double wr = cos(2 * pi * k / M), wi = sin(2 * pi * k / M);
	const double s = bin < L ? 1.0 : -1.0;
	const double fr = 1.0 - s * wi, fi = s * wr;          // 1 +- i w^k
	slot[0] += xr_a * fr - xi_a * fi; slot[2] -= xr_a * fi + xi_a * fr;
	slot[1] += xr_b * fr - xi_b * fi; slot[3] -= xr_b * fi + xi_b * fr;
}
DSP_HD ZoomXEntry zoomx_table_entry(int M, int cw, int k, int src, double theta, double scale)
{
	ZoomXEntry e;
	for (int i = 0; i < 4; i++) e.lo[i] = e.hi[i] = 0.0;
	const int L = M / 2;
	const double pi = 3.14159265358979323846264338327950288;
	const int n = src == 0 ? k : src == 1 ? L - k : src == 2 ? L + k : M - k;
	// the four sources coincide in pairs at k = 0 (L - k = L + k; M - k is out of range) and at k = L / 2: each pixel counts once
	if ((src == 2 && k == 0) || (src == 3 && (k == 0 || 2 * k == L)) || (src == 1 && 2 * k == L)) return e;
	if (n >= cw || n >= M) return e;
	const double g = (n == 0 ? 0.5 : 1.0) * scale, ca = g * cos(theta * n), sb = g * sin(theta * n);
	const double er = cos(pi * n / (2.0 * M)), ei = sin(pi * n / (2.0 * M));     // V = (ca | sb) e^{i pi n / 2M}
	// cosine series: vc = Re IDFT(Vc): X[n] += V / 2, X[M - n] += conj(V) / 2 (X[0] = V).  Sine series: vs = Im IDFT(Vs):
	// X[n] += V / (2i), X[M - n] -= conj(V) / (2i).  (On the odd outputs v[M-1-j] the sine series comes out negated: the closing phase adds
	// there instead of subtracting.)
	for (int which = 0; which < 2; which++) {
		const int slot_k = which == 0 ? k : L - k;
		if (which == 1 && (k == 0 || 2 * k == L)) continue;       // slot L does not exist; slot L / 2 is `lo`
		for (int term = 0; term < 2; term++) {
			if (n == 0 && term == 1) continue;
			const int q = term ? M - n : n;
			if (q % L != slot_k % L) continue;
			double xr_a, xi_a, xr_b, xi_b;
			if (n == 0) { xr_a = ca; xi_a = 0; xr_b = 0; xi_b = 0; }
			else if (term == 0) { xr_a = ca * er / 2; xi_a = ca * ei / 2; xr_b = sb * ei / 2; xi_b = -sb * er / 2; }
			else { xr_a = ca * er / 2; xi_a = -ca * ei / 2; xr_b = sb * ei / 2; xi_b = sb * er / 2; }
			zoomx_accumulate(which == 0 ? e.lo : e.hi, M, slot_k, q, xr_a, xi_a, xr_b, xi_b);
		}
	}
	return e;
}

// =================================================================================================
// THREE barrier-separated phases per channel.  (The first cut kept the row kernels' NS + 3 -- phase 0 writing the slots, the stages, the
// natural-order write, a closing phase reading the samples back: 334-346 us at BASELINE config 3, no better than the two-pass kernel it
// was to replace; with two 4-wave workgroups per CU what costs is the head of every phase -- phase 0 and the closing phase alone, no
// butterflies, took 145 of the 341 us.  profiles/r04_zoomx_variants.txt)
//   A  the first DIF stage reads its R0 slots straight from global memory (slot = table x pixel, as phase 0 computed it) -- no phase 0,
//      no LDS round trip in front of the first butterfly;
//   B  the middle stages, in place (ColSpecT::stage);
//   C  the last stage keeps its outputs in registers: thread kb holds FFT outputs j = kb + NBL r.  Output j carries samples x = 4j, 4j + 2
//      (j < L/2) resp. 4j' + 3, 4j' + 1 (j' = L - 1 - j) -- of every channel, so the thread owns those pixels whole and keeps the earlier
//      channels' samples in registers until the last channel stores 12-byte pixels.  The partner slot L - 1 - j = (NBL - 1 - kb) + NBL (RL - 1 - r)
//      belongs to thread NBL - 1 - kb, which the lane map puts on the mirror lane (63 - lane) of the same wave: at store step i lanes < 32
//      bring slot i and lanes >= 32 slot RL - 1 - i, so one store instruction writes 24 contiguous bytes of every 48 over one run of 32 x 48 bytes.
//      (Round 4 exchanged im across the mirror lanes to hold pixel PAIRS: profiles/r05_isa_zoomx.txt.)
// Needs T = NBL = L / RL threads (one last-stage butterfly each), T a multiple of 64.
// item i < nsrc (M/4 + 1) of the per-slot table tab[q][s] (q < nsrc sources, s < M/2 slots, four floats each): the entries of slot pair
// (k, L - k) for source q.  Every slot is written (slot L/2 by k = L/2, slot 0 by k = 0), so the table needs no clearing between frames.
DSP_HD void zoomx_table_item(float *tab, int M, int cw, int nsrc, double theta, double scale, int i)
{
	const int L = M / 2, KT = M / 4 + 1;
	if (i >= nsrc * KT) return;
	const int q = i / KT, k = i - q * KT;
	const ZoomXEntry e = zoomx_table_entry(M, cw, k, q, theta, scale);
	for (int j = 0; j < 4; j++) {
		tab[((size_t)q * L + k) * 4 + j] = (float)e.lo[j];
		if (k > 0 && 2 * k != L) tab[((size_t)q * L + (L - k)) * 4 + j] = (float)e.hi[j];
	}
}

// CLIP: the viewport is narrower than the scaled line (vw < N): every pixel store is tested (the usual frame stores the whole line)
template <class S, int C, int NSRC, bool CLIP = true>
struct ZoomXLeanT {
	typedef typename S::B B;
	typedef typename S::V V;
	typedef typename S::LC LC;
	static constexpr int N = S::N, L = S::L, T = S::T, NS = S::NS;
	static constexpr int R0 = B::R0, RL = B::RL, SB = B::SB, NBL = B::NBL;
	static_assert(NS >= 2 && NBL == T && T % 64 == 0, "one last-stage butterfly per thread");
	static_assert(RL % 2 == 1, "odd last radix (conflict-free gather; the middle slot index pairs with itself)");
	static constexpr int A_ROUNDS = (SB + T - 1) / T;
	struct State { float hold[(C > 1 ? C - 1 : 1) * 2 * RL]; };      // per store step, the two samples of the channels before the last
	struct Ex { float p[RL], s[RL]; };                                 // per slot r: the samples of its two pixels, the lower pixel first

	// source pixel q of slot s: k = min(s, L - s); q = 0: k, 1: L - k, 2: L + k, 3: N - k
	static DSP_HD int src_pixel(int q, int s) { const int k = s < L - s ? s : L - s; return q == 0 ? k : q == 1 ? L - k : q == 2 ? L + k : N - k; }

	// phase A: slots s = m + r SB from global memory, first DIF butterfly + twiddles, into the padded plane
	static DSP_HD void phase_a(const ZoomXArgs &a, const PassArgs &w, V *buf, long long bin, int ch, int tid)
	{
		const V *tab = reinterpret_cast<const V *>(a.tab);
		const float *line = a.in + bin + ch;
		static_for<0, A_ROUNDS>([&](auto i) {
			const int m = tid + i * T;
			if ((i + 1) * T <= SB || m < SB) {
				LC x[R0];
				static_for<0, R0>([&](auto r) {
					const int s = m + r * SB;
					LC v = B::lzero();
					static_for<0, NSRC>([&](auto q) {
						int n = src_pixel(q, s);
						n = n < a.cw ? n : a.cw - 1;                       // beyond the window the multiplier is zero: any valid pixel will do
						const float f = line[(unsigned)(n * C)];               // (32-bit offset from a uniform base: no 64-bit address arithmetic per load)
						const LC t = B::l_get(tab[(unsigned)(q * L + s)]);
						v = cmk<Pk2>(v.x + t.x * f, v.y + t.y * f);
					});
					x[r] = v;
				});
				Dft<R0>::run(x);
				cf tw[R0];
				tw[1] = w.W[m];
				static_for<2, R0>([&](auto r) { if constexpr (r % 2 == 0) tw[r] = csqr(tw[r / 2]); else tw[r] = cmul(tw[r / 2], tw[r - r / 2]); });
				static_for<1, R0>([&](auto r) { x[r] = B::lmul(x[r], tw[r]); });
				static_for<0, R0>([&](auto r) { buf[m + r * (SB + B::PADC)] = B::l_put(x[r]); });
			}
		});
	}
	// The same with the line's pixels loaded WHOLE, once (NSRC == 1, one round: BASELINE config 3): a 12-byte load per slot serves the three channels,
	// where phase_a issues a 4-byte load from the same pixel in each of them and waits for it again (round 5: phase A was the longest of a
	// channel's three phases, 7.8K / 7.1K / 5.4K clocks of a workgroup's 52K, for 2K clocks of vector instructions: tools/kstamp zoomx)
	static constexpr bool WHOLE = NSRC == 1 && A_ROUNDS == 1;
	struct Pixels { Pix<C, float> p[R0]; };
	static DSP_HD void load_pixels(const ZoomXArgs &a, long long bin, int tid, Pixels &px)
	{
		const float *line = a.in + bin;
		const int m = tid < SB ? tid : SB - 1;                      // (threads beyond the butterflies load a valid pixel and drop it)
		int mc = m * C;
		DSP_PIN1(mc);                                               // (one per-thread element offset + a constant per slot, not a multiplication per load)
		const int last = (a.cw - 1) * C;
		static_for<0, R0>([&](auto r) {
			// pixel of slot s = m + r SB: min(s, L - s), on which side of L/2 the slot lies known per r wherever SB divides L/2
			int e;
			if constexpr ((r + 1) * SB - 1 <= L / 2) e = mc + r * SB * C;
			else if constexpr (r * SB >= L / 2) e = (L - r * SB) * C - mc;
			else { const int s = m + r * SB; e = (s < L - s ? s : L - s) * C; }
			e = e < last ? e : last;                                // beyond the window the multiplier is zero: any valid pixel will do
			px.p[r] = load_pix<C, float>(at(line, e));
		});
	}
	template <int CH>
	static DSP_HD void phase_a_held(const ZoomXArgs &a, const PassArgs &w, V *buf, int tid, const Pixels &px)
	{
		const int m = tid;
		if (m < SB) {
			LC x[R0];
			static_for<0, R0>([&](auto r) {
				const LC t = B::l_get(*reinterpret_cast<const V *>(reinterpret_cast<const char *>(a.tab) + (unsigned)((m + r * SB) * (int)sizeof(V))));
				const float f = px.p[r].v[CH];
				x[r] = cmk<Pk2>(t.x * f, t.y * f);
			});
			Dft<R0>::run(x);
			cf tw[R0];
			tw[1] = w.W[m];
			static_for<2, R0>([&](auto r) { if constexpr (r % 2 == 0) tw[r] = csqr(tw[r / 2]); else tw[r] = cmul(tw[r / 2], tw[r - r / 2]); });
			static_for<1, R0>([&](auto r) { x[r] = B::lmul(x[r], tw[r]); });
			static_for<0, R0>([&](auto r) { buf[m + r * (SB + B::PADC)] = B::l_put(x[r]); });
		}
	}
	// phases B: stage I = 1 .. NS - 2
	template <int I> static DSP_HD void phase_b(const PassArgs &w, V *buf, int tid) { B::template stage<I>(w, buf, tid); }

	// element e of a line by a 32-bit BYTE offset from the (uniform) line base: the address needs no 64-bit vector arithmetic
	static DSP_HD float *at(float *line, int e) { return reinterpret_cast<float *>(reinterpret_cast<char *>(line) + (unsigned)(e * 4)); }
	static DSP_HD const float *at(const float *line, int e) { return reinterpret_cast<const float *>(reinterpret_cast<const char *>(line) + (unsigned)(e * 4)); }
	// thread -> last-stage butterfly: partners kb, NBL - 1 - kb on mirror lanes of one wave
	static DSP_HD int kb_of(int tid) { const int wv = tid >> 6, l = tid & 63; return l < 32 ? 32 * wv + l : NBL - 1 - (32 * wv + 63 - l); }
	// Output j = kb + NBL r of the last butterfly holds plane elements n = 2j (re), 2j + 1 (im): v[n] for even n, -v[n] for odd n.  Of the four
	// pixels 4g .. 4g + 3 of group g, output j = g (j < L/2, `low`) holds 4g (re) and 4g + 2 (im), output j = L - 1 - g holds 4g + 3 (re) and
	// 4g + 1 (im) -- of every channel, so a thread owns whole pixels and nothing needs to change hands.  `low` is known at compile time except
	// for the middle slot (RL odd), where it is the half of the wave the lane sits in.
	// (Round 4 traded im with the partner thread on the mirror lane so that each thread held the pixel PAIRS 4g, 4g + 1 / 4g + 2, 4g + 3, and ran
	// lanes >= 32 through their slots backwards to make partners meet: 15 ds_bpermute and 75 selects per channel for nothing -- pixels are
	// stored one 12-byte pixel per lane either way.)
	static_assert(RL == 1 || L % (2 * NBL) == NBL, "the middle slot's outputs split between the low and the high half at kb = NBL / 2");
	template <int R> static DSP_HD bool low_of(int tid) { if constexpr (2 * R + 1 < RL) return true; else if constexpr (2 * R + 1 > RL) return false; else return (tid & 63) < 32; }
	// phase C, first half: last butterfly from the plane, the two series combined: per slot r its two pixels' samples in ascending order
	// (p: 4g or 4g + 1, s: 4g + 2 or 4g + 3)
	static DSP_HD void phase_c(const V *buf, int tid, Ex &e)
	{
		const int kb = kb_of(tid);
		const V *p = buf + B::padded(B::last_blk(kb) * RL);
		LC x[RL];
		static_for<0, RL>([&](auto r) { x[r] = B::l_get(p[r]); });
		Dft<RL>::run(x);
		static_for<0, RL>([&](auto r) {
			// even samples: cosine minus sine series (j < L/2); odd samples: cosine plus sine
			if constexpr (2 * r + 1 == RL) {
				const float sg = low_of<r>(tid) ? -1.f : 1.f;
				const float re = x[r].x.x + sg * x[r].x.y, im = -(x[r].y.x + sg * x[r].y.y);
				e.p[r] = low_of<r>(tid) ? re : im; e.s[r] = low_of<r>(tid) ? im : re;
			} else if constexpr (2 * r + 1 < RL) { e.p[r] = x[r].x.x - x[r].x.y; e.s[r] = -(x[r].y.x - x[r].y.y); }
			else { e.s[r] = x[r].x.x + x[r].x.y; e.p[r] = -(x[r].y.x + x[r].y.y); }
		});
	}
	// Store step i: lanes < 32 bring their slot i, lanes >= 32 their slot RL - 1 - i -- the other two pixels of the same 32 groups, so that one
	// store instruction of the wave writes 24 contiguous bytes of every 48 over one 1.5 KB stretch of the line (each half of the wave by its own
	// slot order touches twice the cache lines per instruction: measured 2.1x the time of the closing phase).  x0: the pixel of y0; y1 goes to x0 + 2
	template <int I> static DSP_HD void step_of(int tid, const Ex &e, float &y0, float &y1)
	{
		const bool up = (tid & 63) >= 32;
		y0 = up ? e.p[RL - 1 - I] : e.p[I]; y1 = up ? e.s[RL - 1 - I] : e.s[I];
	}
	// the pixels of the store steps: x0 of step I = (2 I + 1 <= RL ? lo + 4 NBL I : hi + 4 NBL (RL - 1 - I)), and x0 + 2.  Two per-thread bases, pinned
	// in registers (left alone, the compiler rebuilds them from the thread index before every store: 8 instructions, one a quarter-rate multiply)
	struct Bases { int lo, hi; };
	static DSP_HD Bases bases_of(int tid)
	{
		const bool up = (tid & 63) >= 32;
		const int kb = kb_of(tid), g0 = up ? NBL - 1 - kb : kb;
		Bases b;
		b.lo = 4 * g0 + (up ? 1 : 0); b.hi = 4 * (NBL - 1 - g0) + (up ? 0 : 1);
		return b;
	}
	// phase C, second half.  Channels before the last keep their samples; the last stores whole pixels
	static DSP_HD void phase_c_emit(const ZoomXArgs &a, long long bout, int ch, int tid, const Ex &e, State &st)
	{
		if (ch + 1 < C) {
			static_for<0, RL>([&](auto i) {
				float y0, y1;
				step_of<i>(tid, e, y0, y1);
				static_for<0, C - 1>([&](auto c) {
					float &h0 = st.hold[(c * RL + i) * 2], &h1 = st.hold[(c * RL + i) * 2 + 1];
					h0 = (ch == c) ? y0 : h0; h1 = (ch == c) ? y1 : h1;
				});
			});
			return;
		}
		float *line = a.out + bout;
		const Bases b = bases_of(tid);
		int elo = b.lo * C, ehi = b.hi * C;                  // element offsets; the clip test compares them with vw C
		DSP_PIN1(elo); DSP_PIN1(ehi);
		const int ve = a.vw * C;
		static_for<0, RL>([&](auto i) {
			float y0, y1;
			step_of<i>(tid, e, y0, y1);
			const int e0 = 2 * i + 1 <= RL ? elo + 4 * NBL * i * C : ehi + 4 * NBL * (RL - 1 - i) * C;
			Pix<C, float> o0, o1;
			// (handed over one by one: read as a Pix straight from the array, the compiler uses a vector load that keeps the array in scratch)
			static_for<0, C - 1>([&](auto c) { float v0 = st.hold[(c * RL + i) * 2], v1 = st.hold[(c * RL + i) * 2 + 1]; DSP_PIN1(v0); DSP_PIN1(v1); o0.v[c] = v0; o1.v[c] = v1; });
			o0.v[C - 1] = y0; o1.v[C - 1] = y1;
			if (!CLIP || e0 < ve) store_pix<C, float>(at(line, e0), o0);
			if (!CLIP || e0 + 2 * C < ve) store_pix<C, float>(at(line, e0 + 2 * C), o1);
		});
	}
	// the same with the channel known at compile time (WHOLE): no selects on the held samples
	template <int CH>
	static DSP_HD void phase_c_emit_ch(const ZoomXArgs &a, long long bout, int tid, const Ex &e, State &st)
	{
		float *line = a.out + bout;
		int elo = 0, ehi = 0;
		if constexpr (CH + 1 == C) { const Bases b = bases_of(tid); elo = b.lo * C; ehi = b.hi * C; DSP_PIN1(elo); DSP_PIN1(ehi); }
		const int ve = a.vw * C;
		static_for<0, RL>([&](auto i) {
			float y0, y1;
			step_of<i>(tid, e, y0, y1);
			if constexpr (CH + 1 < C) {
				DSP_PIN1(y0); DSP_PIN1(y1);            // (computed NOW: left alone, the compiler keeps the butterfly's inputs instead and sinks it to the stores)
				st.hold[(CH * RL + i) * 2] = y0; st.hold[(CH * RL + i) * 2 + 1] = y1;
			} else {
				const int e0 = 2 * i + 1 <= RL ? elo + 4 * NBL * i * C : ehi + 4 * NBL * (RL - 1 - i) * C;
				Pix<C, float> o0, o1;
				static_for<0, C - 1>([&](auto c) { o0.v[c] = st.hold[(c * RL + i) * 2]; o1.v[c] = st.hold[(c * RL + i) * 2 + 1]; });
				o0.v[C - 1] = y0; o1.v[C - 1] = y1;
				if (!CLIP || e0 < ve) store_pix<C, float>(at(line, e0), o0);
				if (!CLIP || e0 + 2 * C < ve) store_pix<C, float>(at(line, e0 + 2 * C), o1);
			}
		});
	}
};

}  // namespace dspfft
