// dct_spec.h -- compile-time-specialised ROW / COL passes for the hot sizes (same mathematics as
// dct_core.h; geometry, radices, thread count and trip counts are template parameters so every
// index computation constant-folds and every loop unrolls).
//
// Differences from the generic passes that matter for MI355X:
//   ROW  global I/O goes straight between registers and LDS in PIXEL order (one pixel = C floats per
//        lane, e.g. global_load_dwordx3 for the image tools' RGB buffers: 64 lanes x 12 B = 768 B
//        contiguous per wave instruction); there is no raw staging copy of the line, so a
//        3840x3 line needs 46 KB of LDS instead of 92 KB and three workgroups fit on a CU.
//   COL  each lane moves 16 bytes = four adjacent float columns = two complex signals (double: two columns, one signal) and
//        carries the pair through the butterflies as ONE complex number over Pk2 (packed FP32: v_pk_fma_f32 ...); LDS rows
//        are [n][K/2] complex; the last radix is odd so the final stage's strided ds_read_b128 are bank-conflict free;
//        workgroup -> tile mapping is XCD-aware (tiles that share 128-B lines run on the same XCD's L2).
// Everything is a template over the sample type: float (the image tools) and double (the fftw_ API of spec / zoom's default build).
// Phases are barrier-separated and numbered 0 .. NS+1 (load/pre, NS FFT stages, post/unpack).
#pragma once
#include "dct_core.h"
#include "elementwise_core.h"
#include "spec_list.h"

// LDS pad (elements / rows per first-stage sub-block); overridable for experiments (tools/sbench.hip)
#ifndef DSP_ROW_PADC
#define DSP_ROW_PADC 1
#endif
#ifndef DSP_COL_PADC
#define DSP_COL_PADC 1
#endif

namespace dspfft {

template <int I, int... Rs> constexpr int pack_get() { constexpr int a[] = {Rs..., 1}; return a[I]; }
// product of radices I.. (the length of the sub-transforms stage I splits)
template <int I, int... Rs> constexpr int pack_lc() { constexpr int a[] = {Rs..., 1}; int l = 1; for (int j = I; j < (int)sizeof...(Rs); j++) l *= a[j]; return l; }

// slot of FFT output k after the in-place DIF stages (mixed-radix digit reversal)
template <int SPAN, int... Rs> struct PosCalc;
template <int SPAN> struct PosCalc<SPAN> { static DSP_HD int run(int) { return 0; } };
template <int SPAN, int R0, int... Rest>
struct PosCalc<SPAN, R0, Rest...> {
	static DSP_HD int run(int k) { constexpr int span = SPAN / R0; return (k % R0) * span + PosCalc<span, Rest...>::run(k / R0); }
};

// digit reversal over only the first CNT radices of the pack (SPAN = product of those radices)
template <int SPAN, int CNT, int R0, int... Rest>
struct PosCalcFirst {
	static DSP_HD int run(int k)
	{
		if constexpr (CNT == 0) return 0;
		else { constexpr int span = SPAN / R0; return (k % R0) * span + PosCalcFirst<span, CNT - 1, Rest..., 1>::run(k / R0); }
	}
};

// strided work loop with a compile-time trip count
template <int TOTAL, int T, class F>
DSP_HD void tloop(int tid, F &&f)
{
	constexpr int FULL = TOTAL / T, REM = TOTAL % T;
	static_for<0, FULL>([&](auto i) { f(tid + i * T); });
	if constexpr (REM > 0) { if (tid < REM) f(tid + FULL * T); }
}

template <class Re> DSP_HD cx<Re> csqr(cx<Re> a)
{
#if defined(DSP_PK_CX)
	if constexpr (std::is_same<Re, float>::value) return cmul(a, a);      // two packed issues for the four scalar ones
	else
#endif
	return cmk<Re>(a.x * a.x - a.y * a.y, (Re)2 * a.x * a.y);
}

// x[r] *= w1^r, r = 1..R-1, powers built by a balanced product tree (depth log2 R) from ONE table
// value, so a butterfly costs one twiddle load instead of R-1.
template <int R, class CX>
DSP_HD void twiddle_chain(CX *x, CX w1)
{
	CX w[R > 1 ? R : 2];
	w[1] = w1;
	static_for<2, R>([&](auto r) { if constexpr (r % 2 == 0) w[r] = csqr(w[r / 2]); else w[r] = cmul(w[r / 2], w[r - r / 2]); });
	static_for<1, R>([&](auto r) { x[r] = cmul(x[r], w[r]); });
}

// the same with the powers built one after another (w^r = w^(r-1) w1): a longer dependency chain but two twiddle values
// live instead of R - 1.  Used where registers are scarcer than latency (row_pair_kernel).
template <int R, class CX>
DSP_HD void twiddle_chain_seq(CX *x, CX w1)
{
	CX w = w1;
	static_for<1, R>([&](auto r) { x[r] = cmul(x[r], w); if constexpr (r + 1 < R) w = cmul(w, w1); });
}

// ---- pixel (C samples) global access ------------------------------------------------------------
// Re = float (the tuned image path) or double (the fftw_ API: spec / zoom's default build)
// What one lane of a column pass moves: 16 bytes = NCS complex signals = 2 NCS adjacent real columns (float: two signals = four
// columns, double: one signal = two columns).  s[i].x is column 2i, s[i].y column 2i + 1: two real columns ride through the complex
// FFT as one signal.  One global_load_dwordx4 / ds_read_b128 either way.
template <class Re, int NCS_> struct alignas(16) SigVec { cx<Re> s[NCS_]; };
template <class Re> struct sig_of { static constexpr int NCS = 16 / (2 * (int)sizeof(Re)); typedef SigVec<Re, NCS> type; };

template <int C, class Re = float> struct Pix { Re v[C]; };

template <int C, class Re> DSP_HD Pix<C, Re> load_pix(const Re *p)
{
	Pix<C, Re> r;
#if defined(__HIP_DEVICE_COMPILE__)
	if constexpr (C == 3) { typedef Re r3 __attribute__((ext_vector_type(3))); r3 t; __builtin_memcpy(&t, p, 3 * sizeof(Re)); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; return r; }
	if constexpr (C == 4) { typedef Re r4 __attribute__((ext_vector_type(4))); const r4 t = *reinterpret_cast<const r4 *>(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r; }
	if constexpr (C == 2) { typedef Re r2 __attribute__((ext_vector_type(2))); const r2 t = *reinterpret_cast<const r2 *>(p); r.v[0] = t.x; r.v[1] = t.y; return r; }
#endif
	static_for<0, C>([&](auto c) { r.v[c] = p[c]; });
	return r;
}
template <int C, class Re> DSP_HD void store_pix(Re *p, const Pix<C, Re> &r)
{
#if defined(__HIP_DEVICE_COMPILE__)
	if constexpr (C == 3) { typedef Re r3 __attribute__((ext_vector_type(3))); r3 t; t.x = r.v[0]; t.y = r.v[1]; t.z = r.v[2]; __builtin_memcpy(p, &t, 3 * sizeof(Re)); return; }
	if constexpr (C == 4) { typedef Re r4 __attribute__((ext_vector_type(4))); r4 t; t.x = r.v[0]; t.y = r.v[1]; t.z = r.v[2]; t.w = r.v[3]; *reinterpret_cast<r4 *>(p) = t; return; }
	if constexpr (C == 2) { typedef Re r2 __attribute__((ext_vector_type(2))); r2 t; t.x = r.v[0]; t.y = r.v[1]; *reinterpret_cast<r2 *>(p) = t; return; }
#endif
	static_for<0, C>([&](auto c) { p[c] = r.v[c]; });
}

// MASKED is decided once per prefetch (a.mask is uniform), not per load: a branch around every load would make each
// load wait for its own data before the next is issued (measured: 8K row pass 208 -> 257 us)
template <int C, bool MASKED, class Re> DSP_HD Pix<C, Re> load_pix_m(const PassArgsT<Re> &a, long long off)
{
	if constexpr (MASKED) {
		// fused scan step: look at the owner ids first and do not fetch coefficients that are masked out
		// (a frame keeps 1/32 of them at BASELINE config 4, so most of the first pass's reads disappear)
		uint32_t id[C];
		bool any = false;
		static_for<0, C>([&](auto c) { id[c] = a.mask[a.mask_div.div((uint32_t)(off + c))]; any = any || id[c] == a.mask_id; });
		Pix<C, Re> v;
		if (!any) { static_for<0, C>([&](auto c) { v.v[c] = (Re)0; }); return v; }
		v = load_pix<C, Re>(a.in + off);
		static_for<0, C>([&](auto c) { if (id[c] != a.mask_id) v.v[c] = (Re)0; });
		return v;
	}
	return load_pix<C, Re>(a.in + off);
}
// ROW pass behind a masked COL pass that skipped its empty tiles (PassGeom::zflags): o = offset of the pixel within its line.
// No branch around the load (see load_pix_m): a pixel whose tiles were all skipped loads from the zero page instead.
template <int C, class Re> DSP_HD Pix<C, Re> load_pix_z(const PassArgsT<Re> &a, const uint8_t *zf, int o, long long off)
{
	const int t0 = o >> a.zshift, tl = (o + C - 1) >> a.zshift;
	const uint8_t f0 = zf[t0], fl = zf[tl];
	// (round 6 tried `if (f0 | fl) load` here, as prefetch01_bits does with its precomputed flags: with the flags themselves still in flight the loads
	// serialise behind them -- planar 8K scan step 350 -> 427 us)
	const Re *p = (f0 | fl) ? a.in + off : reinterpret_cast<const Re *>(a.zpage);
	Pix<C, Re> v = load_pix<C, Re>(p);
	static_for<0, C>([&](auto c) { const uint8_t nz = (((o + c) >> a.zshift) == t0) ? f0 : fl; if (!nz) v.v[c] = (Re)0; });
	return v;
}
template <int C, class Re> DSP_HD void store_pix_a(const PassArgsT<Re> &a, long long off, Pix<C, Re> r)
{
	if (a.accumulate) { const Pix<C, Re> o = load_pix<C, Re>(a.out + off); static_for<0, C>([&](auto c) { r.v[c] += o.v[c]; }); }
	store_pix<C, Re>(a.out + off, r);
}
template <bool MASKED, class Re> DSP_HD typename sig_of<Re>::type loadv_m(const PassArgsT<Re> &a, long long off, bool &hit)
{
	typedef typename sig_of<Re>::type V;
	constexpr int NCS = sig_of<Re>::NCS;
	if constexpr (MASKED) {
		// owner ids of the lane's 2 NCS consecutive elements: when they span at most two owners (three or more elements per owner:
		// RGB pixels) two loads serve all of them
		uint32_t id[2 * NCS], own[2 * NCS];
		static_for<0, 2 * NCS>([&](auto c) { own[c] = a.mask_div.div((uint32_t)off + c); });
		if (own[2 * NCS - 1] - own[0] <= 1) {
			const uint32_t ia = a.mask[own[0]], ib = a.mask[own[2 * NCS - 1]];
			static_for<0, 2 * NCS>([&](auto c) { id[c] = own[c] == own[0] ? ia : ib; });
		} else {
			static_for<0, 2 * NCS>([&](auto c) { id[c] = a.mask[own[c]]; });
		}
		bool any = false;
		static_for<0, 2 * NCS>([&](auto c) { any = any || id[c] == a.mask_id; });
		V v;
		static_for<0, NCS>([&](auto i) { v.s[i].x = v.s[i].y = (Re)0; });
		if (!any) return v;
		hit = true;                  // this thread selected at least one coefficient (sparse scan frames: see PassGeom::zflags)
		v = *reinterpret_cast<const V *>(a.in + off);
		static_for<0, NCS>([&](auto i) { if (id[2 * i] != a.mask_id) v.s[i].x = (Re)0; if (id[2 * i + 1] != a.mask_id) v.s[i].y = (Re)0; });
		return v;
	}
	return *reinterpret_cast<const V *>(a.in + off);
}
// The same from a table the plan prepared (dspfft_plan_scan_prepare, PassGeom::eids): the owner id of every ELEMENT of the image, one or two
// bytes each, in the order the column tiles read them -- [tile][row][K] (rows of a split pass: even rows, then odd rows).  A lane's ids
// are then one 4- or 8-byte load next to its neighbours' (a wave: 256 or 512 contiguous bytes), where the owner-id array itself costs one
// 128-byte line per tile ROW for 22 bytes of it: 6.8 M L2 requests per 8K frame, every one a miss in the vector L1 -- the 8K column pass
// spent 155 of its 200 us on them, whatever the order of the loads and whichever XCD the neighbours ran on (profiles/r04_scan_mask.txt).
// Two steps: all ids of a thread's items, then all coefficients (an item nobody selected is not loaded).
// EB = bytes per id; ids that do not fit (and the DC pixel's "no frame") are stored as all ones and match no frame the table is used for.
template <int EB, class Re> struct MaskIds { uint32_t v[EB == 1 ? 1 : 2 * sig_of<Re>::NCS * EB / 4]; };
template <int EB, class Re> DSP_HD void mask_fetch_ids(const PassArgsT<Re> &a, long long eoff, MaskIds<EB, Re> &m)
{
	constexpr int NW = sizeof(m.v) / 4;
	const uint32_t *p = reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(a.eids) + eoff * EB);
	static_for<0, NW>([&](auto w) { m.v[w] = p[w]; });
}
template <int EB, class Re> DSP_HD typename sig_of<Re>::type mask_select_load(const PassArgsT<Re> &a, long long off, const MaskIds<EB, Re> &m, bool &hit)
{
	typedef typename sig_of<Re>::type V;
	constexpr int NCS = sig_of<Re>::NCS, NE = 2 * NCS;
	const uint32_t want = a.mask_id, ones = EB == 1 ? 0xffu : 0xffffu;
	bool sel[NE];
	bool any = false;
	static_for<0, NE>([&](auto c) {
		constexpr int bit = c * 8 * EB;
		sel[c] = ((m.v[bit / 32] >> (bit % 32)) & ones) == want;
		any = any || sel[c];
	});
	hit = hit || any;
	// an item nobody selected is not loaded at all: the lanes that selected one load under the execution mask, a wave without any skips the instruction
	// (round 6; until then it loaded from the page of zeros: scan step 329.8 -> 327.4 us)
	V v;
	static_for<0, NCS>([&](auto i) { v.s[i].x = v.s[i].y = (Re)0; });
	if (any) v = *reinterpret_cast<const V *>(a.in + off);
	static_for<0, NCS>([&](auto i) { if (!sel[2 * i]) v.s[i].x = (Re)0; if (!sel[2 * i + 1]) v.s[i].y = (Re)0; });
	return v;
}
// NITEM items: off_of(i, off, eoff) -> whether item i exists for this thread, its element offset in the image and in the id table; sink(i, v)
template <int EB, int NITEM, class Re, class OFF, class SINK>
DSP_HD void masked_two_step(const PassArgsT<Re> &a, bool &hit, OFF off_of, SINK sink)
{
	MaskIds<EB, Re> ids[NITEM];
	static_for<0, NITEM>([&](auto i) { long long off, eoff; if (off_of(i, off, eoff)) mask_fetch_ids<EB, Re>(a, eoff, ids[i]); });
	static_for<0, NITEM>([&](auto i) { long long off, eoff; if (off_of(i, off, eoff)) sink(i, mask_select_load<EB, Re>(a, off, ids[i], hit)); });
}
// element (row y, tile t, k) of the id table; rows of a split pass (halves == 2) are stored by parity
DSP_HD long long eid_index(int N, int K, int halves, int t, int y, int k)
{
	const int r = halves == 2 ? (y & 1) * (N / 2) + (y >> 1) : y;
	return ((long long)t * N + r) * K + k;
}

template <class Re> DSP_HD void storev_a(const PassArgsT<Re> &a, long long off, typename sig_of<Re>::type r)
{
	typedef typename sig_of<Re>::type V;
	V *p = reinterpret_cast<V *>(a.out + off);
	if (a.accumulate) { const V o = *p; static_for<0, sig_of<Re>::NCS>([&](auto i) { r.s[i].x += o.s[i].x; r.s[i].y += o.s[i].y; }); }
	*p = r;
}

// 8-bit samples at the ends of motion's pipeline (motion/motion.c:617-640 load, :760-776 store), fused into the
// planar (C = 1) row passes: `in` replaces the float input of a REDFT10 pass, `out` receives
// quantise_u8(value * mul) instead of the float output of a REDFT01 pass.  Offsets are the plan's element offsets.
struct U8IO { const uint8_t *in; uint8_t *out; double mul; };

// =================================================================================================
// GS_ = distance in samples between consecutive pixels of the line in global memory.  GS_ == C_: the line's C_ interleaved channels all
// pass through this workgroup's LDS.  C_ == 1 with GS_ > 1 ("channel lines", RowChanSpecT below): the workgroup transforms ONE channel of
// an interleaved line, a 1/GS_ of the LDS -- 7680 x 3 floats / 3840 x 3 doubles are 92 KB per line, one workgroup per CU, whose load,
// butterfly and store phases then overlap with nothing; a channel line is 31 KB and five workgroups share the CU.
template <class Re_, int N_, int C_, int GS_, int T_, int... Rs>
struct RowSpecG {
	typedef Re_ Re;                                    // sample type
	typedef cx<Re_> CX;
	typedef PassArgsT<Re_> PA;
	static constexpr int N = N_, C = C_, GS = GS_, T = T_, L = N_ / 2, NS = (int)sizeof...(Rs), NPH = NS + 3;
	template <int T2> using with_threads = RowSpecG<Re_, N_, C_, GS_, T2, Rs...>;      // the same line on another workgroup size
	static_assert(GS_ == C_ || C_ == 1, "channel lines hold one channel");
	// min waves per SIMD asked of the register allocator.  double: as many workgroups as the line's LDS allows, capped at 4 (128 VGPRs);
	// left alone the allocator spends 140+ on the double kernels and a second workgroup no longer fits a CU
	static constexpr int WPE_D = (int)((160 * 1024) / ((size_t)C_ * (N_ / 2 + 16) * sizeof(CX))) * T_ / 256;
	static constexpr int WPE = std::is_same<Re, double>::value ? (WPE_D < 1 ? 1 : WPE_D > 4 ? 4 : WPE_D) : 1;
	static_assert((1 * ... * Rs) == L, "radices must multiply to N/2");
	static_assert(N % 2 == 0 && NS >= 1, "ROW needs even N");
	// LDS layout of one channel plane while the DIF stages run: slot p lives at p + (p / SB) * PADC,
	// SB = first-stage sub-block; the pad de-phases the SB-strided accesses of the digit-reversed
	// gather in the last stage.  After the last stage the plane is in natural order (no pad).
	static constexpr int R0 = pack_get<0, Rs...>(), RL = pack_get<NS - 1, Rs...>();
	static constexpr int SB = L / R0, PADC = (NS >= 2) ? DSP_ROW_PADC : 0;
	static constexpr int PL = L + R0 * PADC;           // plane pitch (complex)
	static constexpr size_t LDS = (size_t)C * PL * sizeof(CX);
	static constexpr int NBL = L / RL;                 // butterflies of the last stage per channel
	static constexpr int LAST_ROUNDS = (C * NBL + T - 1) / T;
	static constexpr int PIX_ROUNDS = (N + T - 1) / T;           // REDFT10: pixels per thread
	static constexpr int K_ROUNDS = (L / 2 + 1 + T - 1) / T;     // REDFT01: (k, L-k) pairs per thread
	// 8-bit ends (U8IO, planar rows): a thread moves FOUR consecutive pixels as one dword, x = 4 (tid + i T) + q
	static constexpr int U8_ROUNDS = (N / 4 + T - 1) / T;
	static constexpr bool U8_OK = (C == 1) && (GS == 1) && (N % 4 == 0) && std::is_same<Re, float>::value;
	// butterflies of stage I per thread, and where its twiddles start in State::stw
	template <int I> static constexpr int stage_rounds() { return (C * (L / pack_get<I, Rs...>()) + T - 1) / T; }
	template <int I> static constexpr int stw_off() { if constexpr (I <= 0) return 0; else return stw_off<I - 1>() + stage_rounds<I - 1>(); }
	static constexpr int STW_N = stw_off<NS - 1>();
	// per-thread registers that live across barriers: the last stage's butterflies and the
	// line's global data, loaded before the first LDS phase
	template <int KIND> struct State {
		CX x[LAST_ROUNDS * RL];
		Re pre[(KIND == KIND_REDFT10 ? (U8_OK && 4 * U8_ROUNDS > PIX_ROUNDS ? 4 * U8_ROUNDS : PIX_ROUNDS) : 4 * K_ROUNDS) * C];
		CX tw[K_ROUNDS];     // T[k] of this thread's (k, L - k) pairs, fetched with the line: a load at its point of use (REDFT01's phase 0,
		                     // REDFT10's closing phase) is a full cache round trip in front of every line's arithmetic
		CX stw[STW_N > 0 ? STW_N : 1];   // the stages' twiddles W[m TW] of this thread's butterflies (fetch_stage_twiddles), for the same reason
	};
	// Issue the loads of every stage's twiddles at the head of the kernel (round 5).  A stage used to load W[m TW] behind its barrier: a cache round
	// trip in front of a few hundred cycles of butterflies, in each of the NS - 1 stage phases (motion's 1920-sample rows: the two stage phases
	// took 5.2K + 5.7K of a workgroup's 26K clocks for 40 + 150 vector instructions; tools/kstamp u8).
	template <class ST>
	static DSP_HD void fetch_stage_twiddles(const PA &a, int tid, ST &st)
	{
		static_for<0, NS - 1>([&](auto I) {
			constexpr int R = pack_get<I, Rs...>(), Lc = pack_lc<I, Rs...>(), M1 = Lc / R, NB = L / R, TW = L / Lc;
			static_for<0, stage_rounds<I>()>([&](auto j) {
				const int it = tid + j * T;
				if ((j + 1) * T <= C * NB || it < C * NB) st.stw[stw_off<I>() + j] = a.W[((it % NB) % M1) * TW];
			});
		});
	}

	// issue the global loads of one line into registers (no LDS access, no waiting)
	template <int KIND, class ST>
	// ch: channel lines only -- which channel of the interleaved line `bin` already points at (the tile flags go by the offset in the line)
	static DSP_HD void prefetch(const PA &a, long long bin, int tid, ST &st, const U8IO *io = nullptr, const uint8_t *zf = nullptr, int ch = 0)
	{
		if (a.mask) prefetch_m<KIND, true, false>(a, bin, tid, st, io, nullptr, ch);
		else if (zf) prefetch_m<KIND, false, true>(a, bin, tid, st, io, zf, ch);       // zf: this line's tile flags (PassGeom::zflags)
		else prefetch_m<KIND, false, false>(a, bin, tid, st, io, nullptr, ch);
	}
	template <int KIND, bool MASKED, bool FLAGGED, class ST>
	static DSP_HD void prefetch_m(const PA &a, long long bin, int tid, ST &st, const U8IO *io, const uint8_t *zf, int ch = 0)
	{
		static_for<0, K_ROUNDS>([&](auto i) {
			const int k = tid + i * T;
			if ((i + 1) * T <= L / 2 + 1 || k <= L / 2) st.tw[i] = a.T[k];
		});
		auto ld = [&](int x) {
			if constexpr (FLAGGED) return load_pix_z<C, Re>(a, zf, x * GS + ch, bin + (long long)x * GS);
			else if constexpr (MASKED) return load_pix_m<C, true, Re>(a, bin + (long long)x * GS);
			else {
				// dspfft_plan_set_input_window: pixels outside [win_lo, win_hi) are zero by contract and are not read (no branch around the
				// load, see load_pix_m: they load from a page of zeros).  Folds away in the plain instantiation.
				// dspfft_plan_set_input_modulation: the sample sits at position px of the line and is multiplied by in_mul[px] on the way in
				const int px = a.in_rev > 0 ? a.in_rev - x : x;
				const bool outside = a.win_hi > 0 && (x < a.win_lo || x >= a.win_hi);
				const Re *p = outside ? reinterpret_cast<const Re *>(a.zpage) : a.in + bin + (long long)px * GS;
				Pix<C, Re> v = load_pix<C, Re>(p);
				if (a.in_mul) {
					const Re m = *(outside ? reinterpret_cast<const Re *>(a.zpage) : reinterpret_cast<const Re *>(a.in_mul) + px);
					static_for<0, C>([&](auto c) { v.v[c] *= m; });
				}
				return v;
			}
		};
		if constexpr (KIND == KIND_REDFT10 && U8_OK) {
			if (io && io->in) {
				static_for<0, U8_ROUNDS>([&](auto i) {
					const int g = tid + i * T;
					if ((i + 1) * T <= N / 4 || g < N / 4) {
						uint32_t w4;
						__builtin_memcpy(&w4, io->in + bin + 4 * g, 4);
						static_for<0, 4>([&](auto q) { st.pre[i * 4 + q] = (Re)((w4 >> (8 * q)) & 0xffu); });
					}
				});
				return;
			}
		}
		if constexpr (KIND == KIND_REDFT10) {
			static_for<0, PIX_ROUNDS>([&](auto i) {
				const int x = tid + i * T;
				if ((i + 1) * T <= N || x < N) {
					const Pix<C, Re> v = ld(x);
					static_for<0, C>([&](auto c) { st.pre[i * C + c] = v.v[c]; });
				}
			});
		} else {
			static_for<0, K_ROUNDS>([&](auto i) {
				const int k = tid + i * T;
				if ((i + 1) * T <= L / 2 + 1 || k <= L / 2) {
					const Pix<C, Re> p0 = ld(k), p1 = ld(k ? N - k : 0), p2 = ld(L - k), p3 = ld(L + k);
					static_for<0, C>([&](auto c) {
						st.pre[(i * 4 + 0) * C + c] = p0.v[c]; st.pre[(i * 4 + 1) * C + c] = p1.v[c];
						st.pre[(i * 4 + 2) * C + c] = p2.v[c]; st.pre[(i * 4 + 3) * C + c] = p3.v[c];
					});
				}
			});
		}
	}

	// ---- REDFT01 loads behind a masked column pass, with the tile flags of THIS THREAD's samples precomputed (row_pair_pipe_kernel's scan form) ----
	// The flags are per column tile (zf[(offset in the line) >> zshift]): which of a thread's samples lie in skipped tiles is the same for every line of a pass.
	// A persistent workgroup therefore looks the flags up ONCE -- bit (ri * 4 + j) * C + c of the result: sample c of pixel j (k, N - k, L - k, L + k) of round
	// ri is to be read -- where prefetch_m's FLAGGED form spends two flag loads, three shifts and a handful of selects on every pixel of every line (about 300
	// of a line's 1100 vector instructions on 7680 x 3 lines).  zf == nullptr: everything is read.
	static_assert(4 * K_ROUNDS * C <= 64 || C > 4, "one bit per sample of a REDFT01 line's loads");
	static DSP_HD int pixel01(int k, int j) { return j == 0 ? k : j == 1 ? (k ? N - k : 0) : j == 2 ? L - k : L + k; }
	static DSP_HD unsigned long long flag_bits01(const uint8_t *zf, int zshift, int tid)
	{
		unsigned long long m = 0;
		static_for<0, K_ROUNDS>([&](auto ri) {
			const int k = tid + ri * T;
			if (!((ri + 1) * T <= L / 2 + 1 || k <= L / 2)) return;
			static_for<0, 4>([&](auto j) {
				const int o = pixel01(k, j) * GS;
				static_for<0, C>([&](auto c) { if (!zf || zf[(o + c) >> zshift]) m |= 1ull << ((ri * 4 + j) * C + c); });
			});
		});
		return m;
	}
	template <class ST>
	static DSP_HD void prefetch01_bits(const PA &a, long long bin, int tid, ST &st, unsigned long long bits)
	{
		static_for<0, K_ROUNDS>([&](auto i) {
			const int k = tid + i * T;
			if ((i + 1) * T <= L / 2 + 1 || k <= L / 2) st.tw[i] = a.T[k];
		});
		static_for<0, K_ROUNDS>([&](auto i) {
			const int k = tid + i * T;
			if ((i + 1) * T <= L / 2 + 1 || k <= L / 2) {
				static_for<0, 4>([&](auto j) {
					const unsigned b = (unsigned)(bits >> ((i * 4 + j) * C)) & ((1u << C) - 1u);
					// a pixel of skipped tiles is not loaded at all (the lanes that need it load under the execution mask; a wave none of whose lanes
					// does skips the instruction): against a load from the page of zeros, as load_pix_z does it, the scan step 337.5 -> 329.5 us
					Pix<C, Re> v;
					static_for<0, C>([&](auto c) { v.v[c] = (Re)0; });
					if (b) v = load_pix<C, Re>(a.in + bin + (long long)pixel01(k, j) * GS);
					static_for<0, C>([&](auto c) { st.pre[(i * 4 + j) * C + c] = ((b >> c) & 1u) ? v.v[c] : (Re)0; });
				});
			}
		});
	}

	static DSP_HD int padded(int p) { return p + (p / SB) * PADC; }

	// stages 0 .. NS-2 (in place, padded layout)
	// stw: the thread's twiddles of this stage as fetch_stage_twiddles left them, or nullptr: load them here
	template <int I, bool SEQTW = false>
	static DSP_HD void stage(const PA &a, CX *planes, int tid, const CX *stw = nullptr)
	{
		constexpr int R = pack_get<I, Rs...>(), Lc = pack_lc<I, Rs...>(), M1 = Lc / R, NB = L / R, TW = L / Lc;
		static_for<0, stage_rounds<I>()>([&](auto j) {
			const int it = tid + j * T;
			if (!((j + 1) * T <= C * NB || it < C * NB)) return;
			const int c = it / NB, q = it - c * NB;
			const int blk = q / M1, m = q - blk * M1;
			CX *p;
			int stride;
			if constexpr (I == 0) { p = planes + c * PL + m; stride = SB + PADC; }
			else { p = planes + c * PL + padded(blk * Lc) + m; stride = M1; }
			CX x[R];
			static_for<0, R>([&](auto r) { x[r] = p[r * stride]; });
			Dft<R>::run(x);
			if constexpr (M1 > 1) {
				const CX w1 = stw ? stw[stw_off<I>() + j] : a.W[m * TW];
				if constexpr (SEQTW) twiddle_chain_seq<R, CX>(x, w1); else twiddle_chain<R, CX>(x, w1);
			}
			static_for<0, R>([&](auto r) { p[r * stride] = x[r]; });
		});
	}

	// last stage, part 1: gather (digit-reversed) + butterfly into registers
	template <class ST>
	static DSP_HD void last_read(CX *planes, ST &st, int tid)
	{
		static_for<0, LAST_ROUNDS>([&](auto i) {
			const int it = tid + i * T;
			if (it < C * NBL) {
				const int c = it / NBL, kb = it - c * NBL;
				int blk;
				if constexpr (NS >= 2) blk = PosCalcFirst<NBL, NS - 1, Rs..., 1>::run(kb); else blk = 0;
				const CX *p = planes + c * PL + (NS >= 2 ? padded(blk * RL) : 0);
				static_for<0, RL>([&](auto r) { st.x[i * RL + r] = p[r]; });
				Dft<RL>::run(&st.x[i * RL]);
			}
		});
	}
	// last stage, part 2 (after a barrier): natural-order write  k = kb + NBL * r
	template <class ST>
	static DSP_HD void last_write(CX *planes, const ST &st, int tid)
	{
		static_for<0, LAST_ROUNDS>([&](auto i) {
			const int it = tid + i * T;
			if (it < C * NBL) {
				const int c = it / NBL, kb = it - c * NBL;
				CX *p = planes + c * PL + kb;
				static_for<0, RL>([&](auto r) { p[r * NBL] = st.x[i * RL + r]; });
			}
		});
	}

	// phase 0 consumes the prefetched registers; phases 1.. work on LDS; the last one stores to `bout`
	// STW: st.stw holds the stages' twiddles (the caller ran fetch_stage_twiddles)
	template <int KIND, int PH, class ST, bool SEQTW = false, bool STW = false>
	static DSP_HD void phase(const PA &a, CX *planes, long long bout, int tid, ST &st, const U8IO *io = nullptr)
	{
		Re *pf = reinterpret_cast<Re *>(planes);
		if constexpr (PH == 0) {
			if constexpr (KIND == KIND_REDFT10 && U8_OK) {
				if (io && io->in) {       // the prefetch took four consecutive pixels per round
					// pixels 4g, 4g + 2 are reordered samples 2g, 2g + 1 = slot g whole; 4g + 3, 4g + 1 are samples N - 2 - 2g, N - 1 - 2g = slot L - 1 - g
					// whole: two 8-byte writes at a stride of one slot per lane.  (As four 4-byte writes at a stride of two floats every one of them
					// was a two-way bank conflict: 40 % of this kernel's LDS cycles, profiles/r05_motion_sq.txt.)
					static_for<0, U8_ROUNDS>([&](auto i) {
						const int g = tid + i * T;
						if ((i + 1) * T <= N / 4 || g < N / 4) {
							Re v0 = st.pre[i * 4];
							if constexpr (i == 0) { if (g == 0) v0 *= a.in_scale0; }
							planes[padded(g)] = cmk<Re>(v0, st.pre[i * 4 + 2]);
							planes[padded(L - 1 - g)] = cmk<Re>(st.pre[i * 4 + 3], st.pre[i * 4 + 1]);
						}
					});
					return;
				}
			}
			if constexpr (KIND == KIND_REDFT10) {
				// pixel x = tid + i T -> reordered sample n = x / 2 (x even) or N - 1 - (x - 1) / 2 (x odd); Re index inside the (padded) channel plane.
				// T is a multiple of 4, so a thread's pixels all have ITS parity: n walks from n0 in steps of +-T/2 and keeps its low bit, and only the
				// first round can hold pixel 0.  (Round 5: the general form -- a parity select, a signed division by the sub-block length and a select +
				// multiply for in_scale0 on every sample -- was 184 of this phase's 197 vector instructions on 7680 x 3 lines, where the phases run at
				// the vector issue rate: profiles/r05_isa_row_pair.txt.)
				static_assert(T % 4 == 0, "a thread's pixels share a parity and a low bit");
				const unsigned odd = (unsigned)tid & 1u, n0 = odd ? (unsigned)(N - 1) - ((unsigned)tid >> 1) : ((unsigned)tid >> 1), low = n0 & 1u;
				const int dn = odd ? -(T / 2) : (T / 2);
				static_for<0, PIX_ROUNDS>([&](auto i) {
					const int x = tid + i * T;
					if ((i + 1) * T <= N || x < N) {
						const unsigned ph_ = (unsigned)((int)n0 + i * dn) >> 1;
						const unsigned f = 2u * (ph_ + (ph_ / (unsigned)SB) * (unsigned)PADC) + low;
						static_for<0, C>([&](auto c) {
							Re v = st.pre[i * C + c];
							if constexpr (i == 0) { if (x == 0) v *= a.in_scale0; }
							pf[c * (2 * PL) + f] = v;
						});
					}
				});
			} else {
				static_for<0, K_ROUNDS>([&](auto i) {
					const int k = tid + i * T;
					if ((i + 1) * T <= L / 2 + 1 || k <= L / 2) {
						const CX tk = st.tw[i];
						const CX tlk = cmul(cconj(tk), cmk<Re>((Re)0.70710678118654752440, (Re)-0.70710678118654752440));   // T[L-k]
						const CX t1 = csqr(csqr(tk));                                                      // T[4k]
						static_for<0, C>([&](auto c) {
							const Re xk = st.pre[(i * 4 + 0) * C + c], xnk = st.pre[(i * 4 + 1) * C + c];
							const Re xlk = st.pre[(i * 4 + 2) * C + c], xlpk = st.pre[(i * 4 + 3) * C + c];
							const Re x0 = (k == 0) ? xk * a.in_scale0 : xk;
							const CX Vk = cmulc(cmk<Re>(x0, k ? -xnk : (Re)0), tk);
							const CX Vm = cmulc(cmk<Re>(xlk, -xlpk), tlk);
							const CX S = cadd(Vk, cconj(Vm)), D = csub(Vk, cconj(Vm));
							const CX Q = cmul_pi(cmulc(D, t1));
							planes[c * PL + padded(k)] = cconj(cadd(S, Q));
							if (k > 0) planes[c * PL + padded(L - k)] = csub(S, Q);
						});
					}
				});
			}
		} else if constexpr (PH < NS) {
			if constexpr (STW) stage<PH - 1, SEQTW>(a, planes, tid, st.stw); else stage<PH - 1, SEQTW>(a, planes, tid);
		} else if constexpr (PH == NS) {
			last_read(planes, st, tid);
		} else if constexpr (PH == NS + 1) {
			last_write(planes, st, tid);
		} else {
			if constexpr (KIND == KIND_REDFT10) {
				static_for<0, K_ROUNDS>([&](auto ri) {
					const int k = tid + ri * T;
					if (!((ri + 1) * T <= L / 2 + 1 || k <= L / 2)) return;
					const int km = k ? L - k : 0;
					const CX tk = st.tw[ri];
					const CX tlk = cmul(cconj(tk), cmk<Re>((Re)0.70710678118654752440, (Re)-0.70710678118654752440));
					const CX t1 = csqr(csqr(tk));
					Pix<C, Re> o0, o1, o2, o3;
					static_for<0, C>([&](auto c) {
						const CX zk = planes[c * PL + k];
						const CX zm = cconj(planes[c * PL + km]);
						const CX E = cadd(zk, zm);                  // 2E
						const CX D = cmul_mi(csub(zk, zm));         // 2D
						const CX P = cmul(t1, D);
						const CX wk = cmul(tk, cadd(E, P));         // 2 * T[k] V[k]
						const CX wm = cmul(tlk, cconj(csub(E, P))); // 2 * T[L-k] V[L-k]
						const Re sc = a.scale;
						o0.v[c] = wk.x * (k == 0 ? sc * a.out_scale0 : sc);
						o1.v[c] = -wk.y * sc;
						o2.v[c] = wm.x * sc;
						o3.v[c] = -wm.y * sc;
					});
					store_pix_a<C, Re>(a, bout + (long long)k * GS, o0);
					if (k > 0) store_pix_a<C, Re>(a, bout + (long long)(N - k) * GS, o1);
					if (L - k != k) store_pix_a<C, Re>(a, bout + (long long)(L - k) * GS, o2);
					if (k > 0 && L + k != N - k) store_pix_a<C, Re>(a, bout + (long long)(L + k) * GS, o3);
				});
			} else {
				if constexpr (U8_OK) {
					if (io && io->out) {      // four consecutive quantised pixels per dword store
						const float mulf = (float)io->mul;
						tloop<N / 4, T>(tid, [&](int g) {
							// (the same two slots as REDFT10's phase 0 above, read whole; odd reordered samples carry the minus sign)
							const CX lo = planes[g], hi = planes[L - 1 - g];
							const Re f[4] = {lo.x, -hi.y, -lo.y, hi.x};
							uint32_t w4 = 0;
							static_for<0, 4>([&](auto q) {
								const Re sc = (q == 0 && g == 0) ? a.scale * a.out_scale0 : a.scale;
								w4 |= quantise_u8_of(f[q] * sc, io->mul, mulf) << (8 * q);
							});
							__builtin_memcpy(io->out + bout + 4 * g, &w4, 4);
						});
						return;
					}
				}
				// output pixel x = tid + i T reads reordered sample n (see phase 0 of REDFT10: n = n0 + i dn, the low bit of n and the parity of x are
				// the thread's own): its sign (-1)^n and the alternating output sign fold into ONE factor per thread, out_scale0 touches round 0 only
				static_assert(T % 4 == 0, "a thread's pixels share a parity and a low bit");
				const unsigned odd = (unsigned)tid & 1u, n0 = odd ? (unsigned)(N - 1) - ((unsigned)tid >> 1) : ((unsigned)tid >> 1);
				const int dn = odd ? -(T / 2) : (T / 2);
				Re sg = (n0 & 1u) ? -a.scale : a.scale;
				if (a.alt_out && odd) sg = -sg;                  // dspfft_plan_set_output_alternate; folds away in the plain instantiation
				auto value = [&](auto i, int x) {
					const int n = (int)n0 + i * dn;
					Pix<C, Re> o;
					Re sc = sg;
					if constexpr (i == 0) { if (x == 0) sc *= a.out_scale0; }
					static_for<0, C>([&](auto c) { o.v[c] = pf[c * (2 * PL) + n] * sc; });
					return o;
				};
				if (a.accumulate) {
					// sum += image (scan.c:451-459): ALL the old values first, then add and store -- a load issued right before its store
					// waits for its own data every time (the compiler may not move it above the previous store into the same array)
					Pix<C, Re> old[PIX_ROUNDS];
					static_for<0, PIX_ROUNDS>([&](auto i) {
						const int x = tid + i * T;
						if ((i + 1) * T <= N || x < N) old[i] = load_pix<C, Re>(a.out + bout + (long long)x * GS);
					});
					static_for<0, PIX_ROUNDS>([&](auto i) {
						const int x = tid + i * T;
						if ((i + 1) * T <= N || x < N) {
							Pix<C, Re> o = value(i, x);
							static_for<0, C>([&](auto c) { o.v[c] += old[i].v[c]; });
							store_pix<C, Re>(a.out + bout + (long long)x * GS, o);
						}
					});
				} else {
					static_for<0, PIX_ROUNDS>([&](auto i) {
						const int x = tid + i * T;
						if ((i + 1) * T <= N || x < N) store_pix<C, Re>(a.out + bout + (long long)x * GS, value(i, x));
					});
				}
			}
		}
	}

	// ---- a plain line's outputs as VALUES (row_pair_pipe_kernel: the pair's butterfly moved to the output side) ----
	// The closing phase of a plain pass (phase<KIND, NS + 2> without 8-bit ends, accumulation or alternating sign) with its stores replaced by
	// f(slot, offset of the pixel within the line in samples, pixel): `slot` is a compile-time index < out_slots() that names the same pixel in every
	// line of this thread, called for the pixels this thread owns only.
	template <int KIND> static constexpr int out_slots() { return KIND == KIND_REDFT10 ? 4 * K_ROUNDS : PIX_ROUNDS; }
	template <int KIND> struct OutHold { Pix<C, Re> v[out_slots<KIND>()]; };
	// UNCOND (REDFT10): f is called for EVERY slot of every thread, so that the caller's stores are unconditional and the compiler can count them (a wait for
	// loads issued in front of them is then vmcnt(number of stores), not vmcnt(0)).  A thread beyond the last item recomputes item L/2, and the slots that have
	// no pixel of their own (k = 0: N - k and L + k; k = L/2: L - k and L + k) repeat a sibling slot's pixel -- the same value to the same address, harmless.
	// The twiddle of a clamped item must be item L/2's: the caller loads st.tw through tw_index().
	template <bool UNCOND> static DSP_HD int tw_index(int k) { return UNCOND && k > L / 2 ? L / 2 : k; }
	template <int KIND, bool UNCOND = false, class ST, class F>
	static DSP_HD void final_each(const PA &a, const CX *planes, int tid, const ST &st, F &&f)
	{
		if constexpr (KIND == KIND_REDFT10) {
			static_for<0, K_ROUNDS>([&](auto ri) {
				const int k0 = tid + ri * T;
				if constexpr (!UNCOND) { if (!((ri + 1) * T <= L / 2 + 1 || k0 <= L / 2)) return; }
				const int k = tw_index<UNCOND>(k0);
				const int km = k ? L - k : 0;
				const CX tk = st.tw[ri];
				const CX tlk = cmul(cconj(tk), cmk<Re>((Re)0.70710678118654752440, (Re)-0.70710678118654752440));
				const CX t1 = csqr(csqr(tk));
				const Re sc = a.scale, s0 = (k == 0) ? sc * a.out_scale0 : sc;
				Pix<C, Re> o0, o1, o2, o3;
				static_for<0, C>([&](auto c) {
					const CX zk = planes[c * PL + k];
					const CX zm = cconj(planes[c * PL + km]);
					const CX E = cadd(zk, zm);
					const CX D = cmul_mi(csub(zk, zm));
					const CX P = cmul(t1, D);
					const CX wk = cmul(tk, cadd(E, P));
					const CX wm = cmul(tlk, cconj(csub(E, P)));
					o0.v[c] = wk.x * s0; o1.v[c] = -wk.y * sc; o2.v[c] = wm.x * sc; o3.v[c] = -wm.y * sc;
				});
				if constexpr (UNCOND) {
					const bool first = k == 0, mid = 2 * k == L;
					const int e0 = k * GS, e1 = first ? e0 : (N - k) * GS, e2 = mid ? e0 : (L - k) * GS, e3 = first ? e2 : mid ? e1 : (L + k) * GS;
					Pix<C, Re> q1, q2, q3;
					static_for<0, C>([&](auto c) {
						q1.v[c] = first ? o0.v[c] : o1.v[c];
						q2.v[c] = mid ? o0.v[c] : o2.v[c];
						q3.v[c] = first ? q2.v[c] : mid ? q1.v[c] : o3.v[c];
					});
					f(std::integral_constant<int, ri * 4 + 0>(), (long long)e0, o0);
					f(std::integral_constant<int, ri * 4 + 1>(), (long long)e1, q1);
					f(std::integral_constant<int, ri * 4 + 2>(), (long long)e2, q2);
					f(std::integral_constant<int, ri * 4 + 3>(), (long long)e3, q3);
					return;
				}
				f(std::integral_constant<int, ri * 4 + 0>(), (long long)k * GS, o0);
				if (k > 0) f(std::integral_constant<int, ri * 4 + 1>(), (long long)(N - k) * GS, o1);
				if (L - k != k) f(std::integral_constant<int, ri * 4 + 2>(), (long long)(L - k) * GS, o2);
				if (k > 0 && L + k != N - k) f(std::integral_constant<int, ri * 4 + 3>(), (long long)(L + k) * GS, o3);
			});
		} else {
			const Re *pf = reinterpret_cast<const Re *>(planes);
			static_assert(T % 4 == 0, "a thread's pixels share a parity and a low bit");
			const unsigned odd = (unsigned)tid & 1u, n0 = odd ? (unsigned)(N - 1) - ((unsigned)tid >> 1) : ((unsigned)tid >> 1);
			const int dn = odd ? -(T / 2) : (T / 2);
			const Re sg = (n0 & 1u) ? -a.scale : a.scale;
			static_for<0, PIX_ROUNDS>([&](auto i) {
				const int x = tid + i * T;
				if (!((i + 1) * T <= N || x < N)) return;
				const int n = (int)n0 + i * dn;
				Re sc = sg;
				if constexpr (i == 0) { if (x == 0) sc *= a.out_scale0; }
				Pix<C, Re> o;
				static_for<0, C>([&](auto c) { o.v[c] = pf[c * (2 * PL) + n] * sc; });
				f(std::integral_constant<int, i>(), (long long)x * GS, o);
			});
		}
	}
	// ---- two REDFT01 transforms summed into one output line (row_sum2_kernel: out = A(in_a) + B(in_b)) ----
	// the first transform's output line waits in registers (N C / T values per thread) while the second runs through the same LDS
	struct Hold { Pix<C, Re> v[PIX_ROUNDS]; };
	// the closing phase of REDFT01 (phase<KIND_REDFT01, NS + 2>) with MODE 1: the line kept in `h`, MODE 2: h + this line stored
	template <int MODE>
	static DSP_HD void final01_hold(const PA &a, CX *planes, long long bout, int tid, Hold &h)
	{
		const Re *pf = reinterpret_cast<const Re *>(planes);
		static_for<0, PIX_ROUNDS>([&](auto i) {
			const int x = tid + i * T;
			if ((i + 1) * T <= N || x < N) {
				const int n = makhoul_dst(x, N);
				Re sc = (x == 0) ? a.scale * a.out_scale0 : a.scale;
				if (a.alt_out && (x & 1)) sc = -sc;
				Pix<C, Re> o;
				static_for<0, C>([&](auto c) { const Re f = pf[c * (2 * PL) + n]; o.v[c] = ((n & 1) ? -f : f) * sc; });
				if constexpr (MODE == 1) h.v[i] = o;
				else { static_for<0, C>([&](auto c) { o.v[c] += h.v[i].v[c]; }); store_pix_a<C, Re>(a, bout + (long long)x * GS, o); }
			}
		});
	}
};

template <class Re_, int N_, int C_, int T_, int... Rs> using RowSpecT = RowSpecG<Re_, N_, C_, C_, T_, Rs...>;
template <class Re_, int N_, int G_, int T_, int... Rs> using RowChanSpecT = RowSpecG<Re_, N_, 1, G_, T_, Rs...>;     // one channel of a G_-channel line
template <int N_, int C_, int T_, int... Rs> using RowSpec = RowSpecT<float, N_, C_, T_, Rs...>;

// =================================================================================================
DSP_HD int xcd_remap(int bid, int n)
{
	// blocks b and b+8 share an XCD (observed round-robin dispatch; speed only, never correctness):
	// give each XCD a contiguous run of tiles so neighbours that share cache lines share an L2.
	const int per = n >> 3, full = per << 3;
	if (bid >= full) return bid;
	return (bid & 7) * per + (bid >> 3);
}

// What a lane of a column pass computes with: the NCS complex signals of its 16 bytes as ONE complex number over a "lane real" --
// float: Pk2 (both signals side by side: every butterfly instruction is a packed one), double: the scalar itself.
template <class Re> struct lane_of;
template <> struct lane_of<float> { typedef Pk2 LR; };
template <> struct lane_of<double> { typedef double LR; };

template <class Re_, int N_, int K_, int T_, int... Rs>
struct ColSpecT {
	typedef Re_ Re;
	typedef cx<Re_> CX;
	typedef PassArgsT<Re_> PA;
	typedef typename sig_of<Re_>::type V;            // NCS complex signals = VW real columns (16 bytes)
	typedef typename lane_of<Re_>::LR LR;
	typedef cx<LR> LC;                               // the lane's signals as one complex number
	static constexpr int NCS = sig_of<Re_>::NCS, VW = 2 * NCS;
	static constexpr int N = N_, K = K_, T = T_, B = K_ / 2, NP = K_ / VW, NS = (int)sizeof...(Rs), NPH = NS + 3;
	static constexpr int WPE_D = (int)((160 * 1024) / ((size_t)(N_ + 16) * (K_ / 2) * sizeof(CX))) * T_ / 256;
	static constexpr int WPE = std::is_same<Re, double>::value ? (WPE_D < 1 ? 1 : WPE_D > 4 ? 4 : WPE_D) : 1;   // see RowSpecT
	static_assert((1 * ... * Rs) == N, "radices must multiply to N");
	static_assert(K % VW == 0 && NS >= 1, "tile width must be a multiple of the lane vector (4 floats / 2 doubles)");
	// rows of the tile are padded by one row per first-stage sub-block while the DIF stages run
	// (see RowSpec); natural order, unpadded, after the last stage.
	static constexpr int R0 = pack_get<0, Rs...>(), RL = pack_get<NS - 1, Rs...>();
	static constexpr int SB = N / R0, PADC = (NS >= 2) ? DSP_COL_PADC : 0;
	static constexpr int ROWS = N + R0 * PADC;
	static constexpr size_t LDS = (size_t)ROWS * B * sizeof(CX);
	static constexpr int NBL = N / RL;
	static constexpr int LAST_ROUNDS = (NBL * NP + T - 1) / T;
	static constexpr int Y_ROUNDS = (N * NP + T - 1) / T;              // REDFT10: (row, lane vector) items per thread
	static constexpr int K_ROUNDS = ((N / 2 + 1) * NP + T - 1) / T;    // REDFT01: (k, lane vector) items per thread
	template <int KIND> struct State {
		LC x[LAST_ROUNDS * RL];
		LC pre[KIND == KIND_REDFT10 ? Y_ROUNDS : 2 * K_ROUNDS];   // the loaded rows, column order (g_get)
		CX tw[KIND == KIND_REDFT01 ? K_ROUNDS : 1];   // REDFT01: T[k] of this thread's items, fetched with the data (measured: -5 us);
		                                              // REDFT10 loads T[k] where it is used (prefetching it there measured slower)
	};

	// ---- lane vector <-> lane complex ----
	// Global memory order is column order (c0, c1, c2, c3).  g_get gives x = the first column of each of the lane's signals, y = the
	// second.  In LDS a float lane keeps (re_a, re_b, im_a, im_b) so that l_get / l_put are register-pair moves.
	// (round 3) float lanes pair columns (c0, c2) and (c1, c3) -- any two real columns may share a complex signal -- so that the first
	// columns of both signals, c0 and c1, already sit in one aligned register pair as loaded (likewise c2, c3): g_get / g_put cost no
	// register moves (the (c0, c1) | (c2, c3) pairing needed two v_mov per 16-byte access: 40 of the column kernel's 109)
	static DSP_HD LC g_get(V v)
	{
		if constexpr (NCS == 2) return cmk<LR>(pk2(v.s[0].x, v.s[0].y), pk2(v.s[1].x, v.s[1].y)); else return cmk<LR>(v.s[0].x, v.s[0].y);
	}
	static DSP_HD V g_put(LC c)
	{
		V v;
		if constexpr (NCS == 2) { v.s[0].x = c.x.x; v.s[0].y = c.x.y; v.s[1].x = c.y.x; v.s[1].y = c.y.y; } else { v.s[0].x = c.x; v.s[0].y = c.y; }
		return v;
	}
	static DSP_HD LC l_get(V v)
	{
		if constexpr (NCS == 2) return cmk<LR>(pk2(v.s[0].x, v.s[0].y), pk2(v.s[1].x, v.s[1].y)); else return cmk<LR>(v.s[0].x, v.s[0].y);
	}
	static DSP_HD V l_put(LC c)
	{
		V v;
		if constexpr (NCS == 2) { v.s[0].x = c.x.x; v.s[0].y = c.x.y; v.s[1].x = c.y.x; v.s[1].y = c.y.y; } else { v.s[0].x = c.x; v.s[0].y = c.y; }
		return v;
	}
	// lane complex times a scalar complex (the same twiddle for every signal of the lane)
	static DSP_HD LC lmul(LC a, CX w) { return cmk<LR>(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
	static DSP_HD LC lmulc(LC a, CX w) { return cmk<LR>(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y); }   // a * conj(w)
	static DSP_HD LC lscale(LC a, Re f) { return cmk<LR>(a.x * f, a.y * f); }
	static DSP_HD LC lzero() { return cmk<LR>((LR)(Re)0, (LR)(Re)0); }

	// REDFT10's last step for FFT outputs zk = Z[k], zm = Z[N-k] of one lane: coefficient rows k (r0) and N-k (r1), column order
	static DSP_HD void post10(LC zk, LC zm, CX t, Re s0, Re sc, LC &r0, LC &r1)
	{
		const LC A = cmk<LR>(zk.x + zm.x, zk.y - zm.y), Bq = cmul_mi(cmk<LR>(zk.x - zm.x, zk.y + zm.y));
		const LC wa = lmul(A, t), wb = lmul(Bq, t);
		r0 = cmk<LR>(wa.x * s0, wb.x * s0);
		r1 = cmk<LR>(-wa.y * sc, -wb.y * sc);
	}
	// REDFT01's first step for coefficient rows xk (k) and xm (N-k), column order: FFT inputs at slots k (lo) and N-k (hi)
	static DSP_HD void pre01(LC xk, LC xm, CX t, LC &lo, LC &hi)
	{
		const LC Va = lmulc(cmk<LR>(xk.x, -xm.x), t), Vb = lmulc(cmk<LR>(xk.y, -xm.y), t);
		lo = cmk<LR>(Va.x - Vb.y, -Va.y - Vb.x);
		hi = cmk<LR>(Va.x + Vb.y, Va.y - Vb.x);
	}

	template <int KIND, class ST>
	static DSP_HD void prefetch(const PA &a, long long bin, int tid, ST &st) { bool hit = false; prefetch<KIND>(a, bin, tid, st, hit); }
	// hit (masked runs): set when this thread selected at least one coefficient
	template <int KIND, class ST>
	static DSP_HD void prefetch(const PA &a, long long bin, int tid, ST &st, bool &hit)
	{
		if (a.mask && a.mask_mode == 1) prefetch_2step<KIND, 1>(a, bin, tid, st, hit);
		else if (a.mask && a.mask_mode == 2) prefetch_2step<KIND, 2>(a, bin, tid, st, hit);
		else if (a.mask) prefetch_m<KIND, true>(a, bin, tid, st, hit); else prefetch_m<KIND, false>(a, bin, tid, st, hit);
	}
	// masked loads with the ids from the plan's table (masked_two_step; one image: bin = tile * K): the items and their order are prefetch_m's
	template <int KIND, int EB, class ST>
	static DSP_HD void prefetch_2step(const PA &a, long long bin, int tid, ST &st, bool &hit)
	{
		const int t = (int)(bin / K);
		if constexpr (KIND == KIND_REDFT10) {
			masked_two_step<EB, Y_ROUNDS, Re>(a, hit, [&](auto i, long long &off, long long &eoff) {
				const int it = tid + i * T;
				if (!((i + 1) * T <= N * NP || it < N * NP)) return false;
				const int y = it / NP, jp = it - y * NP;
				off = bin + (long long)y * a.es_in + VW * jp;
				eoff = eid_index(N, K, 1, t, y, VW * jp);
				return true;
			}, [&](auto i, V v) { st.pre[i] = g_get(v); });
		} else {
			static_for<0, K_ROUNDS>([&](auto i) {
				const int it = tid + i * T;
				if ((i + 1) * T <= (N / 2 + 1) * NP || it < (N / 2 + 1) * NP) st.tw[i] = a.T[it / NP];
			});
			masked_two_step<EB, 2 * K_ROUNDS, Re>(a, hit, [&](auto j, long long &off, long long &eoff) {
				constexpr int i = j / 2;
				const int it = tid + i * T;
				if (!((i + 1) * T <= (N / 2 + 1) * NP || it < (N / 2 + 1) * NP)) return false;
				const int k = it / NP, jp = it - k * NP;
				const int y = j % 2 ? (k ? N - k : 0) : k;
				off = bin + VW * jp + (long long)y * a.es_in;
				eoff = eid_index(N, K, 1, t, y, VW * jp);
				return true;
			}, [&](auto j, V v) { st.pre[j] = g_get(v); });
		}
	}
	template <int KIND, bool MASKED, class ST>
	static DSP_HD void prefetch_m(const PA &a, long long bin, int tid, ST &st, bool &hit)
	{
		if constexpr (KIND == KIND_REDFT01)
			static_for<0, K_ROUNDS>([&](auto i) {
				const int it = tid + i * T;
				if ((i + 1) * T <= (N / 2 + 1) * NP || it < (N / 2 + 1) * NP) st.tw[i] = a.T[it / NP];
			});
		if constexpr (KIND == KIND_REDFT10) {
			static_for<0, Y_ROUNDS>([&](auto i) {
				const int it = tid + i * T;
				if ((i + 1) * T <= N * NP || it < N * NP) {
					const int y = it / NP, jp = it - y * NP;
					st.pre[i] = g_get(loadv_m<MASKED, Re>(a, bin + (long long)y * a.es_in + VW * jp, hit));
				}
			});
		} else {
			const bool win = !MASKED && a.win_hi > 0;      // uniform; folds away in the plain instantiation
			static_for<0, K_ROUNDS>([&](auto i) {
				const int it = tid + i * T;
				if ((i + 1) * T <= (N / 2 + 1) * NP || it < (N / 2 + 1) * NP) {
					const int k = it / NP, jp = it - k * NP;
					const int km = k ? N - k : 0;
					const long long p = bin + VW * jp;
					if (win) {
						// rows outside the window are zero by contract: load the zero page instead (no branch around the loads, see load_pix_m)
						// dspfft_plan_set_input_modulation: row k sits at row pk = in_rev - k (or k) of the array and is multiplied by in_mul[pk]
						const bool ik = k >= a.win_lo && k < a.win_hi, im = km >= a.win_lo && km < a.win_hi;
						const int qk = a.in_rev > 0 ? a.in_rev - k : k, qm = a.in_rev > 0 ? a.in_rev - km : km;
						const V *pk = ik ? reinterpret_cast<const V *>(a.in + p + (long long)qk * a.es_in) : reinterpret_cast<const V *>(a.zpage);
						const V *pm = im ? reinterpret_cast<const V *>(a.in + p + (long long)qm * a.es_in) : reinterpret_cast<const V *>(a.zpage);
						LC vk = g_get(*pk), vm = g_get(*pm);
						if (a.in_mul) {
							const Re mk = *(ik ? reinterpret_cast<const Re *>(a.in_mul) + qk : reinterpret_cast<const Re *>(a.zpage));
							const Re mm = *(im ? reinterpret_cast<const Re *>(a.in_mul) + qm : reinterpret_cast<const Re *>(a.zpage));
							vk = lscale(vk, mk); vm = lscale(vm, mm);
						}
						st.pre[2 * i] = vk;
						st.pre[2 * i + 1] = vm;
					} else {
						st.pre[2 * i] = g_get(loadv_m<MASKED, Re>(a, p + (long long)k * a.es_in, hit));
						st.pre[2 * i + 1] = g_get(loadv_m<MASKED, Re>(a, p + (long long)km * a.es_in, hit));
					}
				}
			});
		}
	}

	static DSP_HD int padded(int n) { return n + (n / SB) * PADC; }

	static DSP_HD void base(const PA &a, int work, long long &bin, long long &bout) { int tile; base(a, work, bin, bout, tile); }
	// tile = index of the tile among all tiles of the launch (what PassGeom::zflags is indexed by)
	static DSP_HD void base(const PA &a, int work, long long &bin, long long &bout, int &tile)
	{
		const int bt = work / a.ntiles, t0 = work - bt * a.ntiles;
		// sparse scan frames keep a contiguous run of tiles and skip the rest: leave those runs spread over the XCDs
		const int t = (a.zflags && a.mask) ? t0 : xcd_remap(t0, a.ntiles);
		tile = bt * a.ntiles + t;
		const int i1 = bt / a.nb0, i0 = bt - i1 * a.nb0;
		bin = i0 * a.sb0_in + i1 * a.sb1_in + (long long)t * K;
		bout = i0 * a.sb0_out + i1 * a.sb1_out + (long long)t * K;
	}

	template <int I>
	static DSP_HD void stage(const PA &a, V *buf, int tid)
	{
		constexpr int R = pack_get<I, Rs...>(), Lc = pack_lc<I, Rs...>(), M1 = Lc / R, NB = N / R, TW = N / Lc;
		tloop<NB * NP, T>(tid, [&](int it) {
			const int q = it / NP, jp = it - q * NP;
			const int blk = q / M1, m = q - blk * M1;
			V *p;
			int stride;
			if constexpr (I == 0) { p = buf + m * NP + jp; stride = (SB + PADC) * NP; }
			else { p = buf + (padded(blk * Lc) + m) * NP + jp; stride = M1 * NP; }
			LC x[R];
			static_for<0, R>([&](auto r) { x[r] = l_get(p[r * stride]); });
			Dft<R>::run(x);
			if constexpr (M1 > 1) {
				CX w[R];
				w[1] = a.W[m * TW];
				static_for<2, R>([&](auto r) { if constexpr (r % 2 == 0) w[r] = csqr(w[r / 2]); else w[r] = cmul(w[r / 2], w[r - r / 2]); });
				static_for<1, R>([&](auto r) { x[r] = lmul(x[r], w[r]); });
			}
			static_for<0, R>([&](auto r) { p[r * stride] = l_put(x[r]); });
		});
	}

	// first slot block of last-stage butterfly kb (digit reversal over the earlier stages)
	static DSP_HD int last_blk(int kb) { if constexpr (NS >= 2) return PosCalcFirst<NBL, NS - 1, Rs..., 1>::run(kb); else return 0; }
	template <class ST>
	static DSP_HD void last_read(const V *buf, ST &st, int tid)
	{
		static_for<0, LAST_ROUNDS>([&](auto i) {
			const int it = tid + i * T;
			if (it < NBL * NP) {
				const int kb = it / NP, jp = it - kb * NP;
				int blk;
				if constexpr (NS >= 2) blk = PosCalcFirst<NBL, NS - 1, Rs..., 1>::run(kb); else blk = 0;
				const V *p = buf + (NS >= 2 ? padded(blk * RL) : 0) * NP + jp;
				static_for<0, RL>([&](auto r) { st.x[i * RL + r] = l_get(p[r * NP]); });
				Dft<RL>::run(&st.x[i * RL]);
			}
		});
	}
	template <class ST>
	static DSP_HD void last_write(V *buf, const ST &st, int tid)
	{
		static_for<0, LAST_ROUNDS>([&](auto i) {
			const int it = tid + i * T;
			if (it < NBL * NP) {
				const int kb = it / NP, jp = it - kb * NP;
				V *p = buf + kb * NP + jp;
				static_for<0, RL>([&](auto r) { p[r * NBL * NP] = l_put(st.x[i * RL + r]); });
			}
		});
	}

	// REDFT01 without accumulation closes in the LAST STAGE (round 4): the butterfly's outputs are FFT outputs n = kb + NBL r in natural order,
	// i.e. output rows makhoul_src(n) up to a sign and the scale -- a permutation, so they are stored from registers instead of going
	// through the natural-order write and a closing phase that reads them back (two LDS round trips and two barriers fewer per tile).
	// (The mirror image for REDFT10 -- the first butterflies fed from the registers the global loads fill, in butterfly order -- was built
	// and measured slower: three quarters of the threads then issue all the loads, 12 each; profiles/r04_lean_col.txt.)
	static DSP_HD bool lean01(const PA &a) { return !a.accumulate && !a.lean_off; }
	template <int KIND> static DSP_HD bool skip_phase(const PA &a, int ph) { return KIND == KIND_REDFT01 && ph > NS && lean01(a); }
	template <int KIND> static DSP_HD bool barrier_after(const PA &a, int ph) { return (KIND == KIND_REDFT01 && lean01(a)) ? ph < NS : ph + 1 < NPH; }
	static DSP_HD void last_store01(const PA &a, const V *buf, long long bout, int tid)
	{
		static_for<0, LAST_ROUNDS>([&](auto i) {
			const int it = tid + i * T;
			if (it < NBL * NP) {
				const int kb = it / NP, jp = it - kb * NP;
				const V *p = buf + (NS >= 2 ? padded(last_blk(kb) * RL) : 0) * NP + jp;
				LC x[RL];
				static_for<0, RL>([&](auto r) { x[r] = l_get(p[r * NP]); });
				Dft<RL>::run(x);
				static_for<0, RL>([&](auto r) {
					const int y = makhoul_src(kb + NBL * r, N);
					Re sc = (y == 0) ? a.scale * a.out_scale0 : a.scale;
					if (a.alt_out && (y & 1)) sc = -sc;          // folds away in the plain instantiation
					*reinterpret_cast<V *>(a.out + bout + (long long)y * a.es_out + VW * jp) = g_put(cmk<LR>(x[r].x * sc, -x[r].y * sc));
				});
			}
		});
	}

	// ---- fused forward -> pointwise filter -> inverse along this axis (dspfft_execute_roundtrip) ----
	// The tile never leaves LDS between the two transforms.  mid_read turns the forward FFT output into the
	// coefficient rows k and N-k (exactly REDFT10's last phase), filters them, and applies REDFT01's first phase;
	// the results wait in registers across one barrier because the inverse's padded layout overlaps slots other
	// threads still have to read.
	struct StateRT {
		LC x[LAST_ROUNDS * RL];
		LC pre[(Y_ROUNDS > 2 * K_ROUNDS) ? Y_ROUNDS : 2 * K_ROUNDS];
		CX tw[1];
	};
	template <class ST, class F>
	static DSP_HD void mid_read(const PA &af, const PA &ai, const V *buf, long long bout, int tid, ST &st, const F &filt, unsigned long long &coded)
	{
		static_for<0, K_ROUNDS>([&](auto ri) {
			const int it = tid + ri * T;
			if (!((ri + 1) * T <= (N / 2 + 1) * NP || it < (N / 2 + 1) * NP)) return;
			const int k = it / NP, jp = it - k * NP;
			const int km = k ? N - k : 0;
			const LC zk = l_get(buf[k * NP + jp]), zm = l_get(buf[km * NP + jp]);
			const CX t = af.T[k];
			const Re sc = af.scale, s0 = (k == 0) ? sc * af.out_scale0 : sc;
			const long long o = bout + VW * jp;
			LC r0, r1;
			post10(zk, zm, t, s0, sc, r0, r1);                                   // coefficient rows k and N-k
			V xk = g_put(r0), xm = g_put(r1);
			DSP_SCHED_FENCE();
			xk = filt(o + (long long)k * af.es_out, xk, coded);
			DSP_SCHED_FENCE();
			if (k > 0 && km != k) xm = filt(o + (long long)km * af.es_out, xm, coded);
			else if (km == k && k > 0) xm = xk;
			DSP_SCHED_FENCE();                                                   // k = N/2: the same row
			// REDFT01's first phase on rows k and N-k (ColSpec::phase<KIND_REDFT01, 0>)
			LC ck = g_get(xk), cm = g_get(xm);
			if (k == 0) { ck = lscale(ck, ai.in_scale0); cm = lzero(); }
			LC lo, hi;
			pre01(ck, cm, t, lo, hi);
			st.pre[2 * ri] = lo; st.pre[2 * ri + 1] = hi;
			DSP_SCHED_FENCE();
		});
	}
	template <class ST>
	static DSP_HD void mid_write(V *buf, int tid, const ST &st)
	{
		static_for<0, K_ROUNDS>([&](auto ri) {
			const int it = tid + ri * T;
			if (!((ri + 1) * T <= (N / 2 + 1) * NP || it < (N / 2 + 1) * NP)) return;
			const int k = it / NP, jp = it - k * NP;
			const int km = k ? N - k : 0;
			buf[padded(k) * NP + jp] = l_put(st.pre[2 * ri]);
			if (k > 0) buf[padded(km) * NP + jp] = l_put(st.pre[2 * ri + 1]);
		});
	}

	template <int KIND, int PH, class ST>
	static DSP_HD void phase(const PA &a, V *buf, long long bout, int tid, ST &st)
	{
		if constexpr (PH == 0) {
			if constexpr (KIND == KIND_REDFT10) {
				static_for<0, Y_ROUNDS>([&](auto i) {
					const int it = tid + i * T;
					if ((i + 1) * T <= N * NP || it < N * NP) {
						const int y = it / NP, jp = it - y * NP;
						LC v = st.pre[i];
						if (y == 0) v = lscale(v, a.in_scale0);
						buf[padded(makhoul_dst(y, N)) * NP + jp] = l_put(v);
					}
				});
			} else {
				static_for<0, K_ROUNDS>([&](auto i) {
					const int it = tid + i * T;
					if ((i + 1) * T <= (N / 2 + 1) * NP || it < (N / 2 + 1) * NP) {
						const int k = it / NP, jp = it - k * NP;
						const int km = k ? N - k : 0;
						LC xk = st.pre[2 * i], xm = st.pre[2 * i + 1];
						if (k == 0) { xk = lscale(xk, a.in_scale0); xm = lzero(); }
						LC lo, hi;
						pre01(xk, xm, st.tw[i], lo, hi);
						buf[padded(k) * NP + jp] = l_put(lo);
						if (k > 0) buf[padded(km) * NP + jp] = l_put(hi);
					}
				});
			}
		} else if constexpr (PH < NS) {
			stage<PH - 1>(a, buf, tid);
		} else if constexpr (PH == NS) {
			if (KIND == KIND_REDFT01 && lean01(a)) last_store01(a, buf, bout, tid); else last_read(buf, st, tid);
		} else if constexpr (PH == NS + 1) {
			if (KIND == KIND_REDFT01 && lean01(a)) return;
			last_write(buf, st, tid);
		} else {
			if (KIND == KIND_REDFT01 && lean01(a)) return;
			if constexpr (KIND == KIND_REDFT10) {
				static_for<0, K_ROUNDS>([&](auto ri) {
					const int it = tid + ri * T;
					if (!((ri + 1) * T <= (N / 2 + 1) * NP || it < (N / 2 + 1) * NP)) return;
					const int k = it / NP, jp = it - k * NP;
					const int km = k ? N - k : 0;
					const LC zk = l_get(buf[k * NP + jp]), zm = l_get(buf[km * NP + jp]);
					const Re sc = a.scale, s0 = (k == 0) ? sc * a.out_scale0 : sc;
					const long long o = bout + VW * jp;
					LC r0, r1;
					post10(zk, zm, a.T[k], s0, sc, r0, r1);
					storev_a<Re>(a, o + (long long)k * a.es_out, g_put(r0));
					if (k > 0 && km != k) storev_a<Re>(a, o + (long long)km * a.es_out, g_put(r1));
				});
			} else {
				auto value = [&](int it, long long &off) {
					const int n = it / NP, jp = it - n * NP;
					const LC F = l_get(buf[n * NP + jp]);
					const int y = makhoul_src(n, N);
					Re sc = (y == 0) ? a.scale * a.out_scale0 : a.scale;
					if (a.alt_out && (y & 1)) sc = -sc;          // folds away in the plain instantiation
					off = bout + (long long)y * a.es_out + VW * jp;
					return g_put(cmk<LR>(F.x * sc, -F.y * sc));
				};
				if (a.accumulate) {
					// read-modify-write: all the old values first (the loads go out back to back), then add and store -- a load issued
					// right before its store would wait for its own data every time (the compiler cannot move it above the previous
					// store: same array)
					constexpr int ROUNDS = (N * NP + T - 1) / T;
					V old[ROUNDS];
					static_for<0, ROUNDS>([&](auto i) {
						const int it = tid + i * T;
						if ((i + 1) * T <= N * NP || it < N * NP) {
							const int n = it / NP, jp = it - n * NP;
							old[i] = *reinterpret_cast<const V *>(a.out + bout + (long long)makhoul_src(n, N) * a.es_out + VW * jp);
						}
					});
					static_for<0, ROUNDS>([&](auto i) {
						const int it = tid + i * T;
						if ((i + 1) * T <= N * NP || it < N * NP) {
							long long off;
							V r = value(it, off);
							static_for<0, NCS>([&](auto q) { r.s[q].x += old[i].s[q].x; r.s[q].y += old[i].s[q].y; });
							*reinterpret_cast<V *>(a.out + off) = r;
						}
					});
				} else {
					tloop<N * NP, T>(tid, [&](int it) { long long off; const V r = value(it, off); *reinterpret_cast<V *>(a.out + off) = r; });
				}
			}
		}
	}
};

template <int N_, int K_, int T_, int... Rs> using ColSpec = ColSpecT<float, N_, K_, T_, Rs...>;

// =================================================================================================
// Column pass split by an OUTER RADIX 2 across two kernels, so that a long column (2160, 4320 rows) is transformed on
// tiles of N/2 rows x twice the width: 64-B row segments instead of 32-B ones for the same LDS footprint (measured with
// no arithmetic on MI355X, tools/membench2: 2160x8 tiles 38.0 us per 4K frame, 1080x16 tiles 32.8 us, linear copy 29.5 us).
//
// With M = N/2, w = exp(-2 pi i / N) and v[] the even/odd ("Makhoul") reordered column, v[n] = x[2n], v[n+M] = x[N-1-2n]:
//   REDFT10  (decimation in frequency)  F[2q] = FFT_M(v[n] + v[n+M])[q],  F[2q+1] = FFT_M((v[n] - v[n+M]) w^n)[q]
//   REDFT01  (decimation in time)       v[n] = E[n] + conj(w)^n O[n],  v[n+M] = E[n] - conj(w)^n O[n],
//                                        E, O = inverse M-point transforms of the even / odd coefficients
// The butterfly v[n] +- v[n+M] pairs image rows y1 = 2n and y2 = N-1-2n.  It commutes with the transform along the other
// axis, so the ROW pass does it on its INPUT: one workgroup loads both rows, transforms (r1 + r2) into row y1 and
// (r1 - r2) into row y2 (row_pair_kernel).  The column pass then works on "half" tiles: half 0 = rows 2n (even rows),
// half 1 = rows N-1-2n (odd rows, bottom up), each an M-point FFT with the twiddle w^n folded into its load (REDFT10)
// or store (REDFT01).  Everything stays in place; the intermediate layout is internal to one execute().
template <class Re_, int N_, int K_, int T_, int... Rs>
struct ColHalfSpecT {
	typedef Re_ Re;
	typedef cx<Re_> CX;
	typedef PassArgsT<Re_> PA;
	typedef ColSpecT<Re_, N_ / 2, K_, T_, Rs...> B;          // stages / last stage of the M-point FFT are the plain column pass's
	typedef typename B::V V;
	typedef typename B::LC LC;
	typedef typename B::LR LR;
	static constexpr int NCS = B::NCS, VW = B::VW;
	static constexpr int N = N_, M = N_ / 2, K = K_, T = T_, NP = B::NP;
	static constexpr int NS = B::NS, NPH = B::NPH, WPE = 1;
	static constexpr size_t LDS = B::LDS;
	static constexpr int Y_ROUNDS = B::Y_ROUNDS, Q_ROUNDS = B::K_ROUNDS, NQ = (M / 2 + 1) * NP;
	static_assert(N_ % 4 == 0, "half tiles need N divisible by 4");
	template <int KIND> struct State : B::template State<KIND> {
		CX hw[KIND == KIND_REDFT10 ? Y_ROUNDS : 1];       // REDFT10, half 1: w^n of this thread's rows
	};

	// image row of tile row n in half h
	static DSP_HD int row_of(int n, int h) { return h ? N - 1 - 2 * n : 2 * n; }
	// FFT slot of the partner F[N-k] of coefficient k = 2q + h
	static DSP_HD int partner(int q, int h) { return h ? M - 1 - q : (q ? M - q : 0); }

	static DSP_HD void base(const PA &a, int work, long long &bin, long long &bout, int &h) { int tile; base(a, work, bin, bout, h, tile); }
	// tile = half * ntiles + tile of the row (PassGeom::zflags, zhalf = ntiles), per batch
	static DSP_HD void base(const PA &a, int work, long long &bin, long long &bout, int &h, int &tile)
	{
		const int per = 2 * a.ntiles;
		const int bt = work / per, t0 = work - bt * per;
		const int t1 = (a.zflags && a.mask) ? t0 : xcd_remap(t0, per);     // see ColSpecT::base
		tile = bt * per + t1;
		h = t1 / a.ntiles;
		const int t = t1 - h * a.ntiles;
		const int i1 = bt / a.nb0, i0 = bt - i1 * a.nb0;
		bin = i0 * a.sb0_in + i1 * a.sb1_in + (long long)t * K;
		bout = i0 * a.sb0_out + i1 * a.sb1_out + (long long)t * K;
	}

	template <int KIND, class ST>
	static DSP_HD void prefetch_tw(const PA &a, int h, int tid, ST &st)
	{
		if constexpr (KIND == KIND_REDFT10) {
			if (h) static_for<0, Y_ROUNDS>([&](auto i) {
				const int it = tid + i * T;
				if ((i + 1) * T <= M * NP || it < M * NP) st.hw[i] = a.H[it / NP];
			});
		} else {
			static_for<0, Q_ROUNDS>([&](auto i) {
				const int it = tid + i * T;
				if ((i + 1) * T <= NQ || it < NQ) {
					const int q = it / NP;
					if (h && q >= M / 2) return;
					st.tw[i] = a.T[2 * q + h];
				}
			});
		}
	}

	// the fused scan step masks the first pass's loads (loadv_m) and adds in the last pass's stores (storev_a), as in ColSpec
	template <int KIND, class ST>
	static DSP_HD void prefetch(const PA &a, long long bin, int h, int tid, ST &st, bool &hit)
	{
		if (a.mask && a.mask_mode == 1) prefetch_2step<KIND, 1>(a, bin, h, tid, st, hit);
		else if (a.mask && a.mask_mode == 2) prefetch_2step<KIND, 2>(a, bin, h, tid, st, hit);
		else if (a.mask) prefetch_m<KIND, true>(a, bin, h, tid, st, hit); else prefetch_m<KIND, false>(a, bin, h, tid, st, hit);
	}
	// masked loads with the ids from the plan's table (masked_two_step; one image: bin = tile * K): the items and their order are prefetch_m's
	template <int KIND, int EB, class ST>
	static DSP_HD void prefetch_2step(const PA &a, long long bin, int h, int tid, ST &st, bool &hit)
	{
		const int t = (int)(bin / K);
		if constexpr (KIND == KIND_REDFT10) {
			masked_two_step<EB, Y_ROUNDS, Re>(a, hit, [&](auto i, long long &off, long long &eoff) {
				const int it = tid + i * T;
				if (!((i + 1) * T <= M * NP || it < M * NP)) return false;
				const int n = it / NP, jp = it - n * NP, y = row_of(n, h);
				off = bin + (long long)y * a.es_in + VW * jp;
				eoff = eid_index(N, K, 2, t, y, VW * jp);
				return true;
			}, [&](auto i, V v) { st.pre[i] = B::g_get(v); });
			prefetch_tw<KIND>(a, h, tid, st);
		} else {
			auto item = [&](int i, int &q, int &jp) {
				const int it = tid + i * T;
				if (!((i + 1) * T <= NQ || it < NQ)) return false;
				q = it / NP; jp = it - q * NP;
				return !(h && q >= M / 2);
			};
			static_for<0, Q_ROUNDS>([&](auto i) { int q, jp; if (item(i, q, jp)) st.tw[i] = a.T[2 * q + h]; });
			masked_two_step<EB, 2 * Q_ROUNDS, Re>(a, hit, [&](auto j, long long &off, long long &eoff) {
				int q, jp;
				if (!item(j / 2, q, jp)) return false;
				const int k = 2 * q + h, y = j % 2 ? (k ? N - k : 0) : k;
				off = bin + VW * jp + (long long)y * a.es_in;
				eoff = eid_index(N, K, 2, t, y, VW * jp);
				return true;
			}, [&](auto j, V v) { st.pre[j] = B::g_get(v); });
		}
	}
	// WITH_TW = false leaves the twiddles of the tile (hw / tw) to prefetch_tw: the persistent kernel fetches them late, so that only
	// the rows of the next tile occupy registers during the butterfly stages
	template <int KIND, bool MASKED, class ST, bool WITH_TW = true>
	static DSP_HD void prefetch_m(const PA &a, long long bin, int h, int tid, ST &st, bool &hit)
	{
		if constexpr (KIND == KIND_REDFT10) {
			static_for<0, Y_ROUNDS>([&](auto i) {
				const int it = tid + i * T;
				if ((i + 1) * T <= M * NP || it < M * NP) {
					const int n = it / NP, jp = it - n * NP;
					st.pre[i] = B::g_get(loadv_m<MASKED, Re>(a, bin + (long long)row_of(n, h) * a.es_in + VW * jp, hit));
				}
			});
			// half 1: the twiddles w^n of this thread's rows, fetched behind the data so their latency hides under it (rows n .. n + T/NP
			// apart share nothing, but the 8.6 KB table stays in L1/L2)
			if constexpr (WITH_TW) prefetch_tw<KIND>(a, h, tid, st);
		} else {
			static_for<0, Q_ROUNDS>([&](auto i) {
				const int it = tid + i * T;
				if ((i + 1) * T <= NQ || it < NQ) {
					const int q = it / NP, jp = it - q * NP;
					if (h && q >= M / 2) return;
					const int k = 2 * q + h, km = k ? N - k : 0;
					if constexpr (WITH_TW) st.tw[i] = a.T[k];
					const long long p = bin + VW * jp;
					st.pre[2 * i] = B::g_get(loadv_m<MASKED, Re>(a, p + (long long)k * a.es_in, hit));
					st.pre[2 * i + 1] = B::g_get(loadv_m<MASKED, Re>(a, p + (long long)km * a.es_in, hit));
				}
			});
		}
	}

	// REDFT01 without accumulation closes in the last stage (see ColSpecT::last_store01): FFT output n of half h is image row row_of(n, h),
	// times conj(w)^n for the odd half
	static DSP_HD bool lean01(const PA &a) { return B::lean01(a); }
	template <int KIND> static DSP_HD bool skip_phase(const PA &a, int ph) { return KIND == KIND_REDFT01 && ph > NS && lean01(a); }
	template <int KIND> static DSP_HD bool barrier_after(const PA &a, int ph) { return (KIND == KIND_REDFT01 && lean01(a)) ? ph < NS : ph + 1 < NPH; }
	static DSP_HD void last_store01(const PA &a, const V *buf, long long bout, int h, int tid)
	{
		static_for<0, B::LAST_ROUNDS>([&](auto i) {
			const int it = tid + i * T;
			if (it < B::NBL * NP) {
				const int kb = it / NP, jp = it - kb * NP;
				const V *p = buf + (NS >= 2 ? B::padded(B::last_blk(kb) * B::RL) : 0) * NP + jp;
				LC x[B::RL];
				CX hw[B::RL];
				if (h) static_for<0, B::RL>([&](auto r) { hw[r] = a.H[kb + B::NBL * r]; });      // fetched beside the LDS reads
				static_for<0, B::RL>([&](auto r) { x[r] = B::l_get(p[r * NP]); });
				Dft<B::RL>::run(x);
				const Re sc = a.scale;
				static_for<0, B::RL>([&](auto r) {
					const int n = kb + B::NBL * r;
					LC F = x[r];
					if (h) F = B::lmul(F, hw[r]);
					*reinterpret_cast<V *>(a.out + bout + (long long)row_of(n, h) * a.es_out + VW * jp) = B::g_put(cmk<LR>(F.x * sc, -F.y * sc));
				});
			}
		});
	}

	template <int KIND, int PH, class ST>
	static DSP_HD void phase(const PA &a, V *buf, long long bout, int h, int tid, ST &st)
	{
		if constexpr (PH == 0) {
			if constexpr (KIND == KIND_REDFT10) {
				static_for<0, Y_ROUNDS>([&](auto i) {
					const int it = tid + i * T;
					if ((i + 1) * T <= M * NP || it < M * NP) {
						const int n = it / NP, jp = it - n * NP;
						LC v = st.pre[i];
						if (h) v = B::lmul(v, st.hw[i]);
						buf[B::padded(n) * NP + jp] = B::l_put(v);
					}
				});
			} else {
				static_for<0, Q_ROUNDS>([&](auto i) {
					const int it = tid + i * T;
					if ((i + 1) * T <= NQ || it < NQ) {
						const int q = it / NP, jp = it - q * NP;
						if (h && q >= M / 2) return;
						const int k = 2 * q + h, qm = partner(q, h);
						LC xk = st.pre[2 * i], xm = st.pre[2 * i + 1];
						if (k == 0) { xk = B::lscale(xk, a.in_scale0); xm = B::lzero(); }
						LC lo, hi;
						B::pre01(xk, xm, st.tw[i], lo, hi);
						buf[B::padded(q) * NP + jp] = B::l_put(lo);
						if (k > 0) buf[B::padded(qm) * NP + jp] = B::l_put(hi);
					}
				});
			}
		} else if constexpr (PH < NS) {
			B::template stage<PH - 1>(a, buf, tid);
		} else if constexpr (PH == NS) {
			if (KIND == KIND_REDFT01 && lean01(a)) last_store01(a, buf, bout, h, tid); else B::last_read(buf, st, tid);
		} else if constexpr (PH == NS + 1) {
			if (KIND == KIND_REDFT01 && lean01(a)) return;
			B::last_write(buf, st, tid);
		} else {
			if (KIND == KIND_REDFT01 && lean01(a)) return;
			if constexpr (KIND == KIND_REDFT10) {
				static_for<0, Q_ROUNDS>([&](auto ri) {
					const int it = tid + ri * T;
					if (!((ri + 1) * T <= NQ || it < NQ)) return;
					const int q = it / NP, jp = it - q * NP;
					if (h && q >= M / 2) return;
					const int k = 2 * q + h, km = k ? N - k : 0, qm = partner(q, h);
					const LC zk = B::l_get(buf[q * NP + jp]), zm = B::l_get(buf[qm * NP + jp]);
					const Re sc = a.scale, s0 = (k == 0) ? sc * a.out_scale0 : sc;
					const long long o = bout + VW * jp;
					LC r0, r1;
					B::post10(zk, zm, a.T[k], s0, sc, r0, r1);
					storev_a<Re>(a, o + (long long)k * a.es_out, B::g_put(r0));
					if (k > 0 && km != k) storev_a<Re>(a, o + (long long)km * a.es_out, B::g_put(r1));
				});
			} else {
				tloop<M * NP, T>(tid, [&](int it) {
					const int n = it / NP, jp = it - n * NP;
					LC F = B::l_get(buf[n * NP + jp]);
					if (h) F = B::lmul(F, a.H[n]);         // conj(w)^n O[n] = conj(w^n conj(O[n])); the conjugation is the sign below
					const Re sc = a.scale;
					storev_a<Re>(a, bout + (long long)row_of(n, h) * a.es_out + VW * jp, B::g_put(cmk<LR>(F.x * sc, -F.y * sc)));
				});
			}
		}
	}
};

template <int N_, int K_, int T_, int... Rs> using ColHalfSpec = ColHalfSpecT<float, N_, K_, T_, Rs...>;

// ---- channel lines (RowChanSpecT) ----
// work item b of a pass over `lines` interleaved lines of G channels -> (line, channel).  The G channel lines of a line read and write
// the SAME cache lines, a third each: they go to the same XCD (workgroups b, b + 8, ... share one: see xcd_remap), back to back, so that
// their partial stores meet in one L2 before the line is written back.  Measured on MI355X (tools/rowchan.hip, 3840 x 2160 x 3 doubles,
// out of place): 208-217 us with the channels of a line on different XCDs, 103 us on the same one (the interleaved kernel: 125-134 us).
template <int G> DSP_HD void chan_work(int b, int lines, int &line, int &ch)
{
	const int full = (lines >> 3) << 3;
	if (b < full * G) { const int x = b & 7, j = b >> 3, q = j / G; line = q * 8 + x; ch = j - q * G; }
	else { const int r = b - full * G; line = full + r / G; ch = r - (r / G) * G; }
}
// which interleaved row specs (sample type, N, channels) run as channel lines, and on which spec: spec_list.h
template <class Re, int N, int C> struct chan_lines_of { typedef void type; };
#define DSP_CHAN_TRAIT_D(N, G, T, ...) template <> struct chan_lines_of<double, N, G> { typedef RowChanSpecT<double, N, G, T, __VA_ARGS__> type; };
DSPFFT_ROW_CHAN_SPECS_F64(DSP_CHAN_TRAIT_D)
#undef DSP_CHAN_TRAIT_D

}  // namespace dspfft
