// dct_czt.h -- rows of a cosine series at an ARBITRARY sample spacing, by the chirp-z transform (Bluestein's convolution).
//
// zoom's basis (zoom/zoom.c:36-68) samples cos(pi (k + 1/2) n / N) at k = alpha (b + offset): affine in the output index b for all three
// bases (interpolated: alpha = den / num; native: alpha = 1 on N = len num / den; centered: alpha = (len - 1) den / (len num - den)).
// So a line of the product (zoom.c:361-375) is
//     out[b] = scale * sum'_{n < nc} C[n] cos(n (omega b + phi)),    omega = pi alpha / N,  phi = pi (alpha offset + 1/2) / N,   b < nout
// and, with n b = (n^2 + b^2 - (b - n)^2) / 2,
//     out[b] = Re{ E[b] * sum_n a[n] h[b - n] },   a[n] = g_n C[n] e^{i (phi n + omega n^2 / 2)},  h[m] = e^{-i omega m^2 / 2},  E[b] = scale e^{i omega b^2 / 2}
// -- a linear convolution of length nc + nout - 1, done as a circular one of a smooth length P >= nc + nout - 1 in LDS:
// forward FFT_P of a, times FFT_P(h) / P, inverse FFT_P.  Whatever omega: scales whose scaled length is not an integer and the `centered`
// basis, which the DCT-III grid of zoom_fft.hip / dct_duo.h cannot take (they kept the dense MFMA product, 3.3 ms at config 3's size).
//
// The FFT is RowSpecG's (decimation in frequency, in place, padded plane).  No reordering pass anywhere: the forward stages leave the
// spectrum in digit-reversed slot order, FFT(h) is tabulated in the same slot order (made by the same stages: czt_spectrum), and the
// inverse runs the stages backwards (conjugate twiddles, then the inverse butterfly, same addresses) back to natural order.  Phases per line:
//   F0        first forward stage, its inputs C[n] a-table[n] straight from global memory (zeros beyond nc)
//   F1..      middle forward stages in the plane
//   MID       last forward stage, times FFT(h), first inverse stage -- one butterfly's RL slots, in registers
//   ..I1      middle inverse stages
//   I0        last inverse stage to registers, times E, real part, stored
// i.e. 2 NS - 1 phases and 2 NS - 2 barriers for 2 NS butterfly stages.
#pragma once
#include "dct_spec.h"

namespace dspfft {

struct CztArgs {
	const float *in;          // lines of nc real samples, es_in floats apart
	float *out;               // lines of nout real samples, es_out floats apart
	const cf *atab;           // a-table: g_n e^{i (phi n + omega n^2 / 2)}, n < nc
	const cf *hspec;          // FFT_P(h) / P in the slot order the forward stages leave (czt_spectrum)
	const cf *etab;           // scale e^{i omega b^2 / 2}, b < nout
	const cf *W;              // exp(-2 pi i t / P), t < P
	long long in_pitch, out_pitch;     // floats between the lines of a group
	long long in_group, out_group;     // floats between groups (a group = `group` lines that share their output cache lines: the channels of an image row)
	int es_in, es_out, nc, nout, lines, group;
};

template <int P_, int T_, int... Rs>
struct CztSpecT {
	typedef RowSpecG<float, 2 * P_, 1, 1, T_, Rs...> F;        // its L = P-point complex FFT stages on one plane
	typedef cf CX;
	static constexpr int P = P_, T = T_, NS = F::NS, R0 = F::R0, RL = F::RL, SB = F::SB, PADC = F::PADC;
	static constexpr size_t LDS = F::LDS;
	static constexpr int NPH = 2 * NS - 1;
	static_assert(NS >= 2, "at least two stages");

	template <int R> static DSP_HD void idft(CX *x)
	{
		static_for<0, R>([&](auto r) { x[r] = cconj(x[r]); });
		Dft<R>::run(x);
		static_for<0, R>([&](auto r) { x[r] = cconj(x[r]); });
	}
	template <int R> static DSP_HD void powers(CX *w, CX w1)
	{
		w[1] = w1;
		static_for<2, R>([&](auto r) { if constexpr (r % 2 == 0) w[r] = csqr(w[r / 2]); else w[r] = cmul(w[r / 2], w[r - r / 2]); });
	}

	// F0: slots m + r SB from global memory (x = C[n] a[n], or a[n] alone when `in` is null: the spectrum of a table), first butterfly, twiddles
	static DSP_HD void f0(const CztArgs &a, CX *plane, long long bin, int tid)
	{
		tloop<SB, T>(tid, [&](int m) {
			CX x[R0];
			static_for<0, R0>([&](auto r) {
				const int n = m + r * SB;
				const bool ok = n < a.nc;
				const int nn = ok ? n : 0;                               // (no branch around the loads)
				const CX t = a.atab[nn];
				const float c = a.in ? a.in[bin + (long long)nn * a.es_in] : 1.f;
				x[r] = ok ? cmk<float>(t.x * c, t.y * c) : cmk<float>(0.f, 0.f);
			});
			Dft<R0>::run(x);
			CX w[R0];
			powers<R0>(w, a.W[m]);
			static_for<1, R0>([&](auto r) { x[r] = cmul(x[r], w[r]); });
			static_for<0, R0>([&](auto r) { plane[m + r * (SB + PADC)] = x[r]; });
		});
	}
	// forward stage I (1 <= I <= NS - 2)
	template <int I> static DSP_HD void fwd(const PassArgs &w, CX *plane, int tid) { F::template stage<I>(w, plane, tid); }
	// the last forward stage alone (czt_spectrum: leaves the spectrum in slot order)
	static DSP_HD void flast(CX *plane, int tid)
	{
		tloop<P / RL, T>(tid, [&](int q) {
			CX *p = plane + F::padded(q * RL);
			CX x[RL];
			static_for<0, RL>([&](auto r) { x[r] = p[r]; });
			Dft<RL>::run(x);
			static_for<0, RL>([&](auto r) { p[r] = x[r]; });
		});
	}
	// MID: last forward stage, times the kernel's spectrum, first inverse stage
	static DSP_HD void mid(const CztArgs &a, CX *plane, int tid)
	{
		tloop<P / RL, T>(tid, [&](int q) {
			CX *p = plane + F::padded(q * RL);
			CX x[RL], h[RL];
			static_for<0, RL>([&](auto r) { h[r] = a.hspec[q * RL + r]; });
			static_for<0, RL>([&](auto r) { x[r] = p[r]; });
			Dft<RL>::run(x);
			static_for<0, RL>([&](auto r) { x[r] = cmul(x[r], h[r]); });
			idft<RL>(x);
			static_for<0, RL>([&](auto r) { p[r] = x[r]; });
		});
	}
	// inverse of forward stage I (1 <= I <= NS - 2): conjugate twiddles, inverse butterfly, same slots
	template <int I> static DSP_HD void inv(const CztArgs &a, CX *plane, int tid)
	{
		constexpr int R = pack_get<I, Rs...>(), Lc = pack_lc<I, Rs...>(), M1 = Lc / R, NB = P / R, TW = P / Lc;
		tloop<NB, T>(tid, [&](int q) {
			const int blk = q / M1, m = q - blk * M1;
			CX *p = plane + F::padded(blk * Lc) + m;
			CX x[R];
			static_for<0, R>([&](auto r) { x[r] = p[r * M1]; });
			if constexpr (M1 > 1) {
				CX w[R];
				powers<R>(w, a.W[m * TW]);
				static_for<1, R>([&](auto r) { x[r] = cmulc(x[r], w[r]); });
			}
			idft<R>(x);
			static_for<0, R>([&](auto r) { p[r * M1] = x[r]; });
		});
	}
	// I0: inverse of the first stage; sample b = m + r SB leaves as Re(E[b] y[b])
	static DSP_HD void i0(const CztArgs &a, const CX *plane, long long bout, int tid)
	{
		tloop<SB, T>(tid, [&](int m) {
			CX x[R0], w[R0];
			static_for<0, R0>([&](auto r) { x[r] = plane[m + r * (SB + PADC)]; });
			powers<R0>(w, a.W[m]);
			static_for<1, R0>([&](auto r) { x[r] = cmulc(x[r], w[r]); });
			idft<R0>(x);
			static_for<0, R0>([&](auto r) {
				const int b = m + r * SB;
				if (b < a.nout) {
					const CX e = a.etab[b];
					a.out[bout + (long long)b * a.es_out] = e.x * x[r].x - e.y * x[r].y;
				}
			});
		});
	}
	// phase ph of a line (0 .. NPH - 1)
	template <int PH> static DSP_HD void phase(const CztArgs &a, const PassArgs &w, CX *plane, long long bin, long long bout, int tid)
	{
		if constexpr (PH == 0) f0(a, plane, bin, tid);
		else if constexpr (PH < NS - 1) fwd<PH>(w, plane, tid);
		else if constexpr (PH == NS - 1) mid(a, plane, tid);
		else if constexpr (PH < NPH - 1) inv<2 * NS - 2 - PH>(a, plane, tid);
		else i0(a, plane, bout, tid);
	}
	// line `work` of the launch -> (group, member); the members of a group run on workgroups b, b + 8, ... of one XCD back to back (chan_work)
	static DSP_HD void base(const CztArgs &a, int work, long long &bin, long long &bout)
	{
		int grp = work, mem = 0;
		if (a.group == 3) chan_work<3>(work, a.lines / 3, grp, mem);
		else if (a.group > 1) { grp = work / a.group; mem = work - grp * a.group; }
		bin = (long long)grp * a.in_group + (long long)mem * a.in_pitch;
		bout = (long long)grp * a.out_group + (long long)mem * a.out_pitch;
	}
};

// ---- tables, one entry per call (double arithmetic; the phases are reduced before the sine / cosine) ----
DSP_HD void czt_sincos_reduced(double ang, double &s, double &c)
{
	const double twopi = 6.28318530717958647692528676655900577;
	ang -= twopi * floor(ang / twopi + 0.5);
	s = sin(ang); c = cos(ang);
}
// a-table entry n: g_n e^{i (phi n + omega n^2 / 2)}, g_0 = 1/2 (the halved first term of the series)
DSP_HD cf czt_a_entry(int n, double omega, double phi)
{
	double s, c;
	czt_sincos_reduced(phi * n + 0.5 * omega * (double)n * (double)n, s, c);
	const double g = n ? 1.0 : 0.5;
	return cmk<float>((float)(g * c), (float)(g * s));
}
// e-table entry b: scale e^{i omega b^2 / 2}
DSP_HD cf czt_e_entry(int b, double omega, double scale)
{
	double s, c;
	czt_sincos_reduced(0.5 * omega * (double)b * (double)b, s, c);
	return cmk<float>((float)(scale * c), (float)(scale * s));
}
// the convolution kernel on the circle of P points, index t: h[m] / P with m = t for t < nout, m = t - P for t > P - nc, 0 between
DSP_HD cf czt_h_entry(int t, int P, int nc, int nout, double omega)
{
	int m;
	if (t < nout) m = t; else if (t > P - nc) m = t - P; else return cmk<float>(0.f, 0.f);
	double s, c;
	czt_sincos_reduced(-0.5 * omega * (double)m * (double)m, s, c);
	return cmk<float>((float)(c / P), (float)(s / P));
}

}  // namespace dspfft
