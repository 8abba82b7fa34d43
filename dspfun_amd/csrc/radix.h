// radix.h -- in-register small DFTs (forward, e^{-2 pi i/R}) with compile-time twiddles.
//
// Everything here is plain C++17 usable from device code (hipcc) and, for the CPU-side
// emulation tests, from host code (g++).  No reference code corresponds to this file: the
// reference delegates all transform arithmetic to FFTW (include/precision.h:115).
#pragma once
#include <stdint.h>
#include <type_traits>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DSP_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define DSP_HD inline __attribute__((always_inline))
#endif

#if !defined(__HIPCC__)
// host-side stand-ins for the HIP vector types (CPU emulation build only)
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) double2 { double x, y; };
struct alignas(32) double4 { double x, y, z, w; };
#endif

namespace dspfft {

// Two floats that go through every operation together.  On gfx950 a 2 x float vector maps onto the packed FP32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two results per lane per issue), which is how the column kernels carry the two
// complex signals of a lane through the butterflies at half the instruction count.  The host (g++ emulation) stand-in does the same
// arithmetic element by element.
#if defined(__clang__)
typedef float Pk2 __attribute__((ext_vector_type(2)));
DSP_HD Pk2 pk2(float a, float b) { Pk2 r; r.x = a; r.y = b; return r; }
#else
struct Pk2 {
	float x, y;
	constexpr Pk2() : x(0), y(0) {}
	constexpr Pk2(float s) : x(s), y(s) {}
	constexpr Pk2(float a, float b) : x(a), y(b) {}
	Pk2 &operator+=(Pk2 o) { x += o.x; y += o.y; return *this; }
};
constexpr Pk2 operator+(Pk2 a, Pk2 b) { return Pk2(a.x + b.x, a.y + b.y); }
constexpr Pk2 operator-(Pk2 a, Pk2 b) { return Pk2(a.x - b.x, a.y - b.y); }
constexpr Pk2 operator*(Pk2 a, Pk2 b) { return Pk2(a.x * b.x, a.y * b.y); }
constexpr Pk2 operator-(Pk2 a) { return Pk2(-a.x, -a.y); }
DSP_HD Pk2 pk2(float a, float b) { return Pk2(a, b); }
#endif

// complex number over float (the tuned path; 8 bytes, same layout as float2) or double (the fftw_ double API)
template <class R> struct cx { typedef R real; R x, y; };
typedef cx<float> cf;
typedef cx<double> cd;
template <class R> struct same_t { typedef R type; };   // keeps a scalar argument out of template deduction

template <class R> DSP_HD cx<R> cmk(R x, typename same_t<R>::type y) { cx<R> r; r.x = x; r.y = y; return r; }
template <class R> DSP_HD cx<R> cadd(cx<R> a, cx<R> b) { return cmk<R>(a.x + b.x, a.y + b.y); }
template <class R> DSP_HD cx<R> csub(cx<R> a, cx<R> b) { return cmk<R>(a.x - b.x, a.y - b.y); }
template <class R> DSP_HD cx<R> cmul(cx<R> a, cx<R> b) { return cmk<R>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <class R> DSP_HD cx<R> cmulc(cx<R> a, cx<R> b) { return cmk<R>(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)
template <class R> DSP_HD cx<R> cconj(cx<R> a) { return cmk<R>(a.x, -a.y); }
template <class R> DSP_HD cx<R> cscale(cx<R> a, typename same_t<R>::type s) { return cmk<R>(a.x * s, a.y * s); }
template <class R> DSP_HD cx<R> cmul_mi(cx<R> a) { return cmk<R>(a.y, -a.x); }   // a * (-i)
template <class R> DSP_HD cx<R> cmul_pi(cx<R> a) { return cmk<R>(-a.y, a.x); }   // a * (+i)
template <class R> DSP_HD cx<R> cadd_mi(cx<R> a, cx<R> b) { return cmk<R>(a.x + b.y, a.y - b.x); }   // a + (-i) b
template <class R> DSP_HD cx<R> csub_mi(cx<R> a, cx<R> b) { return cmk<R>(a.x - b.y, a.y + b.x); }   // a - (-i) b

// ---- cx<float> as ONE packed pair on the device (round 5) ----
// The row kernels carry one signal per butterfly, (re, im) in a register pair; written component by component every complex add is two
// v_add_f32 and every product four v_mul / v_fma.  The same operations on the pair as a 2 x float vector are v_pk_add_f32 (one issue) and
// v_pk_mul_f32 + v_pk_fma_f32 with the operand halves picked by op_sel (two issues), and the kernels that run at the vector issue rate
// (8K row pairs, motion's 8-bit row ends: profiles/r05_isa_*.txt) get through their butterflies in ~0.6 of the issues.  The results are the
// same sums and products (one rounding per operation either way; which of a product pair is the FMA's addend is the compiler's choice in both).
// Host (g++ emulation): the templates above.
// (only where the ISA has packed FP32 in VOP3P: gfx90a and the gfx94x / gfx950 family.  The library builds for gfx950 alone; the guard keeps an ARCH
// override of the Makefile or a plan-time compile for another device on the scalar templates instead of failing in the assembler.)
#if defined(__clang__) && defined(__HIP_DEVICE_COMPILE__) && !defined(DSP_NO_PK_CX) && \
    (defined(__gfx90a__) || defined(__gfx940__) || defined(__gfx941__) || defined(__gfx942__) || defined(__gfx950__))
DSP_HD Pk2 pk_of(cf a) { return pk2(a.x, a.y); }
DSP_HD cf cf_of(Pk2 v) { cf r; r.x = v.x; r.y = v.y; return r; }
DSP_HD cf cadd(cf a, cf b) { return cf_of(pk_of(a) + pk_of(b)); }
DSP_HD cf csub(cf a, cf b) { return cf_of(pk_of(a) - pk_of(b)); }
DSP_HD cf cscale(cf a, float s) { return cf_of(pk_of(a) * pk2(s, s)); }
// The products and the quarter-turn sums need one operand half negated: the instructions take that as a modifier (neg_lo / neg_hi), the compiler
// spends a v_xor (and often a v_mov for the swapped halves) on it -- so these few are spelled out.  Plain asm, no side effects: scheduled and
// allocated like any other instruction.  op_sel / op_sel_hi: which half of each source feeds the low / high result.
// (a.x b.x - a.y b.y, a.x b.y + a.y b.x) = a.xx * b + a.yy * (-b.y, b.x)
DSP_HD cf cmul(cf a, cf b)
{
	Pk2 t, r; const Pk2 va = pk_of(a), vb = pk_of(b);
	asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(va), "v"(vb));
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(va), "v"(vb), "v"(t));
	return cf_of(r);
}
// a conj(b) = (a.x b.x + a.y b.y, a.y b.x - a.x b.y) = a * b.xx + a.yx * (b.y, -b.y)
DSP_HD cf cmulc(cf a, cf b)
{
	Pk2 t, r; const Pk2 va = pk_of(a), vb = pk_of(b);
	asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(va), "v"(vb));
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(va), "v"(vb), "v"(t));
	return cf_of(r);
}
// a + (-i) b = (a.x + b.y, a.y - b.x);  a - (-i) b = (a.x - b.y, a.y + b.x)
DSP_HD cf cadd_mi(cf a, cf b)
{
	Pk2 r; const Pk2 va = pk_of(a), vb = pk_of(b);
	asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(va), "v"(vb));
	return cf_of(r);
}
DSP_HD cf csub_mi(cf a, cf b)
{
	Pk2 r; const Pk2 va = pk_of(a), vb = pk_of(b);
	asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(va), "v"(vb));
	return cf_of(r);
}
#define DSP_PK_CX 1
#endif

// ---- compile-time trigonometry: cos/sin(2 pi k / n), exact octant reduction on (k, n) ----
namespace ct {
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double taylor_cos(double x) {   // |x| <= pi/4
	double x2 = x * x, term = 1.0, sum = 1.0;
	for (int i = 1; i <= 12; i++) { term *= -x2 / double((2 * i - 1) * (2 * i)); sum += term; }
	return sum;
}
constexpr double taylor_sin(double x) {
	double x2 = x * x, term = x, sum = x;
	for (int i = 1; i <= 12; i++) { term *= -x2 / double((2 * i) * (2 * i + 1)); sum += term; }
	return sum;
}
struct cs { double c, s; };
// cos and sin of 2*pi*k/n
constexpr cs cossin(long k, long n) {
	k %= n; if (k < 0) k += n;
	// angle fraction a = k/n in [0,1)
	bool neg_s = false;
	if (2 * k > n) { k = n - k; neg_s = true; }             // a > 1/2: reflect about pi
	bool neg_c = false;
	if (4 * k > n) { k = n - 2 * k; n = 2 * n; neg_c = true;   // a in (1/4,1/2]: pi - theta; k/n := 1/2 - a
		// now angle fraction = (n_old - 2k_old) / (2 n_old)
	}
	// a in [0, 1/4]
	bool swap = false;
	if (8 * k > n) { // a in (1/8, 1/4]: cos(theta) = sin(pi/2 - theta)
		k = n - 4 * k; n = 4 * n; swap = true;               // 1/4 - a = (n - 4k) / (4n)
	}
	double th = 2.0 * kPi * double(k) / double(n);
	double c = taylor_cos(th), s = taylor_sin(th);
	if (swap) { double t = c; c = s; s = t; }
	if (neg_c) c = -c;
	if (neg_s) s = -s;
	return cs{c, s};
}
}  // namespace ct

// Forward twiddle table w_R^e = exp(-2 pi i e / R), e in [0, R)
template <int R>
struct TwTab {
	double re[R], im[R];      // rounded to the kernel's real type where they are used
	constexpr TwTab() : re{}, im{} {
		for (int e = 0; e < R; e++) {
			ct::cs v = ct::cossin(e, R);
			re[e] = v.c; im[e] = -v.s;
		}
	}
};
template <int R> struct TwHolder { static constexpr TwTab<R> tab = TwTab<R>(); };

// compile-time loop
template <int I, int N, class F>
DSP_HD void static_for(F &&f) {
	if constexpr (I < N) {
		f(std::integral_constant<int, I>{});
		static_for<I + 1, N>(f);
	}
}

// x * w_R^E with the trivial cases folded
template <int R, int E, class Re>
DSP_HD cx<Re> twmul(cx<Re> a) {
	constexpr int e = ((E % R) + R) % R;
	if constexpr (e == 0) return a;
	else if constexpr (2 * e == R) return cmk<Re>(-a.x, -a.y);
	else if constexpr (4 * e == R) return cmul_mi(a);
	else if constexpr (4 * e == 3 * R) return cmul_pi(a);
	else {
		constexpr Re wr = (Re)TwHolder<R>::tab.re[e], wi = (Re)TwHolder<R>::tab.im[e];
#if defined(DSP_PK_CX)
		if constexpr (std::is_same<Re, float>::value) return cf_of(__builtin_elementwise_fma(pk2(a.y, a.x), pk2(-wi, wi), pk_of(a) * pk2(wr, wr)));
		else
#endif
		return cmk<Re>(a.x * wr - a.y * wi, a.x * wi + a.y * wr);
	}
}

constexpr int first_factor(int r) {
	if (r % 4 == 0 && r > 4) return 4;
	if (r % 2 == 0) return 2;
	for (int f = 3; f * f <= r; f += 2) if (r % f == 0) return f;
	return r;
}
constexpr bool is_prime_radix(int r) { return first_factor(r) == r; }

template <int R> struct Dft;

template <> struct Dft<1> { template <class C> static DSP_HD void run(C *) {} };
template <> struct Dft<2> {
	template <class C> static DSP_HD void run(C *x) { C a = x[0], b = x[1]; x[0] = cadd(a, b); x[1] = csub(a, b); }
};
template <> struct Dft<4> {
	template <class C> static DSP_HD void run(C *x) {
		C s0 = cadd(x[0], x[2]), s1 = csub(x[0], x[2]), s2 = cadd(x[1], x[3]), d = csub(x[1], x[3]);
		x[0] = cadd(s0, s2); x[1] = cadd_mi(s1, d); x[2] = csub(s0, s2); x[3] = csub_mi(s1, d);
	}
};

// odd primes: pair up q and p-q
template <int P>
struct DftOddPrime {
	template <class C> static DSP_HD void run(C *x) {
		typedef typename C::real Re;
		constexpr int H = (P - 1) / 2;
		C a[H], b[H];
		static_for<0, H>([&](auto q) { a[q] = cadd(x[q + 1], x[P - 1 - q]); b[q] = csub(x[q + 1], x[P - 1 - q]); });
		C x0 = x[0];
		C sum = x0;
		static_for<0, H>([&](auto q) { sum = cadd(sum, a[q]); });
		x[0] = sum;
#if defined(DSP_PK_CX)
		if constexpr (std::is_same<C, cf>::value) {
			static_for<1, H + 1>([&](auto r) {
				Pk2 cc = pk_of(x0), ss = pk2(0.f, 0.f);
				static_for<0, H>([&](auto q) {
					constexpr int e = ((q + 1) * r) % P;
					constexpr float c = (float)TwHolder<P>::tab.re[e];
					constexpr float s = (float)-TwHolder<P>::tab.im[e];
					cc = __builtin_elementwise_fma(pk_of(a[q]), pk2(c, c), cc);
					ss = __builtin_elementwise_fma(pk_of(b[q]), pk2(s, s), ss);
				});
				// X[r] = C - i S ; X[P-r] = C + i S
				x[r] = cadd_mi(cf_of(cc), cf_of(ss));
				x[P - r] = csub_mi(cf_of(cc), cf_of(ss));
			});
			return;
		}
#endif
		static_for<1, H + 1>([&](auto r) {
			Re cr = x0.x, ci = x0.y, sr = 0, si = 0;
			static_for<0, H>([&](auto q) {
				constexpr int e = ((q + 1) * r) % P;
				constexpr Re c = (Re)TwHolder<P>::tab.re[e];
				constexpr Re s = (Re)-TwHolder<P>::tab.im[e];         // sin(2 pi e / P)
				cr += a[q].x * c; ci += a[q].y * c;
				sr += b[q].x * s; si += b[q].y * s;
			});
			// X[r] = C - i S ; X[P-r] = C + i S
			x[r] = cmk<Re>(cr + si, ci - sr);
			x[P - r] = cmk<Re>(cr - si, ci + sr);
		});
	}
};
template <> struct Dft<3> : DftOddPrime<3> {};
template <> struct Dft<5> : DftOddPrime<5> {};
template <> struct Dft<7> : DftOddPrime<7> {};
template <> struct Dft<11> : DftOddPrime<11> {};
template <> struct Dft<13> : DftOddPrime<13> {};

// composite: R = P*Q, decimation in frequency; natural order in and out
template <int R>
struct Dft {
	template <class C> static DSP_HD void run(C *x) {
		constexpr int P = first_factor(R), Q = R / P;
		static_assert(P != R, "prime radix without a specialisation");
		C y[P][Q];
		static_for<0, Q>([&](auto m) {
			C t[P];
			static_for<0, P>([&](auto n1) { t[n1] = x[n1 * Q + m]; });
			Dft<P>::run(t);
			static_for<0, P>([&](auto k1) { y[k1][m] = twmul<R, m * k1>(t[k1]); });
		});
		static_for<0, P>([&](auto k1) {
			Dft<Q>::run(y[k1]);
			static_for<0, Q>([&](auto q) { x[k1 + P * q] = y[k1][q]; });
		});
	}
};

}  // namespace dspfft
